// Sustained fp32 matrix-core rate of this GPU: back-to-back independent v_mfma_f32_32x32x2_f32 from registers only
// (no LDS, no memory) - the practical ceiling under the 157.3 TFLOP/s datasheet figure (clock under MFMA load).
// build+run on the GPU box:  hipcc --offload-arch=gfx950 -O3 tools/mfma_peak.hip -o /tmp/mfma_peak && /tmp/mfma_peak
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int NACC>
__global__ void __launch_bounds__(256) mfma_loop(float* out, int iters, float a0, float b0) {
  f32x16 acc[NACC];
  for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  float a = a0 + threadIdx.x, b = b0;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 8; ++u)
#pragma unroll
      for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
  }
  float s = 0.f;
  for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int NACC>
void run(int wgs_per_cu, int iters) {
  const int grid = 256 * wgs_per_cu;
  float* out; hipMalloc(&out, (size_t)grid * 256 * 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(mfma_loop<NACC>, dim3(grid), dim3(256), 0, 0, out, 64, 1.f, 1.f);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL(mfma_loop<NACC>, dim3(grid), dim3(256), 0, 0, out, iters, 1.f, 1.f);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double flops = (double)grid * 4 /*waves*/ * iters * 8.0 * NACC * (32.0 * 32 * 2 * 2);
  printf("accumulators/wave=%d  workgroups/CU=%d  iters=%d : %.3f ms  %.1f TFLOP/s  (%.1f %% of 157.3)\n", NACC, wgs_per_cu, iters, ms,
         flops / ms / 1e9, flops / ms / 1e9 / 157.3 * 100);
  hipFree(out);
}
int main() {
  run<1>(1, 20000); run<1>(2, 20000); run<2>(1, 10000); run<4>(1, 5000); run<4>(2, 5000); run<4>(2, 40000);
  return 0;
}
