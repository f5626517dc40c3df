// How much other work does ONE wave hide behind its own dependent v_mfma_f32_32x32x2_f32 chain?  (1 workgroup of 4 waves per CU)
// Per MFMA: NV independent VALU (v_fma), ND ds_read_b128, NS SALU.  Prints cycles per MFMA (64 = the matrix pipe's issue interval).
// build+run on the GPU box:  hipcc --offload-arch=gfx950 -O3 tools/mfma_overlap.hip -o /tmp/mfma_overlap && /tmp/mfma_overlap
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int NV, int ND, int NS, int NG = 0, int NW = 0, int NACC = 1>
__global__ void __launch_bounds__(256) k(float* out, int iters, float a0, float b0) {
  __shared__ float lds[256 * 36];
  f32x16 acc;
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  float a = a0 + threadIdx.x, b = b0;
  float v[8];
  for (int i = 0; i < 8; ++i) v[i] = a + i;
  f32x4 d[4];
  for (int i = 0; i < 4; ++i) d[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  for (int i = threadIdx.x; i < 256 * 36; i += 256) lds[i] = (float)i;
  __syncthreads();
  const unsigned loff = (threadIdx.x & 63) * 144u;     // byte offset inside the block's only LDS array
  f32x4 gl[4];
  for (int i = 0; i < 4; ++i) gl[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  const f32x4* gp = reinterpret_cast<const f32x4*>(out) + threadIdx.x;
  f32x16 acc2;
  for (int r = 0; r < 16; ++r) acc2[r] = 0.f;
  int sacc = 0;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      if (NACC == 2 && (u & 1)) asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+v"(acc2) : "v"(a), "v"(b));
      else asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b));
#pragma unroll
      for (int i = 0; i < NV; ++i) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(v[i & 7]) : "v"(b));
#pragma unroll
      for (int i = 0; i < ND; ++i) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(d[i & 3]) : "v"(loff), "n"(16 * (i & 7)));
#pragma unroll
      for (int i = 0; i < NS; ++i) asm volatile("s_add_u32 %0, %0, 1" : "+s"(sacc) : : "scc");
    }
#pragma unroll
    for (int i = 0; i < NG; ++i) gl[i & 3] += gp[(size_t)((it * 64 + i * 7) & 4095) * 64];     // plain loads, compiler-scheduled
#pragma unroll
    for (int i = 0; i < NW; ++i) asm volatile("ds_write_b128 %0, %1 offset:%2" : : "v"(loff), "v"(d[i & 3]), "n"(16 * (i & 7)));
    asm volatile("s_waitcnt lgkmcnt(0)");

  }
  float s = (float)sacc;
  for (int r = 0; r < 16; ++r) s += acc[r];
  for (int i = 0; i < 8; ++i) s += v[i];
  for (int i = 0; i < 4; ++i) s += d[i][0] + gl[i][0];
  for (int r = 0; r < 16; ++r) s += acc2[r];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int NV, int ND, int NS, int NG = 0, int NW = 0, int NACC = 1>
void run(int wgs_per_cu) {
  const int grid = 256 * wgs_per_cu, iters = 2000;
  float* out; hipMalloc(&out, (size_t)grid * 256 * 4 + (64 << 20)); hipMemset(out, 0, (size_t)grid * 256 * 4 + (64 << 20));
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(HIP_KERNEL_NAME(k<NV, ND, NS, NG, NW, NACC>), dim3(grid), dim3(256), 0, 0, out, 16, 1.f, 1.f);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL(HIP_KERNEL_NAME(k<NV, ND, NS, NG, NW, NACC>), dim3(grid), dim3(256), 0, 0, out, iters, 1.f, 1.f);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double mfmas = (double)iters * 16 * wgs_per_cu;     // per SIMD
  printf("[gload/16=%d dswrite/16=%d chains=%d] VALU=%2d DS=%d SALU=%2d per MFMA, %d wave(s)/SIMD: %.3f ms  -> %.1f ns per MFMA per SIMD = %.1f cycles @2.4GHz\n", NG, NW, NACC, NV, ND, NS, wgs_per_cu, ms,
         ms * 1e6 / mfmas, ms * 1e6 / mfmas * 2.4);
  hipFree(out);
}
int main() {
  run<0, 0, 0>(1); run<0, 0, 0, 0, 0, 2>(1); run<4, 0, 0>(1); run<0, 1, 0>(1); run<0, 1, 0, 0, 0, 2>(1);
  run<0, 0, 4>(1); run<0, 0, 8>(1); run<0, 1, 4>(1);
  run<0, 1, 0, 2, 0, 2>(1); run<0, 1, 0, 2, 2, 2>(1); run<0, 1, 4, 2, 2, 2>(1); run<1, 1, 4, 2, 2, 2>(1);
  run<0, 1, 4, 2, 2, 2>(2); run<4, 1, 4, 2, 2, 2>(2);
  return 0;
}
