// Two questions about running the fp32 contractions on the bf16 matrix pipe of gfx950 (v_mfma_f32_32x32x16_bf16):
//  (1) throughput: cycles per MFMA with NV VALU / ND ds_read_b128 per MFMA next to it, 1 or 2 waves per SIMD
//      (the fp32 MFMA shares the vector ALUs, tools/mfma_overlap.hip; does the bf16 one?)
//  (2) accuracy: C = A·Bᵀ (32x32, K terms) from fp32 operands split into 2 / 3 bf16 pieces (3 / 6 product terms) against a float64
//      host product, next to the plain fp32 MFMA.
// build+run on the GPU box:  hipcc --offload-arch=gfx950 -O3 tools/mfma_bf16_probe.hip -o /tmp/bfp && /tmp/bfp
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));

template <int NV, int ND, int NT>
__global__ void __launch_bounds__(256) kt(float* out, int iters, float a0) {
  __shared__ float lds[256 * 36];
  f32x16 acc;
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  s16x8 a[3], b[3];
  for (int p = 0; p < 3; ++p)
    for (int e = 0; e < 8; ++e) { a[p][e] = (short)(0x3f80 + threadIdx.x + p); b[p][e] = (short)(0x3f00 + e + p); }
  float v[8];
  for (int i = 0; i < 8; ++i) v[i] = a0 + i;
  f32x4 d[4];
  for (int i = 0; i < 4; ++i) d[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  for (int i = threadIdx.x; i < 256 * 36; i += 256) lds[i] = (float)i;
  __syncthreads();
  const unsigned loff = (threadIdx.x & 63) * 144u;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 12; ++u) {
      asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc) : "v"(a[u % NT]), "v"(b[(u / NT) % NT]));
#pragma unroll
      for (int i = 0; i < NV; ++i) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(v[i & 7]) : "v"(a0));
#pragma unroll
      for (int i = 0; i < ND; ++i) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(d[i & 3]) : "v"(loff), "n"(16 * (i & 7)));
    }
    asm volatile("s_waitcnt lgkmcnt(0)");
  }
  float s = 0.f;
  for (int r = 0; r < 16; ++r) s += acc[r];
  for (int i = 0; i < 8; ++i) s += v[i];
  for (int i = 0; i < 4; ++i) s += d[i][0];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int NV, int ND, int NT = 3>
void run(int wgs_per_cu) {
  const int grid = 256 * wgs_per_cu, iters = 4000;
  float* out; hipMalloc(&out, (size_t)grid * 256 * 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(HIP_KERNEL_NAME(kt<NV, ND, NT>), dim3(grid), dim3(256), 0, 0, out, 16, 1.f);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL(HIP_KERNEL_NAME(kt<NV, ND, NT>), dim3(grid), dim3(256), 0, 0, out, iters, 1.f);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double mfmas = (double)iters * 12 * wgs_per_cu;
  printf("bf16 32x32x16: VALU=%2d DS128=%d per MFMA, %d wave(s)/SIMD: %.3f ms -> %.1f cycles @2.4GHz per MFMA per SIMD (%.0f TF bf16 chip)\n", NV, ND, wgs_per_cu, ms,
         ms * 1e6 / mfmas * 2.4, 32768.0 * mfmas * 4 * 256 / (ms * 1e-3) * 1e-12);
  hipFree(out);
}

// ---- accuracy ---------------------------------------------------------------------------------------------------------------
// operands per lane: A[i = lane&31][k = 16*s + 8*(lane>>5) + e], B[j = lane&31][same k]; pieces by truncation (exact remainders)
__device__ inline void split3(float x, unsigned short& p0, unsigned short& p1, unsigned short& p2) {
  const unsigned u0 = __float_as_uint(x) & 0xffff0000u;
  const float r1 = x - __uint_as_float(u0);
  const unsigned u1 = __float_as_uint(r1) & 0xffff0000u;
  const float r2 = r1 - __uint_as_float(u1);
  p0 = (unsigned short)(u0 >> 16); p1 = (unsigned short)(u1 >> 16); p2 = (unsigned short)(__float_as_uint(r2) >> 16);
}
__device__ inline void split_rn(float x, unsigned short& p0, unsigned short& p1, unsigned short& p2) {   // round-to-nearest pieces
  auto rn = [](float f) { unsigned u = __float_as_uint(f); u += 0x7fffu + ((u >> 16) & 1u); return u & 0xffff0000u; };
  const unsigned u0 = rn(x);
  const float r1 = x - __uint_as_float(u0);
  const unsigned u1 = rn(r1);
  const float r2 = r1 - __uint_as_float(u1);
  p0 = (unsigned short)(u0 >> 16); p1 = (unsigned short)(u1 >> 16); p2 = (unsigned short)(rn(r2) >> 16);
}
template <int MODE>   // 0 fp32 MFMA, 1: 3 terms (2 pieces), 2: 6 terms (3 pieces, truncation), 3: 6 terms round-to-nearest pieces, 4: 3 terms RN
__global__ void __launch_bounds__(64) kacc(const float* A, const float* B, float* C, int K) {
  const int lane = threadIdx.x, i = lane & 31, h = lane >> 5;
  f32x16 acc;
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  if (MODE == 0) {
    for (int k = 0; k < K; k += 2) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(A[i * K + k + h], B[i * K + k + h], acc, 0, 0, 0);
  } else {
    for (int s = 0; s < K; s += 16) {
      s16x8 a[3], b[3];
      for (int e = 0; e < 8; ++e) {
        unsigned short p0, p1, p2;
        const float av = A[i * K + s + 8 * h + e], bv = B[i * K + s + 8 * h + e];
        if (MODE == 3 || MODE == 4) split_rn(av, p0, p1, p2); else split3(av, p0, p1, p2);
        a[0][e] = (short)p0; a[1][e] = (short)p1; a[2][e] = (short)p2;
        if (MODE == 3 || MODE == 4) split_rn(bv, p0, p1, p2); else split3(bv, p0, p1, p2);
        b[0][e] = (short)p0; b[1][e] = (short)p1; b[2][e] = (short)p2;
      }
      // small terms first
      if (MODE == 2 || MODE == 3) {
        asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc) : "v"(a[2]), "v"(b[0]));
        asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc) : "v"(a[0]), "v"(b[2]));
        asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc) : "v"(a[1]), "v"(b[1]));
      }
      asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc) : "v"(a[1]), "v"(b[0]));
      asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc) : "v"(a[0]), "v"(b[1]));
      asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc) : "v"(a[0]), "v"(b[0]));
    }
  }
  for (int r = 0; r < 16; ++r) C[((r & 3) + 8 * (r >> 2) + 4 * h) * 32 + i] = acc[r];
}
template <int MODE>
void accuracy(int K, const char* what) {
  std::vector<float> A(32 * K), B(32 * K), C(1024);
  srand(1234 + K);
  auto nrm = [] { double u = (rand() + 1.0) / (RAND_MAX + 2.0), v = (rand() + 1.0) / (RAND_MAX + 2.0); return (float)(sqrt(-2 * log(u)) * cos(6.283185307179586 * v)); };
  for (auto& x : A) x = nrm();
  for (auto& x : B) x = nrm() * 0.05f;
  float *dA, *dB, *dC;
  hipMalloc(&dA, A.size() * 4); hipMalloc(&dB, B.size() * 4); hipMalloc(&dC, 4096);
  hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(HIP_KERNEL_NAME(kacc<MODE>), dim3(1), dim3(64), 0, 0, dA, dB, dC, K);
  hipMemcpy(C.data(), dC, 4096, hipMemcpyDeviceToHost);
  double emax = 0, esum = 0, scale = 0, e32max = 0;
  for (int i = 0; i < 32; ++i)
    for (int j = 0; j < 32; ++j) {
      double ref = 0, mag = 0; float f32 = 0.f;
      for (int k = 0; k < K; ++k) { ref += (double)A[i * K + k] * B[j * K + k]; mag += fabs((double)A[i * K + k] * B[j * K + k]); f32 = fmaf(A[i * K + k], B[j * K + k], f32); }
      const double e = fabs(C[i * 32 + j] - ref) / mag;
      emax = fmax(emax, e); esum += e; scale += mag;
      e32max = fmax(e32max, fabs((double)f32 - ref) / mag);
    }
  printf("K=%5d %-34s max |err|/sum|a||b| = %.3e   mean %.3e   (sequential fp32 fma loop on the host: max %.3e)\n", K, what, emax, esum / 1024, e32max);
  hipFree(dA); hipFree(dB); hipFree(dC);
}
int main() {
  run<0, 0>(1); run<0, 0>(2); run<2, 0>(1); run<4, 0>(1); run<8, 0>(1); run<4, 0>(2); run<8, 0>(2);
  run<0, 1>(1); run<0, 2>(1); run<0, 1>(2); run<4, 1>(1); run<4, 1>(2); run<6, 1>(2);
  for (int K : {288, 1152, 4608, 65536}) {
    accuracy<0>(K, "fp32 MFMA 32x32x2");
    accuracy<1>(K, "bf16 x3 terms (2 trunc pieces)");
    accuracy<4>(K, "bf16 x3 terms (2 RN pieces)");
    accuracy<2>(K, "bf16 x6 terms (3 trunc pieces)");
    accuracy<3>(K, "bf16 x6 terms (3 RN pieces)");
  }
  return 0;
}
