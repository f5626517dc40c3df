#include <hip/hip_runtime.h>
#include <cstdio>
template <int OP>
__global__ void __launch_bounds__(256) k(float* out, int iters) {
  float v[8]; unsigned u[8];
  for (int i = 0; i < 8; ++i) { v[i] = threadIdx.x + i; u[i] = threadIdx.x * 7 + i; }
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 8; ++r) {
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        if (OP == 0) asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(v[i]));
        if (OP == 1) asm volatile("v_cvt_pk_bf16_f32 %0, %1, %1" : "=v"(u[i]) : "v"(v[i]));
        if (OP == 2) asm volatile("v_pk_add_f32 %0, %0, %0" : "+v"(*(double*)&v[i & 6]));
        if (OP == 3) asm volatile("v_and_b32 %0, 0xffff0000, %0" : "+v"(u[i]));
        if (OP == 4) asm volatile("v_lshlrev_b32 %0, 16, %0" : "+v"(u[i]));
        if (OP == 5) asm volatile("v_pk_mul_f32 %0, %0, %0" : "+v"(*(double*)&v[i & 6]));
      }
    }
  }
  float s = 0; for (int i = 0; i < 8; ++i) s += v[i] + u[i];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int OP> void run(const char* n) {
  float* o; hipMalloc(&o, 256 * 256 * 4 * 4);
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  hipLaunchKernelGGL(k<OP>, dim3(256), dim3(256), 0, 0, o, 10);
  hipDeviceSynchronize(); hipEventRecord(a);
  const int iters = 20000;
  hipLaunchKernelGGL(k<OP>, dim3(256), dim3(256), 0, 0, o, iters);
  hipEventRecord(b); hipEventSynchronize(b); float ms; hipEventElapsedTime(&ms, a, b);
  printf("%-22s %.2f cycles@2.4GHz per wave-instruction (1 wave per SIMD)\n", n, ms * 1e-3 * 2.4e9 / (iters * 64.0));
}
int main() { run<0>("v_fma_f32"); run<1>("v_cvt_pk_bf16_f32"); run<2>("v_pk_add_f32"); run<3>("v_and_b32"); run<4>("v_lshlrev_b32"); run<5>("v_pk_mul_f32"); }
