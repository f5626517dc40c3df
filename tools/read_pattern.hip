// HBM read rate as a function of the access pattern of a channels-last [pixels][C] fp32 tensor (C = 768: the qkv tensor of the
// Darcy model's 64x64 level, 805 MB at batch 64).  Every variant reads each byte exactly once.
//   rows      : a wave reads whole pixel rows (64 lanes x float4 = 1 KiB contiguous per instruction)
//   slice128  : one wave = 32 pixels; per step one 128-byte head slice of each pixel (lane = 4 B), 24 slices one after the other
//               (what the attention kernels and the streaming weight gradient do)
//   slice128x4: same, float4 per lane (16 lanes per pixel slice... two halves) - the per-pixel backward kernels' pattern
//   planar    : the same 32x128-byte tile when the tensor is stored plane-major [C/32][pixels][32] (4 KiB contiguous per step)
// build+run on the GPU box:  hipcc --offload-arch=gfx950 -O3 tools/read_pattern.hip -o /tmp/read_pattern && /tmp/read_pattern
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
__global__ void __launch_bounds__(256) rows_kernel(const float* x, float* out, size_t n4) {
  f32x4 s = {0, 0, 0, 0};
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) s += reinterpret_cast<const f32x4*>(x)[i];
  if (s[0] + s[1] + s[2] + s[3] == 1234.5f) out[0] = 1.f;
}
// wave w handles pixels [32w, 32w+32); PLANAR: element (p, c) at (c/32)*npix*32 + p*32 + c%32
template <int MODE>
__global__ void __launch_bounds__(256) slice_kernel(const float* x, float* out, size_t npix, int C) {
  const int lane = threadIdx.x & 63, half = lane >> 5, l31 = lane & 31;
  const size_t w = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  const size_t p0 = w * 32;
  if (p0 >= npix) return;
  float s = 0.f;
  const int nsl = C / 32;
  if (MODE == 0) {          // 4 B per lane: lanes 0-31 pixel 2k, lanes 32-63 pixel 2k+1 -> 16 instructions per slice
    for (int h = 0; h < nsl; ++h)
#pragma unroll
      for (int k = 0; k < 16; ++k) s += x[(p0 + 2 * k + half) * C + h * 32 + l31];
  } else if (MODE == 1) {   // float4 per lane: lane = (pixel l31, half) reads 16 floats [16half, 16half+16) of the slice
    for (int h = 0; h < nsl; ++h)
#pragma unroll
      for (int k = 0; k < 4; ++k) { f32x4 v = *reinterpret_cast<const f32x4*>(x + (p0 + l31) * C + h * 32 + 16 * half + 4 * k); s += v[0] + v[1] + v[2] + v[3]; }
  } else {                  // planar: the 32x32 tile is 4 KiB contiguous
    for (int h = 0; h < nsl; ++h)
#pragma unroll
      for (int k = 0; k < 4; ++k) { f32x4 v = *reinterpret_cast<const f32x4*>(x + (size_t)h * npix * 32 + p0 * 32 + (k * 64 + lane) * 4); s += v[0] + v[1] + v[2] + v[3]; }
  }
  if (s == 1234.5f) out[0] = 1.f;
}
// ---- the same question for stores: every variant writes each byte of the [pixels][C] tensor exactly once ----
//   MODE 0 rows     : 1 KiB contiguous per wave instruction (a fill)
//   MODE 1 line128  : one wave = 32 pixels; per instruction two pixels' 128-byte slices (4 B per lane; the MFMA C layout row-major)
//   MODE 2 pix16    : lane = (pixel, half) stores 16 B: 32 B into each of 32 lines per instruction (result "one pixel per lane")
//   MODE 3 seg512   : lane = 16 B of a 512-byte run of one pixel; two pixels per instruction (the 128-channel conv tile epilogue)
template <int MODE>
__global__ void __launch_bounds__(256) store_kernel(float* x, size_t npix, int C) {
  const int lane = threadIdx.x & 63, half = lane >> 5, l31 = lane & 31;
  const size_t w = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  const f32x4 v4 = {1.f, 2.f, 3.f, 4.f};
  if (MODE == 0) {
    const size_t n4 = npix * C / 4;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) reinterpret_cast<f32x4*>(x)[i] = v4;
    return;
  }
  const size_t p0 = w * 32;
  if (p0 >= npix) return;
  if (MODE == 1) {
    for (int h = 0; h < C / 32; ++h)
#pragma unroll
      for (int k = 0; k < 16; ++k) x[(p0 + 2 * k + half) * C + h * 32 + l31] = 1.f;
  } else if (MODE == 2) {
    for (int h = 0; h < C / 32; ++h)
#pragma unroll
      for (int k = 0; k < 4; ++k) *reinterpret_cast<f32x4*>(x + (p0 + l31) * C + h * 32 + 8 * k + 4 * half) = v4;
  } else {
    for (int g = 0; g < C / 128; ++g)
#pragma unroll
      for (int k = 0; k < 16; ++k) *reinterpret_cast<f32x4*>(x + (p0 + 2 * k + half) * C + g * 128 + 4 * l31) = v4;
  }
}
template <class F> float timeit(F f) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  f(); hipDeviceSynchronize();
  hipEventRecord(e0);
  for (int i = 0; i < 5; ++i) f();
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1); return ms / 5;
}
int main() {
  const size_t npix = 64ull * 4096; const int C = 768;
  const size_t n = npix * C;
  float *x, *out; hipMalloc(&x, n * 4); hipMalloc(&out, 4); hipMemset(x, 0, n * 4);
  const double gb = n * 4 / 1e9;
  float t = timeit([&] { hipLaunchKernelGGL(rows_kernel, dim3(256 * 8), dim3(256), 0, 0, x, out, n / 4); });
  printf("rows       %7.1f us  %5.2f TB/s\n", t * 1e3, gb / t);
  const unsigned grid = (unsigned)(npix / 128);
  t = timeit([&] { hipLaunchKernelGGL(HIP_KERNEL_NAME(slice_kernel<0>), dim3(grid), dim3(256), 0, 0, x, out, npix, C); });
  printf("slice128   %7.1f us  %5.2f TB/s\n", t * 1e3, gb / t);
  t = timeit([&] { hipLaunchKernelGGL(HIP_KERNEL_NAME(slice_kernel<1>), dim3(grid), dim3(256), 0, 0, x, out, npix, C); });
  printf("slice128x4 %7.1f us  %5.2f TB/s\n", t * 1e3, gb / t);
  t = timeit([&] { hipLaunchKernelGGL(HIP_KERNEL_NAME(slice_kernel<2>), dim3(grid), dim3(256), 0, 0, x, out, npix, C); });
  printf("planar     %7.1f us  %5.2f TB/s\n", t * 1e3, gb / t);
  t = timeit([&] { hipLaunchKernelGGL(HIP_KERNEL_NAME(store_kernel<0>), dim3(256 * 8), dim3(256), 0, 0, x, npix, C); });
  printf("store rows     %7.1f us  %5.2f TB/s\n", t * 1e3, gb / t);
  t = timeit([&] { hipLaunchKernelGGL(HIP_KERNEL_NAME(store_kernel<1>), dim3(grid), dim3(256), 0, 0, x, npix, C); });
  printf("store line128  %7.1f us  %5.2f TB/s\n", t * 1e3, gb / t);
  t = timeit([&] { hipLaunchKernelGGL(HIP_KERNEL_NAME(store_kernel<2>), dim3(grid), dim3(256), 0, 0, x, npix, C); });
  printf("store pix16    %7.1f us  %5.2f TB/s\n", t * 1e3, gb / t);
  t = timeit([&] { hipLaunchKernelGGL(HIP_KERNEL_NAME(store_kernel<3>), dim3(grid), dim3(256), 0, 0, x, npix, C); });
  printf("store seg512   %7.1f us  %5.2f TB/s\n", t * 1e3, gb / t);
  return 0;
}
