// How fast can a CU pull L2-resident bytes into LDS (global_load_lds_dwordx4) or registers (global_load_dwordx4) while every CU does
// the same?  The split-form convolution kernels stream a 31.5 KB pre-split weight slab + a 10-20 KB activation tile per stage and
// workgroup; this probe measures the ceiling of that stream alone (no MFMAs, no LDS reads).
//   hipcc --offload-arch=gfx950 -O3 tools/lds_inbound_probe.hip -o /tmp/lds_inbound_probe && /tmp/lds_inbound_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

#define CK(x) do { hipError_t e__ = (x); if (e__ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e__), __LINE__); exit(1); } } while (0)

static constexpr int kSlab = 32 * 1024;   // bytes per stage

// mode 0: LDS-direct copies; mode 1: register loads (16 B per lane), result xor-ed into a sink
// share: number of distinct slab sequences (1 = all workgroups read the same bytes, 8 = eight groups, 0 = one per workgroup)
template <int MODE>
__global__ void __launch_bounds__(512) probe(const char* __restrict__ src, int nslab, int share, int iters, int depth, unsigned* sink,
                                             unsigned long long* cyc) {
  extern __shared__ char lds[];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), nw = blockDim.x >> 6;
  const int grp = share > 0 ? (int)(blockIdx.x % share) : (int)blockIdx.x;
  const char* base = src + (size_t)grp * nslab * kSlab;
  unsigned acc = 0;
  const unsigned long long t0 = clock64();
  for (int it = 0; it < iters; ++it) {
    const char* s = base + (size_t)(it % nslab) * kSlab;
    char* d = lds + (it & 1) * kSlab;
    // a slab = 32 pieces of 1 KB; wave w takes pieces w, w + nw, ...
    for (int p = wave; p < 32; p += nw) {
      if (MODE == 0) {
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(s + 1024 * p + 16 * lane),
                                         (__attribute__((address_space(3))) void*)(d + 1024 * p), 16, 0, 0);
      } else {
        const uint4 v = *reinterpret_cast<const uint4*>(s + 1024 * p + 16 * lane);
        acc ^= v.x ^ v.y ^ v.z ^ v.w;
      }
    }
    if (depth == 0 || (it % depth) == depth - 1) {
      __builtin_amdgcn_s_waitcnt(0x0F70);
      __syncthreads();
    }
  }
  __builtin_amdgcn_s_waitcnt(0x0F70);
  __syncthreads();
  const unsigned long long t1 = clock64();
  if (MODE == 1 && acc == 0x12345678u) sink[blockIdx.x] = acc;
  if (tid == 0) cyc[blockIdx.x] = t1 - t0;
}

int main() {
  const int nwg = 256, nslab = 16, iters = 512;
  const size_t total = (size_t)nwg * nslab * kSlab;   // 128 MB when every workgroup has its own sequence
  char* src; unsigned* sink; unsigned long long* cyc;
  CK(hipMalloc(&src, total)); CK(hipMemset(src, 1, total));
  CK(hipMalloc(&sink, nwg * 4)); CK(hipMalloc(&cyc, nwg * 8));
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&probe<0>), hipFuncAttributeMaxDynamicSharedMemorySize, 2 * kSlab));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  printf("%-10s %-8s %-6s %-6s %10s %12s %14s\n", "mode", "share", "waves", "depth", "us", "TB/s chip", "B/clk/CU (wg0)");
  for (int mode = 0; mode < 2; ++mode)
    for (int share : {1, 8, 32, 0})
      for (int nw : {4, 8})
        for (int depth : {1, 2}) {
          for (int rep = 0; rep < 2; ++rep) {
            CK(hipEventRecord(e0));
            if (mode == 0) hipLaunchKernelGGL(probe<0>, dim3(nwg), dim3(64 * nw), 2 * kSlab, 0, src, nslab, share, iters, depth, sink, cyc);
            else hipLaunchKernelGGL(probe<1>, dim3(nwg), dim3(64 * nw), 2 * kSlab, 0, src, nslab, share, iters, depth, sink, cyc);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
          }
          float ms; CK(hipEventElapsedTime(&ms, e0, e1));
          std::vector<unsigned long long> h(nwg);
          CK(hipMemcpy(h.data(), cyc, nwg * 8, hipMemcpyDeviceToHost));
          const double bytes = (double)nwg * iters * kSlab;
          printf("%-10s %-8d %-6d %-6d %10.1f %12.2f %14.1f\n", mode == 0 ? "lds-dma" : "registers", share, nw, depth, ms * 1e3, bytes / (ms * 1e-3) / 1e12,
                 (double)iters * kSlab / (double)h[0]);
        }
  return 0;
}
