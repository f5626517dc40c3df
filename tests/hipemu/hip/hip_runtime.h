// hipemu: a CPU emulation of the small HIP subset used by physicsinformeddiffusionmodels_amd/csrc.
//
// TEST INFRASTRUCTURE ONLY.  The authoring container has no GPU, and a gpurun round-trip costs
// minutes, so the unmodified gfx950 kernel sources (csrc/*.hip) are additionally compiled for the host
// against THIS header (it shadows <hip/hip_runtime.h> via -I tests/hipemu) and executed with one fiber
// per GPU thread: __syncthreads / wave shuffles / MFMA are emulated with the documented gfx950 lane
// layouts (cdna_hip_programming.md section 3).  It checks index math, LDS addressing, barriers
// placement (deadlock detection), fragment layouts and the host-side orchestration.  It proves nothing
// about performance and it is never loaded by the product package: the product path loads
// libpidm_hip.so only and fails loudly without it.
#pragma once

#include <atomic>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <thread>
#include <vector>

#define __global__
#define __device__
#define __host__
#define __forceinline__ inline __attribute__((always_inline))
#define __launch_bounds__(...)
#define __shared__ static thread_local
#define HIP_DYNAMIC_SHARED(type, var) type* var = reinterpret_cast<type*>(hipemu::dyn_smem());
#define HIP_KERNEL_NAME(...) __VA_ARGS__

typedef int hipError_t;
typedef void* hipStream_t;
typedef void* hipEvent_t;
enum { hipSuccess = 0, hipErrorInvalidValue = 1 };
enum hipMemcpyKind { hipMemcpyHostToHost = 0, hipMemcpyHostToDevice = 1, hipMemcpyDeviceToHost = 2,
                     hipMemcpyDeviceToDevice = 3, hipMemcpyDefault = 4 };

struct dim3 {
  unsigned x, y, z;
  dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};
struct uint3 { unsigned x, y, z; };

typedef float float4 __attribute__((ext_vector_type(4)));
typedef float float2 __attribute__((ext_vector_type(2)));
typedef int int4 __attribute__((ext_vector_type(4)));
typedef int int2 __attribute__((ext_vector_type(2)));
typedef double double2 __attribute__((ext_vector_type(2)));
static inline float4 make_float4(float a, float b, float c, float d) { float4 v = {a, b, c, d}; return v; }
static inline float2 make_float2(float a, float b) { float2 v = {a, b}; return v; }
static inline double2 make_double2(double a, double b) { double2 v = {a, b}; return v; }
static inline int2 make_int2(int a, int b) { int2 v = {a, b}; return v; }

namespace hipemu {
enum State { READY = 0, WAIT_BLOCK = 1, WAIT_WAVE = 2, DONE = 3 };
struct Fiber {
  void* sp;        // saved stack pointer (callee-saved registers live on the fiber's own stack)
  char* stack;
  int state;
};
struct BlockRunner {
  std::vector<Fiber> fibers;
  std::vector<uint64_t> xa, xb;  // cross-lane exchange slots
  std::vector<uint32_t> xw;      // 8 dwords per lane: the 128-bit operand pairs of the bf16 MFMA
  void* sched_sp = nullptr;
  int cur = 0;
  int nthreads = 0;
  char* dyn = nullptr;
  size_t dyn_cap = 0;
  std::function<void()> body;
};
extern thread_local BlockRunner* g_runner;
extern thread_local uint3 g_tid, g_bid;
extern thread_local dim3 g_bdim, g_gdim;
char* dyn_smem();
void yield_state(int st);
void launch(dim3 grid, dim3 block, size_t shmem, const std::function<void()>& body);
inline int lane_id() { return (int)(g_tid.x + g_bdim.x * (g_tid.y + g_bdim.y * g_tid.z)) & 63; }
inline int flat_tid() { return (int)(g_tid.x + g_bdim.x * (g_tid.y + g_bdim.y * g_tid.z)); }
inline void wave_sync() { yield_state(WAIT_WAVE); }
inline uint64_t exchange(uint64_t v, int src_lane) {
  BlockRunner* r = g_runner;
  int t = flat_tid();
  r->xa[t] = v;
  wave_sync();
  int s = (t & ~63) | (src_lane & 63);
  uint64_t out = (s < r->nthreads) ? r->xa[s] : v;
  wave_sync();
  return out;
}
template <typename T>
inline T shfl_any(T v, int src_lane) {
  static_assert(sizeof(T) <= 8, "shfl of >8 bytes");
  uint64_t raw = 0;
  memcpy(&raw, &v, sizeof(T));
  raw = exchange(raw, src_lane);
  T o;
  memcpy(&o, &raw, sizeof(T));
  return o;
}
}  // namespace hipemu

#define threadIdx (hipemu::g_tid)
#define blockIdx (hipemu::g_bid)
#define blockDim (hipemu::g_bdim)
#define gridDim (hipemu::g_gdim)
#define warpSize 64

static inline void __syncthreads() { hipemu::yield_state(hipemu::WAIT_BLOCK); }
static inline void __builtin_amdgcn_s_barrier() { hipemu::yield_state(hipemu::WAIT_BLOCK); }
// s_nop-level scheduling barrier on the GPU (LDS operations of one wave execute in order); here the lanes of a wave are fibers, so
// data exchanged through LDS inside a wave needs a real rendezvous
#define __builtin_amdgcn_wave_barrier() hipemu::wave_sync()
static inline void __threadfence() {}
static inline void __threadfence_block() {}

template <typename T> static inline T __shfl(T v, int src, int width = 64) {
  int l = hipemu::lane_id();
  return hipemu::shfl_any(v, (l & ~(width - 1)) | (src & (width - 1)));
}
template <typename T> static inline T __shfl_xor(T v, int mask, int width = 64) {
  int l = hipemu::lane_id();
  (void)width;
  return hipemu::shfl_any(v, l ^ mask);
}
template <typename T> static inline T __shfl_down(T v, unsigned d, int width = 64) {
  int l = hipemu::lane_id();
  int s = l + (int)d;
  if ((s & ~(width - 1)) != (l & ~(width - 1))) s = l;
  return hipemu::shfl_any(v, s);
}
template <typename T> static inline T __shfl_up(T v, unsigned d, int width = 64) {
  int l = hipemu::lane_id();
  int s = l - (int)d;
  if (s < 0 || (s & ~(width - 1)) != (l & ~(width - 1))) s = l;
  return hipemu::shfl_any(v, s);
}

// wave vote: non-zero when the predicate holds on any lane of the wave
static inline int __any(int pred) {
  int v = pred != 0;
  for (int off = 32; off > 0; off >>= 1) v |= __shfl_xor(v, off);
  return v;
}

// wave ballot: bit l = lane l's predicate (v_cmp into an SGPR pair on the GPU), and the population count of such a mask
static inline unsigned long long __ballot(int pred) {
  unsigned long long v = pred ? (1ull << hipemu::lane_id()) : 0ull;
  for (int off = 32; off > 0; off >>= 1) v |= __shfl_xor(v, off);
  return v;
}
static inline int __popcll(unsigned long long v) { return __builtin_popcountll(v); }

#define PIDM_HAVE_QUAD_XOR 1
static inline float pidm_quad_xor1(float v) { return __shfl_xor(v, 1); }
static inline float pidm_quad_xor2(float v) { return __shfl_xor(v, 2); }

// cycle / real-time counters (trace builds of the kernels only)
static inline unsigned long long hipemu_counter() { return 0ull; }
#define __builtin_amdgcn_s_memrealtime() hipemu_counter()
#define PIDM_HAVE_ROW_SHL 1
static inline float pidm_row_shl4(float v) { const float o = __shfl(v, (hipemu::lane_id() + 4) & 63); return ((hipemu::lane_id() & 15) + 4 < 16) ? o : 0.f; }
static inline float pidm_row_shl8(float v) { const float o = __shfl(v, (hipemu::lane_id() + 8) & 63); return ((hipemu::lane_id() & 15) + 8 < 16) ? o : 0.f; }
static inline float pidm_other_half(float v) { return __shfl_xor(v, 32); }

// ---- MFMA (f32 in / f32 acc), lane layouts per cdna_hip_programming.md section 3 -----------------
typedef float hipemu_f32x16 __attribute__((ext_vector_type(16)));
typedef float hipemu_f32x4 __attribute__((ext_vector_type(4)));
// The matrix instructions ignore the exec mask: a wave that reaches one in divergent control flow computes with whatever the
// inactive lanes hold.  Every lane tags its operand slot with the call-site line; lanes of one wave meeting at different sites
// (e.g. half the wave still in an unrolled main loop, the other half in its tail loop) abort instead of silently pairing up.
static inline void hipemu_mfma_check(hipemu::BlockRunner* r, int wb, int lane, int line) {
  if (lane != 0) return;
  for (int l = 0; l < 64 && wb + l < r->nthreads; ++l)
    if ((int)(r->xa[wb + l] >> 32) != line) {
      fprintf(stderr, "hipemu: MFMA reached in divergent control flow (lane 0 at line %d, lane %d at line %d)\n", line, l,
              (int)(r->xa[wb + l] >> 32));
      abort();
    }
}
static inline hipemu_f32x16 hipemu_mfma_32x32x2f32(float a, float b, hipemu_f32x16 c, int, int, int, int line = __builtin_LINE()) {
  hipemu::BlockRunner* r = hipemu::g_runner;
  int t = hipemu::flat_tid(), lane = t & 63, wb = t & ~63;
  uint32_t ua, ub;
  memcpy(&ua, &a, 4);
  memcpy(&ub, &b, 4);
  r->xa[t] = ua | ((uint64_t)(uint32_t)line << 32);
  r->xb[t] = ub;
  hipemu::wave_sync();
  hipemu_mfma_check(r, wb, lane, line);
  for (int reg = 0; reg < 16; ++reg) {
    int row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5), col = lane & 31;
    float acc = c[reg];
    for (int k = 0; k < 2; ++k) {
      uint32_t xa = (uint32_t)r->xa[wb + k * 32 + row], xb = (uint32_t)r->xb[wb + k * 32 + col];
      float fa, fb;
      memcpy(&fa, &xa, 4);
      memcpy(&fb, &xb, 4);
      acc = fmaf(fa, fb, acc);
    }
    c[reg] = acc;
  }
  hipemu::wave_sync();
  return c;
}
static inline hipemu_f32x4 hipemu_mfma_16x16x4f32(float a, float b, hipemu_f32x4 c, int, int, int, int line = __builtin_LINE()) {
  hipemu::BlockRunner* r = hipemu::g_runner;
  int t = hipemu::flat_tid(), lane = t & 63, wb = t & ~63;
  uint32_t ua, ub;
  memcpy(&ua, &a, 4);
  memcpy(&ub, &b, 4);
  r->xa[t] = ua | ((uint64_t)(uint32_t)line << 32);
  r->xb[t] = ub;
  hipemu::wave_sync();
  hipemu_mfma_check(r, wb, lane, line);
  for (int reg = 0; reg < 4; ++reg) {
    int row = (lane >> 4) * 4 + reg, col = lane & 15;
    float acc = c[reg];
    for (int k = 0; k < 4; ++k) {
      uint32_t xa = (uint32_t)r->xa[wb + k * 16 + row], xb = (uint32_t)r->xb[wb + k * 16 + col];
      float fa, fb;
      memcpy(&fa, &xa, 4);
      memcpy(&fb, &xb, 4);
      acc = fmaf(fa, fb, acc);
    }
    c[reg] = acc;
  }
  hipemu::wave_sync();
  return c;
}
// ---- bf16 operands (pidm_common.h: split form of the fp32 contraction) ---------------------------------------------------------
#define PIDM_HAVE_BF16_OPS 1
static inline float __uint_as_float(unsigned u) { float f; memcpy(&f, &u, 4); return f; }
static inline unsigned __float_as_uint(float f) { unsigned u; memcpy(&u, &f, 4); return u; }
typedef unsigned hipemu_u32x4 __attribute__((ext_vector_type(4)));
static inline unsigned hipemu_bf16_rne(float f) {   // v_cvt_pk_bf16_f32 rounds to nearest even
  unsigned u = __float_as_uint(f);
  if ((u & 0x7fffffffu) > 0x7f800000u) return (u >> 16) | 0x40u;   // NaN stays NaN
  return (u + 0x7fffu + ((u >> 16) & 1u)) >> 16;
}
static inline unsigned pidm_cvt_pk_bf16(float lo, float hi) { return hipemu_bf16_rne(lo) | (hipemu_bf16_rne(hi) << 16); }
// v_mfma_f32_32x32x16_bf16: lane l holds row / column l & 31, k = 8 (l >> 5) + 0..7; D as the fp32 32x32 MFMAs
static inline hipemu_f32x16 pidm_mfma_bf16_32x32x16(hipemu_u32x4 a, hipemu_u32x4 b, hipemu_f32x16 c, int line = __builtin_LINE()) {
  hipemu::BlockRunner* r = hipemu::g_runner;
  int t = hipemu::flat_tid(), lane = t & 63, wb = t & ~63;
  r->xa[t] = (uint64_t)(uint32_t)line << 32;
  for (int i = 0; i < 4; ++i) { r->xw[8 * t + i] = a[i]; r->xw[8 * t + 4 + i] = b[i]; }
  hipemu::wave_sync();
  hipemu_mfma_check(r, wb, lane, line);
  for (int reg = 0; reg < 16; ++reg) {
    int row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5), col = lane & 31;
    // the 16 products of bf16 pairs are exact in fp32; the hardware sums them before ONE rounding into the accumulator
    // (tools/mfma_bf16_probe.hip: the 6-term split form is more accurate than the fp32 MFMA, which a per-product rounding
    // would not allow) - modelled as a double sum rounded once
    double acc = (double)c[reg];
    for (int k = 0; k < 16; ++k) {
      const uint32_t wa = r->xw[8 * (wb + (k >> 3) * 32 + row) + ((k & 7) >> 1)], wbv = r->xw[8 * (wb + (k >> 3) * 32 + col) + 4 + ((k & 7) >> 1)];
      const float fa = __uint_as_float((k & 1) ? (wa & 0xffff0000u) : (wa << 16)), fb = __uint_as_float((k & 1) ? (wbv & 0xffff0000u) : (wbv << 16));
      acc += (double)fa * (double)fb;
    }
    c[reg] = (float)acc;
  }
  hipemu::wave_sync();
  return c;
}
// raw buffer loads (pidm_common.h): base + per-lane offset + uniform offset, zero when the offset leaves the range
#define PIDM_HAVE_BUFLOAD 1
struct pidm_rsrc { const char* base; unsigned bytes; };
static inline pidm_rsrc pidm_make_rsrc(const void* base, unsigned bytes) { return pidm_rsrc{static_cast<const char*>(base), bytes}; }
static inline float pidm_buf_load_f32(pidm_rsrc r, unsigned voff, unsigned soff) {
  const uint64_t o = (uint64_t)voff + soff;
  if (o + 4 > r.bytes) return 0.f;
  float f;
  memcpy(&f, r.base + o, 4);
  return f;
}
#define PIDM_HAVE_BUFLOAD4 1
typedef float hipemu_f32x4 __attribute__((ext_vector_type(4)));
static inline hipemu_f32x4 pidm_buf_load_f32x4(pidm_rsrc r, unsigned voff, unsigned soff) {
  const uint64_t o = (uint64_t)voff + soff;
  hipemu_f32x4 f = {0.f, 0.f, 0.f, 0.f};
  if (voff < r.bytes && o + 16 <= r.bytes) memcpy(&f, r.base + o, 16);   // (the hardware checks the per-lane offset only)
  return f;
}
#define PIDM_WAVES_PER_SIMD(n)
#define PIDM_HAVE_BUFSTORE4 1
typedef unsigned hipemu_u32x4 __attribute__((ext_vector_type(4)));
static inline void pidm_buf_store_u32x4(pidm_rsrc r, unsigned voff, unsigned soff, hipemu_u32x4 v) {
  const uint64_t o = (uint64_t)voff + soff;
  if (voff < r.bytes && o + 16 <= r.bytes) memcpy(const_cast<char*>(r.base) + o, &v, 16);
}
// global_load_lds_dwordx4 (pidm_common.h): synchronous here
#define PIDM_HAVE_GLDS 1
static inline void pidm_glds_b128(const void* gsrc_lane, void* lds_base_uniform) {
  memcpy(static_cast<char*>(lds_base_uniform) + 16 * hipemu::lane_id(), gsrc_lane, 16);
}
static inline void pidm_glds_b128_untracked(const void* gsrc_lane, void* lds_base_uniform) { pidm_glds_b128(gsrc_lane, lds_base_uniform); }
// v_permlane32_swap / v_permlane16_swap (gfx950): rows (16 lanes) 2,3 of the first operand <-> rows 0,1 of the second; odd rows
// of the first <-> even rows of the second.  Returns {new first, new second}.
typedef unsigned hipemu_u32x2 __attribute__((ext_vector_type(2)));
static inline hipemu_u32x2 hipemu_permlane_swap(unsigned a, unsigned b, int dist) {
  const int l = hipemu::lane_id();
  const bool upper = (l & dist) != 0;
  const uint64_t mine = (uint64_t)a | ((uint64_t)b << 32);
  const uint64_t other = hipemu::exchange(mine, l ^ dist);
  hipemu_u32x2 r;
  if (upper) { r[0] = (unsigned)(other >> 32); r[1] = b; }      // first operand's upper part receives the partner's second operand
  else { r[0] = a; r[1] = (unsigned)(other & 0xffffffffu); }    // second operand's lower part receives the partner's first operand
  return r;
}
#define __builtin_amdgcn_permlane32_swap(a, b, fi, bc) hipemu_permlane_swap((a), (b), 32)
#define __builtin_amdgcn_permlane16_swap(a, b, fi, bc) hipemu_permlane_swap((a), (b), 16)
static inline unsigned __builtin_amdgcn_alignbit(unsigned hi, unsigned lo, unsigned sh) {
  return (unsigned)(((((uint64_t)hi) << 32) | lo) >> (sh & 31));
}
// v_readfirstlane_b32: lane 0's value for the whole wave (kernels use it to tell the compiler a value is wave-uniform)
static inline int hipemu_readfirstlane(int v) { return __shfl(v, 0); }
#define __builtin_amdgcn_readfirstlane(v) hipemu_readfirstlane(v)
#define __builtin_amdgcn_mfma_f32_32x32x2f32 hipemu_mfma_32x32x2f32
#define __builtin_amdgcn_mfma_f32_16x16x4f32 hipemu_mfma_16x16x4f32

// ---- atomics / math ---------------------------------------------------------------------------
static inline float atomicAdd(float* p, float v) {
  auto* a = reinterpret_cast<std::atomic<float>*>(p);
  float old = a->load();
  while (!a->compare_exchange_weak(old, old + v)) {}
  return old;
}
static inline int atomicAdd(int* p, int v) { return reinterpret_cast<std::atomic<int>*>(p)->fetch_add(v); }
static inline unsigned atomicAdd(unsigned* p, unsigned v) { return reinterpret_cast<std::atomic<unsigned>*>(p)->fetch_add(v); }
static inline float hipemu_fast_expf(float x) { return exp2f(x * 1.4426950408889634f); }   // v_mul_f32 + v_exp_f32
#define __expf(x) hipemu_fast_expf(x)
static inline float __fdividef(float a, float b) { return a / b; }
#define PIDM_WAIT_VMEM() ((void)0)
#define PIDM_WAIT_VMEM_LEAVE(n_) ((void)0)
#define PIDM_UNTRACKED_LOAD_F32X4(dst_, ptr_, OFF_) \
  memcpy(&(dst_), reinterpret_cast<const char*>(ptr_) + (OFF_), 16)
#define PIDM_HAVE_FAST_SIGMOID 1
static inline float pidm_sigmoid(float v) { return 1.0f / (1.0f + expf(-v)); }
static inline float __frcp_rn(float a) { return 1.0f / a; }
#define PIDM_OPAQUE_F32(x) do { volatile float t__ = (x); (x) = t__; } while (0)   // value barrier (pidm_common.h)
#define PIDM_OPAQUE_I32(x) do { volatile int t__ = (x); (x) = t__; } while (0)
#define PIDM_WAVE_LDS_SYNC() hipemu::wave_sync()
static inline float rsqrtf(float x) { return 1.0f / sqrtf(x); }
static inline float __ldg(const float* p) { return *p; }

// ---- runtime API subset ------------------------------------------------------------------------
static inline hipError_t hipGetLastError() { return hipSuccess; }
static inline hipError_t hipPeekAtLastError() { return hipSuccess; }
static inline const char* hipGetErrorString(hipError_t) { return "hipemu"; }
// ---- stream capture / graphs: while a capture is open every launch, async memset and async copy is RECORDED (not executed),
// exactly like hipStreamBeginCapture; hipGraphLaunch replays the recorded operations in order.  Streams are not modelled beyond
// "null or not" (the null stream cannot be captured, as on the GPU); cross-stream fork / join by events is program order here.
namespace hipemu {
struct Graph { std::vector<std::function<void()>> ops; };
Graph*& capturing_graph();     // hipemu.cpp: the open capture, or nullptr
}
typedef hipemu::Graph* hipGraph_t;
typedef hipemu::Graph* hipGraphExec_t;
typedef void* hipGraphNode_t;
enum hipStreamCaptureMode { hipStreamCaptureModeGlobal = 0, hipStreamCaptureModeThreadLocal, hipStreamCaptureModeRelaxed };
static inline hipError_t hipStreamBeginCapture(hipStream_t s, hipStreamCaptureMode) {
  if (!s || hipemu::capturing_graph()) return hipErrorInvalidValue;
  hipemu::capturing_graph() = new hipemu::Graph();
  return hipSuccess;
}
enum hipStreamCaptureStatus { hipStreamCaptureStatusNone = 0, hipStreamCaptureStatusActive, hipStreamCaptureStatusInvalidated };
static inline hipError_t hipStreamIsCapturing(hipStream_t, hipStreamCaptureStatus* st) {
  *st = hipemu::capturing_graph() ? hipStreamCaptureStatusActive : hipStreamCaptureStatusNone;
  return hipSuccess;
}
static inline hipError_t hipStreamEndCapture(hipStream_t, hipGraph_t* g) {
  if (!hipemu::capturing_graph()) return hipErrorInvalidValue;
  *g = hipemu::capturing_graph();
  hipemu::capturing_graph() = nullptr;
  return hipSuccess;
}
static inline hipError_t hipGraphInstantiate(hipGraphExec_t* e, hipGraph_t g, hipGraphNode_t*, char*, size_t) {
  *e = new hipemu::Graph(*g);
  return hipSuccess;
}
static inline hipError_t hipGraphLaunch(hipGraphExec_t e, hipStream_t) {
  if (!e || hipemu::capturing_graph()) return hipErrorInvalidValue;
  for (auto& op : e->ops) op();
  return hipSuccess;
}
static inline hipError_t hipGraphGetNodes(hipGraph_t g, hipGraphNode_t*, size_t* n) { *n = g ? g->ops.size() : 0; return hipSuccess; }
static inline hipError_t hipGraphDestroy(hipGraph_t g) { delete g; return hipSuccess; }
static inline hipError_t hipGraphExecDestroy(hipGraphExec_t e) { delete e; return hipSuccess; }
static inline hipError_t hipMemsetAsync(void* p, int v, size_t n, hipStream_t) {
  if (hipemu::capturing_graph()) hipemu::capturing_graph()->ops.push_back([=]() { memset(p, v, n); });
  else memset(p, v, n);
  return hipSuccess;
}
static inline hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, hipMemcpyKind, hipStream_t) {
  if (hipemu::capturing_graph()) hipemu::capturing_graph()->ops.push_back([=]() { memcpy(d, s, n); });
  else memcpy(d, s, n);
  return hipSuccess;
}
static inline hipError_t hipMemcpy(void* d, const void* s, size_t n, hipMemcpyKind) { memcpy(d, s, n); return hipSuccess; }
static inline hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
static inline hipError_t hipDeviceSynchronize() { return hipSuccess; }
static inline hipError_t hipMalloc(void** p, size_t n) { *p = malloc(n); return *p ? hipSuccess : hipErrorInvalidValue; }
static inline hipError_t hipFree(void* p) { free(p); return hipSuccess; }

template <typename K, typename... Args>
static inline void hipLaunchKernelGGL(K kernel, dim3 grid, dim3 block, size_t shmem, hipStream_t, Args... args) {
  if (hipemu::capturing_graph())
    hipemu::capturing_graph()->ops.push_back([=]() { hipemu::launch(grid, block, shmem, [=]() { kernel(args...); }); });
  else
    hipemu::launch(grid, block, shmem, [=]() { kernel(args...); });
}

enum hipFuncAttribute { hipFuncAttributeMaxDynamicSharedMemorySize = 8 };
template <typename K> static inline hipError_t hipFuncSetAttribute(K, hipFuncAttribute, int) { return hipSuccess; }

#define hipStreamNonBlocking 1
#define hipEventDisableTiming 2
static inline hipError_t hipStreamCreateWithFlags(hipStream_t* s, unsigned) {
  static uintptr_t next = 0x1000;       // distinct non-null handles (the null stream is special: it cannot be captured)
  *s = reinterpret_cast<hipStream_t>(next += 16);
  return hipSuccess;
}
static inline hipError_t hipStreamDestroy(hipStream_t) { return hipSuccess; }
static inline hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t, unsigned) { return hipSuccess; }   // launches run in program order
static inline hipError_t hipEventCreateWithFlags(hipEvent_t* e, unsigned) { *e = nullptr; return hipSuccess; }
static inline hipError_t hipEventCreate(hipEvent_t* e) { *e = nullptr; return hipSuccess; }
static inline hipError_t hipEventDestroy(hipEvent_t) { return hipSuccess; }
static inline hipError_t hipEventRecord(hipEvent_t, hipStream_t) { return hipSuccess; }
static inline hipError_t hipEventSynchronize(hipEvent_t) { return hipSuccess; }
static inline hipError_t hipEventElapsedTime(float* ms, hipEvent_t, hipEvent_t) { *ms = 0.f; return hipSuccess; }

static inline long long clock64() { return 0; }
#define HIP_SYMBOL(x) x
template <typename T> static inline hipError_t hipMemcpyFromSymbol(void* dst, const T& sym, size_t n) { memcpy(dst, &sym, n); return hipSuccess; }
static inline unsigned __umulhi(unsigned a, unsigned b) { return (unsigned)(((uint64_t)a * (uint64_t)b) >> 32); }

#define __builtin_amdgcn_sched_group_barrier(mask, size, id) ((void)0)
#define __builtin_amdgcn_sched_barrier(mask) ((void)0)
#define __builtin_amdgcn_s_setprio(p) ((void)0)
