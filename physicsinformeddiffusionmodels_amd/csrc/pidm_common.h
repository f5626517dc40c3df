// Common host/device declarations for the gfx950 engine (libpidm_hip.so).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>
#include <string.h>

#include "../../include/pidm.h"

namespace pidm {

// thread-local error string behind pidm_last_error()
void set_error(const char* fmt, ...);
int fail(const char* fmt, ...);  // sets the error, returns -1

// bench-only per-launch timing (pidm_api.cpp).  classes: 0 conv fwd/dgrad (flops), 1 conv wgrad (flops)
bool prof_enabled();
void prof_set_label(const char* label);   // attached to the next prof_begin_launch records (per-shape tables)
void prof_begin_launch(int cls, double work, hipStream_t st);
void prof_end_launch(hipStream_t st);
void prof_reclass_last(int cls);          // the launcher learned which kernel took the launch (2 / 3 = split form of class 0 / 1)
void prof_cancel_last();                  // nothing was launched after all (the problem went to a grouped launch's queue)
// per-KERNEL timing (pidm_prof_kernels_begin / _collect, round 5): while on, every PIDM_CHECK_LAUNCH records one event on the
// registered stream; a launch's time = the interval since the previous library launch's event (launches are back to back on one
// stream: graphs and the side-stream overlap are off while any profiling hook is on).  PIDM_PROF_NAME names the kernel a shared
// launch site is about to enqueue (a family's launch sites end in ONE PIDM_CHECK_LAUNCH).
extern bool g_prof_marks;
extern const char* g_prof_name;
void prof_mark(const char* what);
// (written only while the per-kernel timing is on: with it off a launch touches no process-global state; the timing pass itself is
// single-threaded by contract - pidm_prof_kernels_begin / _collect bracket launches of ONE host thread on one stream)
#define PIDM_PROF_NAME(n) (::pidm::g_prof_marks ? (void)(::pidm::g_prof_name = (n)) : (void)0)

// Tuning / A-B knobs (the PIDM_* environment variables listed in DESIGN.md section 4): read from the environment ONCE per process
// and name - `knob("PIDM_X")` returns what getenv returned the first time it was asked (or null) - until pidm_reload_knobs()
// (include/pidm.h; tests and A/B scripts that change a variable inside a live process call it).  knob_signature() identifies
// the current snapshot of all PIDM_* variables (part of the hipGraph key: a replayed graph has its knobs frozen in).
const char* knob(const char* name);
uint64_t knob_signature();

inline hipStream_t as_stream(void* s) { return reinterpret_cast<hipStream_t>(s); }

// kernels enqueued through the library since it was loaded: directly (eager) + recorded into graphs during a capture; the engine
// subtracts what a capture recorded (pidm_debug_launch_counts)
extern long long g_kernel_enqueues;

#define PIDM_CHECK_LAUNCH(what)                                                     \
  do {                                                                              \
    ++::pidm::g_kernel_enqueues;                                                    \
    hipError_t e__ = hipGetLastError();                                             \
    if (e__ != hipSuccess) return ::pidm::fail("%s: %s", what, hipGetErrorString(e__)); \
    if (::pidm::g_prof_marks) ::pidm::prof_mark(what);                              \
  } while (0)

// exact n / d for small operands via one mulhi (d == 1 handled by the caller's magic == 0 convention)
__device__ __forceinline__ int fast_div(int n, int d, unsigned magic) { return d == 1 ? n : (int)__umulhi((unsigned)n, magic); }

// value barrier: the compiler may not look through it (used where a product must be ROUNDED before it is added - no fma).
// The host emulator's shadow <hip/hip_runtime.h> provides its own definition.
#ifndef PIDM_OPAQUE_F32
#define PIDM_OPAQUE_F32(x) asm volatile("" : "+v"(x))
#define PIDM_OPAQUE_I32(x) asm volatile("" : "+v"(x))   // e.g. a zero the optimiser cannot see: keeps loop-invariant loads inside the loop
#endif

// Data handed between the lanes of ONE wave through LDS: the hardware executes a wave's LDS operations in order, so only the
// compiler has to be kept from reordering the accesses (the host emulator runs lanes as fibers and needs a real rendezvous).
#ifndef PIDM_WAVE_LDS_SYNC
#define PIDM_WAVE_LDS_SYNC() asm volatile("" ::: "memory")
#endif

// value of the lane whose index differs in bit 0 / bit 1 (within a group of four lanes): one DPP move on the GPU; the host
// emulator's shadow header supplies shuffle-based versions
#ifndef PIDM_HAVE_QUAD_XOR
__device__ __forceinline__ float pidm_quad_xor1(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1 /* quad_perm [1,0,3,2] */, 0xF, 0xF, true));
}
__device__ __forceinline__ float pidm_quad_xor2(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E /* quad_perm [2,3,0,1] */, 0xF, 0xF, true));
}
#endif

// value of lane + 4 / lane + 8 inside the lane's row of 16 (0 past the row's end; one DPP move: row_shl), and of lane ^ 32
// (v_permlane32_swap): sums over groups of 8 / 16 lanes and over the two halves of a wave without a trip through LDS
#ifndef PIDM_HAVE_ROW_SHL
__device__ __forceinline__ float pidm_row_shl4(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x104 /* row_shl:4 */, 0xF, 0xF, true));
}
__device__ __forceinline__ float pidm_row_shl8(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x108 /* row_shl:8 */, 0xF, 0xF, true));
}
__device__ __forceinline__ float pidm_other_half(float v) {
  const unsigned u = __builtin_bit_cast(unsigned, v);
  const auto sw = __builtin_amdgcn_permlane32_swap(u, u, false, false);
  return __builtin_bit_cast(float, (__lane_id() < 32) ? sw[1] : sw[0]);
}
#endif

// logistic function on the hardware's exp2 and reciprocal units (v_exp_f32, v_rcp_f32: ~1 ulp each; relative error of the result
// <= 2e-7 + |v| * 6e-8 * min(1, e^v)) instead of libm expf + a correctly rounded division (~5x the VALU instructions, which made
// the GroupNorm kernels - one or two SiLUs per element - and the dgrad epilogue sums instruction-bound rather than HBM-bound).
// The host emulator's shadow header supplies its own.
#ifndef PIDM_HAVE_FAST_SIGMOID
__device__ __forceinline__ float pidm_sigmoid(float v) { return __builtin_amdgcn_rcpf(1.f + __expf(-v)); }
#endif

// ---- fp32 contractions on the bf16 matrix pipe ("split" form) ---------------------------------------------------------------
// x = p0 + p1 + p2 with three round-to-nearest bf16 pieces (p0 = bf16(x), p1 = bf16(x - p0), p2 = bf16(x - p0 - p1): 24 mantissa
// bits, the remainders are exact in fp32), and a*b ~ a0 b0 + a0 b1 + a1 b0 + a0 b2 + a1 b1 + a2 b0 (the dropped terms are below
// 2^-24 |a||b|), accumulated in fp32 by v_mfma_f32_32x32x16_bf16.  Measured against a float64 product
// (tools/mfma_bf16_probe.hip, profiles/r02_bf16_split_probe.txt): max error 1.1-1.4e-7 of sum|a||b| for K = 288 ... 65536, the
// fp32 MFMA itself: 1.6-2.5e-7.  6 bf16 MFMAs (8 passes each) replace 8 fp32 MFMAs (16 passes each) and - unlike the fp32
// MFMA, which runs on the SIMD's vector ALUs - leave the VALU to the other wave of the SIMD.
// Operand fragment of the 32x32x16 MFMA: lane l holds row (A) / column (B) l & 31, k = 8 (l >> 5) + 0..7, as 4 dwords of bf16 pairs.
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
#ifndef PIDM_HAVE_BF16_OPS
typedef __bf16 pidm_bf16x2 __attribute__((ext_vector_type(2)));
typedef __bf16 pidm_bf16x8 __attribute__((ext_vector_type(8)));
typedef float pidm_f32x2 __attribute__((ext_vector_type(2)));
typedef float pidm_f32x16 __attribute__((ext_vector_type(16)));
// (lo, hi) -> bf16 pair, round to nearest even: one v_cvt_pk_bf16_f32
__device__ __forceinline__ unsigned pidm_cvt_pk_bf16(float lo, float hi) {
  const pidm_f32x2 v = {lo, hi};
  return __builtin_bit_cast(unsigned, __builtin_convertvector(v, pidm_bf16x2));
}
__device__ __forceinline__ pidm_f32x16 pidm_mfma_bf16_32x32x16(u32x4 a, u32x4 b, pidm_f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(pidm_bf16x8, a), __builtin_bit_cast(pidm_bf16x8, b), c, 0, 0, 0);
}
#endif
// Asynchronous global -> LDS copy of 16 bytes per lane (global_load_lds_dwordx4): the LDS destination is wave-uniform base +
// 16 * lane, the global source is per lane; completion is counted by vmcnt (a following __syncthreads() drains it).
// wait until every vector-memory operation of this wave has completed (s_waitcnt vmcnt(0); gfx9 encoding: vmcnt = 0, expcnt and
// lgkmcnt at their maxima) - global_load_lds writes LDS behind the compiler's back, a barrier that publishes such data says so
#ifndef PIDM_WAIT_VMEM
#define PIDM_WAIT_VMEM() __builtin_amdgcn_s_waitcnt(0x0F70)
// ... all but the youngest n_ (a compile-time constant < 64; vmcnt = n_: low four bits + bits 15:14).  Vector-memory loads return in
// issue order, so the n_ loads issued LAST may stay in flight: a producer wave waits for the LDS-direct copies that the barrier
// publishes without also waiting for the register prefetch of a later stage it issued behind them (round 6).  The issue order the
// count relies on is pinned with __builtin_amdgcn_sched_barrier(0) at the call sites.
#define PIDM_WAIT_VMEM_LEAVE(n_) __builtin_amdgcn_s_waitcnt(0x0F70 | ((n_) & 15) | ((((n_) >> 4) & 3) << 14))
// A 16-byte global load the COMPILER DOES NOT TRACK (dst_ = *(f32x4*)((char*)ptr_ + OFF_), OFF_ a literal): hipcc counts the
// LDS-direct copies and the plain loads of a wave as returning out of order and answers every use of a loaded register that has a
// global_load_lds behind it with s_waitcnt vmcnt(0) - the producer waves of the split-form kernels then wait for the prefetch they
// have just issued (ISA of rounds 3-5: a full memory latency in front of the staging arithmetic of every other stage).  The
// hardware returns loads in issue order, so these kernels count by hand: PIDM_WAIT_VMEM_LEAVE(n) + a sched_barrier before the
// first use of dst_.  "memory" pins the issue order against the surrounding copies.
#define PIDM_UNTRACKED_LOAD_F32X4(dst_, ptr_, OFF_) \
  asm volatile("global_load_dwordx4 %0, %1, off offset:" #OFF_ : "=v"(dst_) : "v"(ptr_) : "memory")
#endif
#ifndef PIDM_HAVE_GLDS
__device__ __forceinline__ void pidm_glds_b128(const void* gsrc_lane, void* lds_base_uniform) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc_lane,
                                   (__attribute__((address_space(3))) void*)lds_base_uniform, 16, 0, 0);
}
// The same copy as an instruction the compiler does not track (see PIDM_UNTRACKED_LOAD_F32X4: a wave that counts its vector-memory
// operations by hand must hide ALL of them from hipcc's own vmcnt bookkeeping, which would otherwise add s_waitcnt vmcnt(0) in front
// of the barrier for the copies it knows about).  M0 = LDS base of the wave's 1 KB piece; one wait state between the write of M0 and
// its use, as the compiler emits it.
__device__ __forceinline__ void pidm_glds_b128_untracked(const void* gsrc_lane, void* lds_base_uniform) {
  const unsigned la = (unsigned)(size_t)((__attribute__((address_space(3))) char*)lds_base_uniform);
  asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" : : "v"(gsrc_lane), "s"(la) : "memory", "m0");
}
#endif
// Raw buffer loads (buffer_load_dword through a 128-bit resource descriptor): per-lane 32-bit byte offset + a wave-uniform
// (SGPR) byte offset, hardware range check - an offset at or beyond `bytes` returns 0 without touching memory, which is how the
// row-streaming weight gradient reads its zero padding (k_wgrad_rs.hip).  The host emulator's shadow header supplies its own.
#ifndef PIDM_HAVE_BUFLOAD
typedef __amdgpu_buffer_rsrc_t pidm_rsrc;
__device__ __forceinline__ pidm_rsrc pidm_make_rsrc(const void* base, unsigned bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, (int)bytes, 0x00020000);
}
__device__ __forceinline__ float pidm_buf_load_f32(pidm_rsrc r, unsigned voff, unsigned soff) {
  return __uint_as_float((unsigned)__builtin_amdgcn_raw_buffer_load_b32(r, (int)voff, (int)soff, 0));
}
#endif
// the three pieces of two floats, as bf16 pairs (element 0 in the low half)
__device__ __forceinline__ void pidm_split3_pk(float x0, float x1, unsigned& p0, unsigned& p1, unsigned& p2) {
  p0 = pidm_cvt_pk_bf16(x0, x1);
  float r0 = x0 - __uint_as_float(p0 << 16), r1 = x1 - __uint_as_float(p0 & 0xffff0000u);
  p1 = pidm_cvt_pk_bf16(r0, r1);
  r0 -= __uint_as_float(p1 << 16);
  r1 -= __uint_as_float(p1 & 0xffff0000u);
  p2 = pidm_cvt_pk_bf16(r0, r1);
}

static inline int cdiv(int a, int b) { return (a + b - 1) / b; }
static inline size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
#ifndef PIDM_HAVE_BUFLOAD4
// 16 bytes per lane through the same descriptor (buffer_load_dwordx4; offsets multiples of 16)
__device__ __forceinline__ f32x4 pidm_buf_load_f32x4(pidm_rsrc r, unsigned voff, unsigned soff) {
  return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, (int)voff, (int)soff, 0));
}
#endif
#ifndef PIDM_HAVE_BUFSTORE4
// 16-byte store through a descriptor: a per-lane offset at or beyond the descriptor's size drops the lane's store (predication
// without a branch; a descriptor of size 0 drops them all)
__device__ __forceinline__ void pidm_buf_store_u32x4(pidm_rsrc r, unsigned voff, unsigned soff, u32x4 v) {
  __builtin_amdgcn_raw_buffer_store_b128(v, r, (int)voff, (int)soff, 0);
}
#endif
// register budget of a kernel as waves per SIMD (512 unified registers / n); the host emulator ignores it
#ifndef PIDM_WAVES_PER_SIMD
#define PIDM_WAVES_PER_SIMD(n) __attribute__((amdgpu_waves_per_eu(n, n)))
#endif

// ---------------------------------------------------------------------------------------------
// Geometry of one implicit-GEMM convolution launch (see k_conv.hip).  "Virtual" output pixels
// (vy,vx) in [0,Hv)x[0,Wv) map to real output (vy*os+ooy, vx*os+oox) and read input rows
// vy*stride + ky - pad.  A transposed 4x4/s2/p1 convolution is 4 such problems (one per output
// parity), selected by blockIdx.z.
// ---------------------------------------------------------------------------------------------
struct ConvGeom {
  int B, Hi, Wi;     // input spatial
  int Hv, Wv;        // virtual output grid
  int Ho, Wo;        // real output spatial
  int C0, C1;        // channels from src0 / src1
  int ld0, ld1;      // channel strides of src0/src1
  int Cin;           // C0 + C1
  int Cout;
  int KH, KW;        // taps actually iterated (2x2 for a parity class of the transposed conv)
  int stride;        // input stride
  int os;            // output stride (2 for transposed)
  int nz;            // 1, or 4 parity classes
  // a 4x4 / stride-2 / pad-1 convolution runs as 4 K-phases of 2x2 stride-1 convolutions over the parity sub-images of
  // the input (space-to-depth without moving data): phase ph reads source pixel ((sy)*in_step + ph_oy, (sx)*in_step + ph_ox)
  int nph, in_step;  // 1,1 for ordinary convolutions; 4,2 for the phased 4x4s2
  int Kw;            // K columns of one packed weight row = nph * Cin
  int ph_oy[4], ph_ox[4], ph_pad_y[4], ph_pad_x[4];
  int pad_y[4], pad_x[4], ooy[4], oox[4];
  long w_off[4];     // float offset of the packed weight slab per z
  // output addressing: out[b*sob + oy*soy + ox*sox + c*soc]
  long sob, soy, sox, soc;
  int ldr;           // channel stride of the (channels-last) residual operand
  // tiling
  int TH, NI;        // rows per tile, images per tile: TH*NI*Wv == 128
  int wsh, tsh;      // log2(Wv), log2(TH): pixel p of a tile -> tx = p & (Wv-1), ty = (p >> wsh) & (TH-1), img = p >> (wsh+tsh)
  int IHt, IWt;      // input halo tile extent per image
  unsigned mIWt, mIHt;  // ceil(2^32 / IWt), ceil(2^32 / IHt): n / d == __umulhi(n, m) for n*d < 2^32 (d > 1)
  int tiles_m;       // number of 128-pixel tiles
  int tpw;           // persistent conv kernels: consecutive m-tiles walked by one workgroup (0/1: one tile per workgroup)
  int rpad;          // split-form 3x3 kernels: extra bytes per halo-tile row in LDS (bank spreading at the 8- / 16-wide levels)
  int xcd;           // split-form kernels: workgroup -> work-item order that follows the XCDs (split_vblock, k_conv.hip); 0: off
  // GroupNorm statistics of the OUTPUT from the epilogue (k_norm.hip consumes them): per (image, 32-pixel wave chunk, group)
  // sum and sum of squares as doubles at gn_part[((b*gn_nchunk + chunk)*gn_G + g)*2]; gn_part == null: off.
  // Requires Cout % 32 == 0, gn_cpg = Cout/gn_G a power of two in [4, 32], Ho*Wo % 32 == 0, channels-last output, no residual.
  double* gn_part;
  int gn_cpg, gn_G, gn_nchunk;
  // backward twin: a dgrad convolution whose OUTPUT is the gradient dy wrt y = SiLU(FiLM(GroupNorm(x))) also leaves the sums the
  // GroupNorm backward needs, per (image, 32-pixel chunk, channel): S1 = sum dv, S2 = sum dv*xhat with dv = dy*SiLU'(v), at
  // bn_part[((b*bn_nchunk + chunk)*Cout + c)*2] (doubles) - the first of the two passes of launch_gn_bwd.  bn_part == null: off.
  const float* bn_x;      // the normalised tensor x [B][Ho*Wo][Cout], channels-last
  const float* bn_stats;  // [B][bn_G][2] (mean, rstd)
  const float* bn_gamma;
  const float* bn_beta;
  const float* bn_ss;     // FiLM activations [B][bn_ldss] (scale | shift), may be null
  const float* bn_ssb;    // their bias [2*Cout]
  double* bn_part;
  int bn_ldss, bn_cpg, bn_G, bn_nchunk;
  int bn_res;             // 1: dy = result + residual (the sums are those of the tensor the launch WRITES; split-form 3x3 epilogues only)
  // host side only: where the launcher reports how many partial chunks per image the kernel it picked writes to gn_part / bn_part
  // (Ho*Wo/32 for the tile kernels - one per wave tile; one per strip for the row-streaming kernel, k_conv_rs.hip, which sums over the
  // rows of a strip first).  A caller that passes null gets Ho*Wo/32 chunks or no epilogue.
  int* part_chunks_out;
};

// Geometry of one weight-gradient launch (k_conv.hip, k_wgrad_rs.hip): dW[m][t][n] = sum_p dY[p][m] * X[p (+tap)][n]
// (M = rows of dY's channels, N = X's channels), deterministic split-K over pixel ranges into per-split partial slabs.
struct WgradGeom {
  ConvGeom g;        // pixel tiling / halo geometry of the forward problem (Cout = channels of dY)
  int ld_dy;         // channel stride of dY
  int tgs, ntg;      // taps per group, number of groups
  int nsplit, tiles_per_split;
  int MP, NP;        // padded (to 32) rows / cols of the partial buffer
  // row-streaming kernel (k_wgrad_rs.hip): rows per strip chunk (power of two), log2 of chunks per image and of 8-pixel
  // column strips per row, strips in total, strip pairs per wave
  int rs_R, rs_csh, rs_xsh, rs_S, rs_ppw;
};

// one weight-gradient problem of a GROUPED launch (k_wgrad_rs.hip: conv_wgrad_rs_multi_kernel, round 5): what the single-problem
// kernel receives as arguments, plus the problem's range of workgroups in the flat grid (blk0 .. blk0 + gx * gy)
// Only the fields the grouped kernels read (no host pointers, no padding: tables of these are compared with memcmp).  Three kernel
// families, one table and one launch each: the 3x3 / stride-1 row-streaming kernel, its 4x4 / stride-2 sibling (k_wgrad_rs.hip)
// and the three 1x1 pixel-stream kernels (k_conv_wgrad.hip; `kind` says which).
enum { kWgFamRs = 0, kWgFamRs4 = 1, kWgFam1x1 = 2, kWgFam1x1Split = 3, kWgFams = 4 };   // (1x1Split: the wide-operand 1x1 problems on the bf16 pipe, round 6)
enum { kWgKindStream = 0, kWgKindStream4Dy = 1, kWgKindStream4X = 2 };
struct WgradItem {
  const float* src0;
  const float* src1;
  const float* dy;
  float* partial;
  float* bias_partial;
  int B, Hi, Wi, wsh;      // images; input rows, pixels per row; log2 of the OUTPUT row length (rs: = Wi)
  int Hv, Wv;              // output grid (rs4: Hi / 2, Wi / 2; 1x1: the pixel grid)
  int ld0, ld1, C0;        // channel strides of the two X sources, channels of src0
  int Cin, Cout;           // channels of X / of dY
  int ld_dy;               // channel stride of dY
  int MP, NP;              // padded rows / columns of the partial slab
  int tiles_per_split;     // 1x1: 128-pixel tiles per split
  int rs_R, rs_csh, rs_xsh, rs_S, rs_ppw;
  int kind;                // 1x1 family: kWgKind*
  unsigned blk0, gx, gy;   // first workgroup of the problem in the grouped grid; the problem's own grid is gx x gy
};

// one tensor of the multi-tensor weight re-pack (k_conv.hip: pack_multi_kernel)
struct PackDesc {
  const float* src;
  float* dst;
  int kind, nz, N, K, Np, Kp, KH, KW, T, n_off, k_off;
  unsigned blk0, nblk;
  unsigned short* split;   // 3x3 / stride-1 and 4x4 / stride-2 tensors also leave their bf16 pieces for conv3x3_split_kernel (null: none)
  int nch;                 // 16-channel chunks of a packed row (Kp / 16)
  int ntn;                 // 32-row tiles of one parity slab (Cout / 32)
  int tiled;               // 1: pack_multi_kernel's LDS-transposed 32 x 32 x taps form (k_conv.hip: pack_tile)
  int ntg;                 // 32-row tiles per group in the piece layout (1; 4 / 2 for the 1x1 tensors conv1x1_split_kernel takes)
};

// one deferred fixed-order reduction (k_conv.hip: reduce_multi_kernel), queued during backward and run in ONE launch:
//   dst[(m*N + n)*T + t] = sum_{s < nsplit} src[s*sstride + (m*T + t)*NP + n]     (m < M, n < N, t < T)
//   bdst[m]              = sum_{s < nsplit} bsrc[s*MP + m]                         (optional)
struct ReduceDesc {
  const float* src;
  float* dst;
  const float* bsrc;
  float* bdst;
  size_t sstride;
  int nsplit, M, N, T, MP, NP;
  unsigned blk0, nblk;
  int mode;   // 2: 128 consecutive floats of the flat slab as 16-byte pieces x 8 split lanes per block (many splits of a large slab);
              // 0: 32 outputs x 8 split lanes per block (many small splits); 1: one row m x 64 columns x all taps per block,
              //    transposed through LDS so that both the partial rows and the [m][n][t] result move as whole lines (large tensors
              //    with few splits: the 130 M-parameter mechanics model spent 4.2 ms per step in the scattered form)
};

}  // namespace pidm
