// Convolutions of the UNet on the matrix cores of gfx950: geometry, weight re-packing, the split-form (bf16 pipe, fp32-faithful)
// forward / input-gradient kernels and the launcher that picks a kernel per layer.
//
// Replaces every `convolution` / `conv_transpose` / `addmm` ATen op of the reference UNet
// (src/unet_model.py:163,197,227,253,275,279,453,517 and the nn.Linear layers :248,332,339,466,468):
// 3x3 p1, 7x7 p3, 1x1, 4x4 s2 p1, ConvTranspose 4x4 s2 p1, and Linear (as a 1x1 conv on a 1x1 image).
//
// Data layout: activations channels-last (NHWC == the reference's [B, P*P, C] interchange layout), weights
// re-packed once per step to [Cout_p][tap][Cin_p] (K contiguous) + their bf16 pieces by `pack_multi_kernel`.
// GEMM view: M = output pixels, N = Cout, K = taps x Cin.  Backward: dgrad runs the SAME kernels on a flipped / transposed weight
// packing; the weight gradients are their own kernels (k_wgrad_rs.hip, k_conv_wgrad.hip).  The fp32-MFMA implicit-GEMM design the
// split-form kernels descend from (and fall back to) is described at the top of k_conv_fp32.hip.
//
// Kernels in this file (round 4: the fp32-MFMA forward kernels moved to k_conv_fp32.hip, every weight gradient and the deferred
// reductions to k_conv_wgrad.hip / k_wgrad_rs.hip):
//   conv3x3_split_kernel<NW,MODE> / conv3x3_split_ws_kernel<NW,MODE,NPW>   3x3 and 4x4/s2 forward / dgrad on the bf16 pipe (split form)
//   conv1x1_split_kernel<NTG,NTGP>, conv7x7_split_kernel<CIN>          compute-bound 1x1 convolutions as a GEMM, the 7x7 init conv
//   pack_kernel / pack_multi_kernel                reference weight layouts -> [Cout_p][tap][K_p] + bf16 pieces (all tensors in one launch)
//   make_geom / launch_conv / the C ABI of the convolutions
#include <stdio.h>
#include <stdlib.h>

#include "pidm_launch.h"
#include "k_conv_epilogue.h"

namespace pidm {

static const int kBM = 128;  // pixels per workgroup tile

// ---------------------------------------------------------------------------------------------------
// The same convolutions on the bf16 matrix pipe, fp32-faithful ("split" form, pidm_common.h).
//
// The fp32 MFMA tops out at ~143 TFLOP/s in practice and shares the vector ALUs with everything else a wave does
// (tools/mfma_overlap.hip); 6 bf16 MFMAs on round-to-nearest 3-piece operands give the same product to within the fp32 MFMA's own
// rounding error in 2.3x less matrix-pipe time (tools/mfma_bf16_probe.hip), and the VALU work of ONE wave can run under the MFMAs
// of the OTHER wave of its SIMD (in this kernel about 40 % of it does: tools/split_ablate.py).  So: 8 waves (two per SIMD), a
// 256-pixel x 32-channel tile (one 32x32 accumulator pair per wave), K chunks of 16 channels x 9 taps = 54 MFMAs per wave and
// stage.  Activations are split on their way from global memory into LDS (36 VALU per 8 channels); the weights arrive pre-split
// from the weight re-pack (pack_kernel: split_store), one contiguous 31.5 KB slab per (32 output channels, 16 input channels)
// that is an image of its LDS rows and is copied by global_load_lds_dwordx4.
// LDS row (pixel of the halo tile, or weight row of a tap) = 112 bytes = [half h: piece 0 | piece 1 | piece 2][h = 1: ...][16 pad],
// 16 bytes = 8 channels of one piece = one MFMA operand; 7 slots per row is odd, so the 16 lanes ds_read_b128 serves per cycle
// (consecutive pixels / output channels) hit 16 different 16-byte bank groups.
// The pipeline is the one of conv3x3_stream_kernel: stage s computes from buffer s&1 while the activation registers loaded
// during stage s-1 go to buffer (s+1)&1 and are re-loaded for stage s+2; one barrier per stage; branch-free loads.
// ---------------------------------------------------------------------------------------------------
// cycle stamps of workgroup 0 / wave 0 of the split-form 3x3 kernels (tools/conv_trace.py; PIDM_STREAM_TRACE=1)
__device__ unsigned long long g_stream_trace[4 * 64];

#ifndef PIDM_SPLIT_CHAINS
#define PIDM_SPLIT_CHAINS 0   // measured (round 3): alternating the six terms of a tap between the two accumulators changes nothing (10.823 vs 10.821 ms per step)
#endif
#ifndef PIDM_SPLIT_ABLATE
#define PIDM_SPLIT_ABLATE 0   // measurement builds only (tools/split_ablate.py): 1 / 2 = B / A fragments read for tap 0 only, 4 / 8 = no A / B staging, 32 = no split arithmetic
#endif
static constexpr int kSplitRow = 112;                 // bytes per LDS row
// extra bytes per halo-tile row (ConvGeom::rpad): see conv3x3_split_kernel; PIDM_SPLIT_ROWPAD=0: off (A/B measurements)
static int split_row_pad(int Wv) {
  const int on = [] { const char* e = knob("PIDM_SPLIT_ROWPAD"); return e ? atoi(e) : 1; }();
  return (on && Wv <= 16) ? 32 : 0;
}
#ifndef PIDM_WS_PRODUCER_PRIO
#define PIDM_WS_PRODUCER_PRIO 0
#endif
#ifndef PIDM_WS_INTERLEAVE
#define PIDM_WS_INTERLEAVE 1    // fragment reads of the warp-specialised consumers pinned one per MFMA (0: the compiler's placement)
#endif
#ifndef PIDM_WS_PF
#define PIDM_WS_PF 2            // fragment prefetch distance of the warp-specialised consumers (taps; 1: rounds 3-5)
#endif
#ifndef PIDM_WS_LEAVE_FETCH
#define PIDM_WS_LEAVE_FETCH 1   // 0: the producer waves drain every load before each stage barrier (rounds 3-5; A/B builds)
#endif
static constexpr bool kWsLeaveFetch = PIDM_WS_LEAVE_FETCH != 0;
static constexpr int kSplitSlab = 9 * 32 * kSplitRow; // bytes of pre-split weights per stage (rows padded like the LDS rows)

// Workgroup -> position in the launch's item order.  Hardware places workgroup b on XCD b % 8 (observed, relied on for speed only),
// and every XCD has its own L2: with the plain order the 32 resident workgroups of an XCD are spread over ALL n-tiles of a layer
// (every XCD streams every weight slab through its L2), with xcd = a > 0 the workgroups of one XCD take CONSECUTIVE items - one or a
// few n-tiles' slabs shared by all of them.  Needs a grid that is a multiple of 8 (else: plain order).
__device__ __forceinline__ int split_vblock(int xcd) {
  const int b = (int)blockIdx.x, n = (int)gridDim.x;
  if (xcd <= 0 || (n & 7)) return b;
  return (b & 7) * (n >> 3) + (b >> 3);
}

// Position of a stage inside a workgroup's range of work items, kept INCREMENTALLY: stage -> (item, chunk) -> (n-tile, m-tile) ->
// (image group, row tile) used to be five integer divisions by launch-time values at the top of every stage; the compiler expands
// each into ~25 vector instructions around v_rcp_iflag_f32 (there is no scalar divide), on the critical path between the stage's
// barrier and its first loads, in both waves of a SIMD at once.  All fields are wave-uniform (SGPRs).
struct SplitCursor {
  int ch, cc, ph;   // 16-channel chunk of the item; MODE 1: chunk inside the K-phase, K-phase
  int tq, tn, zz;   // n-tile index of the item = zz * ntn + tn (zz = output parity, MODE 2 only)
  int tm, bi, rt;   // m-tile = bi * tpi + rt: image group, row tile inside the image
  int left;         // stages after this one
};
template <int MODE>
__device__ __forceinline__ void split_cursor_init(SplitCursor& c, int item0, int nst, int tiles_m, int tpi, int ntn) {
  c.ch = c.cc = c.ph = 0;
  c.tq = item0 / tiles_m;
  c.tm = item0 - c.tq * tiles_m;
  c.zz = c.tq / ntn;
  c.tn = c.tq - c.zz * ntn;
  c.bi = c.tm / tpi;
  c.rt = c.tm - c.bi * tpi;
  c.left = nst - 1;
}
// next stage; stays on the last one (the pipelines fetch "stage nst" and "nst + 1" as copies of the last)
template <int MODE>
__device__ __forceinline__ void split_cursor_next(SplitCursor& c, int NCH, int CCH, int tiles_m, int tpi, int ntn) {
  if (c.left <= 0) return;
  --c.left;
  ++c.ch;
  if (MODE == 1 && ++c.cc == CCH) { c.cc = 0; ++c.ph; }
  if (c.ch == NCH) {
    c.ch = c.cc = c.ph = 0;
    ++c.tm;
    if (++c.rt == tpi) { c.rt = 0; ++c.bi; }
    if (c.tm == tiles_m) {
      c.tm = c.bi = c.rt = 0;
      ++c.tq;
      if (++c.tn == ntn) { c.tn = 0; ++c.zz; }
    }
  }
}

// MODE 0: 3x3 / stride 1 / pad 1.  MODE 1: the 4x4 / stride-2 convolution (and the input gradient of the transposed one) as 4
// K-phases of 2x2 taps over the parity sub-images of the input (ConvGeom::nph = 4: stage = (tile, phase, 16-channel chunk), the
// halo tile is gathered with a stride of 2 pixels).  MODE 2: the transposed 4x4 / stride-2 convolution (and the input gradient of
// the strided one) as 4 output-parity problems of 2x2 taps (ConvGeom::nz = 4: work item = (parity, n-tile, m-tile), the epilogue
// scatters with an output stride of 2).  Both keep the 3x3 halo-tile layout (one halo row / column on either side; the tap
// offsets depend on the phase / parity) and run 4 taps = 24 MFMAs per wave and stage.
template <int NW, int MODE>
__global__ void __launch_bounds__(64 * NW) conv3x3_split_kernel(ConvGeom g, const float* __restrict__ src0, const float* __restrict__ src1,
                                                            const unsigned short* __restrict__ ws, const float* __restrict__ bias,
                                                            const float* __restrict__ residual, float* __restrict__ out, int n_items,
                                                            int items_per_wg, int trace) {
  constexpr int RB = kSplitRow, T = (MODE == 0) ? 9 : 4, NT = 64 * NW;   // NW waves: 8 (256-pixel tile, two waves per SIMD) or 4 (128 pixels)
  constexpr int NB = ((MODE == 0) ? 32 : 16) / NW;          // 1 KB pieces of the weight slab per wave (31.5 KB / 14 KB: the tail spills)
  constexpr int NBT = (MODE == 0) ? 4 : 2;                  // taps the slab copies are spread over
  constexpr int SLAB = T * 32 * kSplitRow, BPAD = (MODE == 0) ? 512 : 2048;
  HIP_DYNAMIC_SHARED(float, smemf)
  char* smem = reinterpret_cast<char*>(smemf);
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int half = lane >> 5, l31 = lane & 31;
  // bytes per halo-tile row: IWt pixels + ConvGeom::rpad (32 bytes at the 8- and 16-wide levels, where the 32 pixels of a wave span
  // 4 / 2 tile rows: with it consecutive rows start 8 / 0 bank groups apart (mod 16) and the 16 lanes one ds_read_b128 cycle serves hit
  // 16 different groups; without it 33-48 % of the LDS cycles of those levels were bank conflicts)
  const int rowB = g.IWt * RB + g.rpad, nrowsA = g.NI * g.IHt;
  const int b_reg = nrowsA * rowB;                       // byte offset of the weight rows inside a buffer
  const int bufsz = b_reg + T * 32 * RB + BPAD;        // bytes per buffer: halo tile | T x 32 weight rows | spill of the last 1 KB pieces
  const int tpi = g.Hv / g.TH;
  const int NCH = (MODE == 1 ? g.Kw : g.Cin) >> 4;     // 16-channel chunks per tile (MODE 1: 4 phases x Cin)
  const int CCH = g.Cin >> 4;
  const int ntn = g.Cout >> 5;
  const int item0 = split_vblock(g.xcd) * items_per_wg;
  const int my_items = (item0 + items_per_wg <= n_items) ? items_per_wg : n_items - item0;
  const int nst = my_items * NCH;

  // operand fragments of this lane: A row = pixel wave*32 + l31 of the tile, B row = output channel l31 of the tap
  const int pm = wave * 32 + l31;
  const int a_tx = pm & (g.Wv - 1), a_ty = (pm >> g.wsh) & (g.TH - 1), a_img = pm >> (g.wsh + g.tsh);
  const int a_frag = ((a_img < g.NI) ? (a_img * g.IHt + a_ty) * rowB + a_tx * RB : 0) + 48 * half;
  const int b_frag = b_reg + l31 * RB + 48 * half;

  // Activation staging: a unit = 8 channels of one pixel (32 bytes in, 3 x 16 bytes out); whole image rows, the halo columns
  // are zero for every tile and written once.  Units tid and tid + NT; the second exists for the first nA1 waves.
  const int SEG = g.NI * g.IHt * g.Wv;                 // staged pixels per tile
  const int nA1 = (2 * SEG - NT) >> 6;
  const int hh = tid & 1;
  int a_lds[2], a_im[2], a_hy[2];
  unsigned a_vo[2], a_ro[2];      // byte offsets inside the source: of the unit inside its row / plus its row's offset from the tile's first
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    int sp = (tid + NT * k) >> 1;
    if (sp >= SEG) sp = SEG - 1;
    const int sr = sp >> g.wsh, x = sp & (g.Wv - 1);
    const int img = fast_div(sr, g.IHt, g.mIHt), hy = sr - img * g.IHt;
    a_lds[k] = (img * g.IHt + hy) * rowB + (x + 1) * RB + 48 * hh;
    a_vo[k] = (unsigned)((MODE == 1 ? 2 * x : x) * g.ld0 + 8 * hh) * 4u;
    a_im[k] = img;
    a_hy[k] = hy;
    // 32-bit: the launcher takes this kernel only for sources below 4 GB
    a_ro[k] = a_vo[k] + (unsigned)(img * g.Hi + (MODE == 1 ? 2 * hy : hy)) * (unsigned)(g.Wi * g.ld0 * 4);
  }
  // Weight staging: the stage's slab (pre-split, rows already padded to 112 bytes: an image of the LDS rows) is copied by
  // global_load_lds_dwordx4 - 1 KB per wave and instruction straight into the buffer being filled, no registers, no ds_write;
  // the __syncthreads() at the end of the stage drains it (vmcnt) together with everything else
  const unsigned b_lane = 16u * lane;

  // zero halo columns of both buffers
  for (int e = tid; e < 2 * g.NI * g.IHt * 2 * 6; e += NT) {
    const int q = e % 6, side = (e / 6) & 1, row = (e / 12) % (g.NI * g.IHt), bufi = (e / 12) / (g.NI * g.IHt);
    const u32x4 zero4 = {0u, 0u, 0u, 0u};
    *reinterpret_cast<u32x4*>(smem + (size_t)bufi * bufsz + (size_t)row * rowB + (size_t)(side ? g.IWt - 1 : 0) * RB + 16 * q) = zero4;
  }

  f32x4 ra[2][2];
  float akeep[2] = {0.f, 0.f};
  const char* l_sp = reinterpret_cast<const char*>(src0);
  const char* l_wn = reinterpret_cast<const char*>(ws);
  int l_b0 = 0, l_iy0 = 0, l_py = 0;
  long long l_rb = 0;             // byte offset of the tile's first staged row (row -1 of the first image: may be negative)
  SplitCursor lc, cs;             // stage whose activations are being fetched (two ahead) / stage being computed
  split_cursor_init<MODE>(lc, item0, nst, g.tiles_m, tpi, ntn);
  cs = lc;
  // geometry of the fetch cursor's stage
#define PIDM_SP_STAGE()                                                                                            \
  {                                                                                                                \
    const int ch__ = lc.ch, tq__ = lc.tq;                                      /* tq = n-tile (MODE 2: parity * ntn + n-tile) */ \
    const int ph__ = (MODE == 1) ? lc.ph : 0;                                                                      \
    const int c0__ = ((MODE == 1) ? lc.cc : lc.ch) * 16;                                                           \
    l_b0 = lc.bi * g.NI;                                                                                           \
    l_iy0 = lc.rt * g.TH - 1;                                                 /* input (sub-image) row of LDS row 0 */ \
    l_py = (MODE == 1) ? g.ph_oy[ph__] : 0;                                                                        \
    l_rb = (long long)(l_b0 * g.Hi + (MODE == 1 ? 2 * l_iy0 + l_py : l_iy0)) * (long long)(g.Wi * g.ld0 * 4);      \
    l_sp = reinterpret_cast<const char*>(((c0__ < g.C0) ? src0 + c0__ : src1 + (c0__ - g.C0)) +                    \
                                         ((MODE == 1) ? g.ph_ox[ph__] * g.ld0 : 0));                               \
    l_wn = reinterpret_cast<const char*>(ws) + ((size_t)tq__ * NCH + ch__) * SLAB;                                 \
  }
  // unconditional loads (rows outside the image read row 0 and are zeroed on their way to LDS)
#define PIDM_SP_LOAD_A(k_)                                                                                         \
  {                                                                                                                \
    /* address = scalar (source + tile's first row) + a per-thread constant; rows outside the image read the unit of row 0 */ \
    const int b__ = l_b0 + a_im[k_], iy__ = l_iy0 + a_hy[k_];                                                      \
    const bool ok__ = (b__ < g.B) & (iy__ >= 0) & (iy__ < (MODE == 1 ? g.Hv : g.Hi));                              \
    const f32x4* p__ = reinterpret_cast<const f32x4*>(ok__ ? l_sp + l_rb + a_ro[k_] : l_sp + a_vo[k_]);            \
    ra[k_][0] = p__[0];                                                                                            \
    ra[k_][1] = p__[1];                                                                                            \
    akeep[k_] = ok__ ? 1.f : 0.f;                                                                                  \
  }
#define PIDM_SP_COPY_B(k_, wn_, buf_) pidm_glds_b128((wn_) + 1024 * (wave + NW * (k_)) + b_lane, (buf_) + b_reg + 1024 * (wave + NW * (k_)));
#define PIDM_SP_WRITE_A(k_, buf_)                                                                                  \
  {                                                                                                                \
    const f32x4 v0__ = ra[k_][0] * akeep[k_], v1__ = ra[k_][1] * akeep[k_];                                        \
    unsigned q0__[4], q1__[4], q2__[4];                                                                            \
    if (PIDM_SPLIT_ABLATE & 32) {                                                                                  \
      _Pragma("unroll") for (int i__ = 0; i__ < 4; ++i__) {                                                        \
        q0__[i__] = __float_as_uint(v0__[i__]); q1__[i__] = __float_as_uint(v1__[i__]); q2__[i__] = q0__[i__];     \
      }                                                                                                            \
    } else {                                                                                                       \
    pidm_split3_pk(v0__[0], v0__[1], q0__[0], q1__[0], q2__[0]);                                                   \
    pidm_split3_pk(v0__[2], v0__[3], q0__[1], q1__[1], q2__[1]);                                                   \
    pidm_split3_pk(v1__[0], v1__[1], q0__[2], q1__[2], q2__[2]);                                                   \
    pidm_split3_pk(v1__[2], v1__[3], q0__[3], q1__[3], q2__[3]);                                                   \
    }                                                                                                              \
    u32x4* d__ = reinterpret_cast<u32x4*>((buf_) + a_lds[k_]);                                                     \
    d__[0] = u32x4{q0__[0], q0__[1], q0__[2], q0__[3]};                                                            \
    d__[1] = u32x4{q1__[0], q1__[1], q1__[2], q1__[3]};                                                            \
    d__[2] = u32x4{q2__[0], q2__[1], q2__[2], q2__[3]};                                                            \
  }

  char* bufc = smem;               // buffer the MFMAs read
  char* bufn = smem + bufsz;       // buffer being filled
  PIDM_SP_STAGE()
  PIDM_SP_LOAD_A(0) PIDM_SP_LOAD_A(1)
#pragma unroll
  for (int k = 0; k < NB; ++k) PIDM_SP_COPY_B(k, l_wn, bufc)
  PIDM_SP_WRITE_A(0, bufc)
  if (wave < nA1) PIDM_SP_WRITE_A(1, bufc)
  split_cursor_next<MODE>(lc, NCH, CCH, g.tiles_m, tpi, ntn);
  PIDM_SP_STAGE()
  PIDM_SP_LOAD_A(0) PIDM_SP_LOAD_A(1)
  __syncthreads();

  // two accumulators (even / odd taps), added once per tile: every MFMA rounds its sum into the accumulator, so the error grows
  // with the number of additions into ONE register - two chains keep it at the level of the fp32-MFMA kernels
  // (tests/test_kernels_conv.py: test_split_form_is_as_accurate_as_the_fp32_mfma)
  f32x16 acc, accb;
  for (int r = 0; r < 16; ++r) { acc[r] = 0.f; accb[r] = 0.f; }
  // PIDM_STREAM_TRACE=1: cycle stamps of workgroup 0, waves 0 and 4 (one SIMD): [wave][stage < 32][top, taps done, epilogue done, past barrier]
  const int tr_base = (trace && blockIdx.x == 0 && lane == 0 && (wave & 3) == 0) ? (wave >> 2) * 128 : -1;
  for (int s = 0; s < nst; ++s) {
    if (tr_base >= 0 && s < 32) g_stream_trace[tr_base + 4 * s + 0] = clock64();
    const char* wn1 = l_wn;        // weight slab of stage s+1 (copied during this stage)
    split_cursor_next<MODE>(lc, NCH, CCH, g.tiles_m, tpi, ntn);
    PIDM_SP_STAGE()                // geometry of the activation loads issued during this stage (stage s+2)
    // the epilogue's bias, fetched a stage ahead of its use (unconditional load, any valid address when there is no bias)
    const float bv_pre = (bias ? bias : reinterpret_cast<const float*>(ws))[cs.tn * 32 + l31];
    // tap offsets inside the halo tile: 3x3: (ky, kx); 2x2 taps: (jy - pad_y + 1, jx - pad_x + 1) with the pads of this stage's
    // phase (MODE 1) or of this item's output parity (MODE 2)
    int oy0 = 0, ox0 = 0;
    if (MODE == 1) {
      oy0 = 1 - g.ph_pad_y[cs.ph]; ox0 = 1 - g.ph_pad_x[cs.ph];
    } else if (MODE == 2) {
      oy0 = 1 - g.pad_y[cs.zz]; ox0 = 1 - g.pad_x[cs.zz];
    }
    const char* afp = bufc + a_frag + ((MODE == 0) ? 0 : oy0 * rowB + ox0 * RB);
    const char* bfp = bufc + b_frag;
    u32x4 fa[2][3], fb[2][3];
#define PIDM_SP_FRAGS(set_, t_)                                                                                    \
  {                                                                                                                \
    const u32x4* ar__ = reinterpret_cast<const u32x4*>(afp + (size_t)((MODE == 0) ? ((t_) / 3) * rowB + ((t_) % 3) * RB : ((t_) >> 1) * rowB + ((t_) & 1) * RB)); \
    const u32x4* br__ = reinterpret_cast<const u32x4*>(bfp + (size_t)((t_)*32) * RB);                              \
    _Pragma("unroll") for (int p = 0; p < 3; ++p) {                                                                \
      if (!(PIDM_SPLIT_ABLATE & 2) || (t_) == 0) fa[set_][p] = ar__[p]; else fa[set_][p] = fa[(set_) ^ 1][p];      \
      if (!(PIDM_SPLIT_ABLATE & 1) || (t_) == 0) fb[set_][p] = br__[p]; else fb[set_][p] = fb[(set_) ^ 1][p];      \
    }                                                                                                              \
  }
    PIDM_SP_FRAGS(0, 0)
    // Measured and left out (tools/split_variants.py, profiles/r02_split_conv_notes.txt): staggering the staging pieces between
    // the two waves of a SIMD (either as two copies of the tap loop or only the two VALU-heavy pieces) and alternating
    // s_setprio per tap or around the MFMAs - all within +-3 % or slower.  Ablation (tools/split_ablate.py): without the staging a
    // stage runs at the matrix pipe's 3780 cycles, with it ~5800 - fragment reads, barrier and arbitration are free, the staging
    // (loads, split arithmetic, LDS writes) is what the hardware does not hide under the other wave's MFMAs.
#pragma unroll
    for (int t = 0; t < T; ++t) {
      const int cur = t & 1;
      if (t + 1 < T) PIDM_SP_FRAGS(cur ^ 1, t + 1)
      // small terms first
      // PIDM_SPLIT_CHAINS = 1: the six terms of a tap alternate between the two accumulators (consecutive MFMAs never depend on
      // each other), 0: even taps -> acc, odd taps -> accb
#if PIDM_SPLIT_CHAINS
      acc = pidm_mfma_bf16_32x32x16(fa[cur][2], fb[cur][0], acc);
      accb = pidm_mfma_bf16_32x32x16(fa[cur][0], fb[cur][2], accb);
      acc = pidm_mfma_bf16_32x32x16(fa[cur][1], fb[cur][1], acc);
      accb = pidm_mfma_bf16_32x32x16(fa[cur][1], fb[cur][0], accb);
      acc = pidm_mfma_bf16_32x32x16(fa[cur][0], fb[cur][1], acc);
      accb = pidm_mfma_bf16_32x32x16(fa[cur][0], fb[cur][0], accb);
#else
      if (t & 1) {
        accb = pidm_mfma_bf16_32x32x16(fa[cur][2], fb[cur][0], accb);
        accb = pidm_mfma_bf16_32x32x16(fa[cur][0], fb[cur][2], accb);
        accb = pidm_mfma_bf16_32x32x16(fa[cur][1], fb[cur][1], accb);
        accb = pidm_mfma_bf16_32x32x16(fa[cur][1], fb[cur][0], accb);
        accb = pidm_mfma_bf16_32x32x16(fa[cur][0], fb[cur][1], accb);
        accb = pidm_mfma_bf16_32x32x16(fa[cur][0], fb[cur][0], accb);
      } else {
        acc = pidm_mfma_bf16_32x32x16(fa[cur][2], fb[cur][0], acc);
        acc = pidm_mfma_bf16_32x32x16(fa[cur][0], fb[cur][2], acc);
        acc = pidm_mfma_bf16_32x32x16(fa[cur][1], fb[cur][1], acc);
        acc = pidm_mfma_bf16_32x32x16(fa[cur][1], fb[cur][0], acc);
        acc = pidm_mfma_bf16_32x32x16(fa[cur][0], fb[cur][1], acc);
        acc = pidm_mfma_bf16_32x32x16(fa[cur][0], fb[cur][0], acc);
      }
#endif
      // staging pieces: a slot's registers go to LDS (the data of stage s+1) and are re-loaded at once with stage s+2's
      if (t == 0 && !(PIDM_SPLIT_ABLATE & 4)) { PIDM_SP_WRITE_A(0, bufn) PIDM_SP_LOAD_A(0) }
      if (t == 1 && !(PIDM_SPLIT_ABLATE & 4)) { if (wave < nA1) PIDM_SP_WRITE_A(1, bufn) PIDM_SP_LOAD_A(1) }
      if (t >= 2 && t < 2 + NBT && !(PIDM_SPLIT_ABLATE & 8)) {    // the weight copies of stage s+1, spread over taps 2..
#pragma unroll
        for (int k = NB * (t - 2) / NBT; k < NB * (t - 1) / NBT; ++k) PIDM_SP_COPY_B(k, wn1, bufn)
      }
      __builtin_amdgcn_sched_barrier(0);
    }
#undef PIDM_SP_FRAGS
    if (tr_base >= 0 && s < 32) g_stream_trace[tr_base + 4 * s + 1] = clock64();
    // ---- last chunk of a tile: epilogue as in conv3x3_stream_kernel (bias, GroupNorm partial sums, 4x4 register transposes,
    //      residual, 16-byte stores), accumulator restarts ----
    if (cs.ch == NCH - 1) {
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[r] += accb[r];
      const int zz = (MODE == 2) ? cs.zz : 0;
      const int b0 = cs.bi * g.NI, vy0 = cs.rt * g.TH, n0 = cs.tn * 32;
      const int c = n0 + l31;
      const float bv = bias ? bv_pre : 0.f;
      const int p0 = wave * 32;
      const int tx0 = p0 & (g.Wv - 1), ty0 = (p0 >> g.wsh) & (g.TH - 1), img0 = p0 >> (g.wsh + g.tsh);
      const int b = b0 + img0;
      if (b < g.B && img0 < g.NI) {        // wave-uniform
        const int pin = (vy0 + ty0) * g.Wv + tx0;
        float v[16];
        float gs1 = 0.f, gs2 = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          v[r] = acc[r] + bv;
          gs1 += v[r];
          gs2 += v[r] * v[r];
        }
        if (g.gn_part) PIDM_GN_PARTIAL(gs1, gs2, b, pin, c)
        if (g.bn_part) {
          const float* xrow = g.bn_x + ((size_t)b * g.Ho * g.Wo + pin) * g.Cout + c;
          PIDM_BN_PARTIAL(acc, bv, b, pin, c, xrow, g.Cout, (g.bn_res && residual != nullptr), residual + ((size_t)b * g.Ho * g.Wo + pin) * g.ldr + c, g.ldr)
        }
        const bool odd1 = (l31 & 1) != 0, odd2 = (l31 & 2) != 0;
        const size_t opix = (size_t)b * g.sob + (size_t)pin * g.sox + n0 + 4 * (l31 >> 2);
        const size_t rpix = ((size_t)b * g.Ho * g.Wo + pin) * g.ldr + n0 + 4 * (l31 >> 2);
        // (stride-1 forms: the four residual quads of the tile as one batch of loads - inside the q4 loop each was a branch, a load
        //  and a wait behind the previous quad's store)
        f32x4 rres[4];
        if (MODE != 2 && residual) {
#pragma unroll
          for (int q4 = 0; q4 < 4; ++q4) rres[q4] = *reinterpret_cast<const f32x4*>(residual + rpix + (size_t)(8 * q4 + 4 * half + (l31 & 3)) * g.ldr);
        }
#pragma unroll
        for (int q4 = 0; q4 < 4; ++q4) {
          float x0 = v[4 * q4], x1 = v[4 * q4 + 1], x2 = v[4 * q4 + 2], x3 = v[4 * q4 + 3];
          const float r01 = pidm_quad_xor1(odd1 ? x0 : x1), r23 = pidm_quad_xor1(odd1 ? x2 : x3);
          x0 = odd1 ? r01 : x0; x1 = odd1 ? x1 : r01;
          x2 = odd1 ? r23 : x2; x3 = odd1 ? x3 : r23;
          const float r02 = pidm_quad_xor2(odd2 ? x0 : x2), r13 = pidm_quad_xor2(odd2 ? x1 : x3);
          x0 = odd2 ? r02 : x0; x2 = odd2 ? x2 : r02;
          x1 = odd2 ? r13 : x1; x3 = odd2 ? x3 : r13;
          const int prow = 8 * q4 + 4 * half + (l31 & 3);
          f32x4 o = {x0, x1, x2, x3};
          if (MODE == 2) {
            // output parity (ooy, oox): pixel (vy, vx) of the tile goes to (2 vy + ooy, 2 vx + oox)
            const int pp = p0 + prow, vx = pp & (g.Wv - 1), vy = vy0 + ((pp >> g.wsh) & (g.TH - 1));
            const size_t opx = (size_t)(2 * vy + g.ooy[zz]) * g.Wo + 2 * vx + g.oox[zz];
            if (residual) o += *reinterpret_cast<const f32x4*>(residual + ((size_t)b * g.Ho * g.Wo + opx) * g.ldr + n0 + 4 * (l31 >> 2));
            *reinterpret_cast<f32x4*>(out + (size_t)b * g.sob + opx * g.sox + n0 + 4 * (l31 >> 2)) = o;
          } else {
          if (residual) o += rres[q4];
          *reinterpret_cast<f32x4*>(out + opix + (size_t)prow * g.sox) = o;
          }
        }
      }
      for (int r = 0; r < 16; ++r) { acc[r] = 0.f; accb[r] = 0.f; }
    }
    if (tr_base >= 0 && s < 32) g_stream_trace[tr_base + 4 * s + 2] = clock64();
    split_cursor_next<MODE>(cs, NCH, CCH, g.tiles_m, tpi, ntn);
    __syncthreads();               // buffer (s+1)&1 complete, buffer s&1 free
    if (tr_base >= 0 && s < 32) g_stream_trace[tr_base + 4 * s + 3] = clock64();
    char* tswap = bufc; bufc = bufn; bufn = tswap;
  }
#undef PIDM_SP_STAGE
#undef PIDM_SP_LOAD_A
#undef PIDM_SP_COPY_B
#undef PIDM_SP_WRITE_A
}

// ---------------------------------------------------------------------------------------------------
// The same kernel with the two jobs of a stage given to DIFFERENT waves ("warp-specialised", round 3).  The ablation of
// conv3x3_split_kernel (profiles/r02_split_conv_notes.txt) shows that a stage without its staging runs at the matrix pipe's pace
// (3850 of 3780 cycles) and that the staging work - global loads, 3-piece split, LDS writes, weight-slab copies: ~900 VALU cycles
// per SIMD - is what the two waves of a SIMD fail to hide under each other's MFMAs: both reach it at the same point of the stage
// (the barrier lines them up) and the matrix pipe idles through it.  Here NW consumer waves (two per SIMD, as before) ONLY read
// operand fragments and issue MFMAs (+ the tile epilogue), and 4 producer waves (one per SIMD) ONLY stage: during stage s they
// write the activations of stage s+1 (fetched during stage s-1) into the other LDS buffer, start that stage's weight-slab copies
// and fetch the activations of stage s+2.  The producers' VALU and LDS-write work runs beside the consumers' MFMAs on every SIMD
// for the whole stage (~900 of ~3850 cycles).  Same buffers, same barrier per stage, same arithmetic in the same order: results
// are bit-identical to conv3x3_split_kernel.  768 (NW = 8) or 512 (NW = 4) threads.
// ---------------------------------------------------------------------------------------------------
template <int NW, int MODE, int NPW, int MS>
__global__ void __launch_bounds__(64 * (NW / MS) + 64 * NPW) conv3x3_split_ws_kernel(ConvGeom g, const float* __restrict__ src0, const float* __restrict__ src1,
                                                                        const unsigned short* __restrict__ ws, const float* __restrict__ bias,
                                                                        const float* __restrict__ residual, float* __restrict__ out,
                                                                        int n_items, int items_per_wg, int trace) {
  constexpr int RB = kSplitRow, T = (MODE == 0) ? 9 : 4;
  // NPW producer waves (4, or 8 = two per SIMD, round 6): what a producer wave spends per stage is ISSUE time - ~150-185 cycles per
  // LDS-direct copy and ~7 per vector instruction beside the consumers' MFMAs (stamps: 8 copies + the fetch = 2400-2600 cycles, the
  // staging arithmetic 600, against 2200-2400 cycles of the consumers' taps): with 4 producers the CONSUMERS waited 700-1100 cycles
  // per stage at the barrier; 8 halve every producer's share
  constexpr int NPT = 64 * NPW;                             // producer threads
  constexpr int NB = ((MODE == 0) ? 32 : 16) / NPW;         // 1 KB pieces of the weight slab per producer wave
  constexpr int KA = 2 * NW / NPW;                          // staging units (8 channels of one pixel) per producer thread
  constexpr int NWC = NW / MS;                              // consumer waves (NW = 32-pixel sub-tiles of the workgroup's tile, MS per wave)
  static_assert(MS == 1 || MS == 2, "sub-tiles per consumer wave");
  static_assert(NPW == 4 || NPW == 8, "producer waves: one or two per SIMD");
  static_assert(KA >= 1 && NB >= 1, "every producer wave stages something");
  constexpr int SLAB = T * 32 * kSplitRow, BPAD = (MODE == 0) ? 512 : 2048;
  HIP_DYNAMIC_SHARED(float, smemf)
  char* smem = reinterpret_cast<char*>(smemf);
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const bool producer = wave >= NWC;
  const int half = lane >> 5, l31 = lane & 31;
  const int rowB = g.IWt * RB + g.rpad, nrowsA = g.NI * g.IHt;    // (row pitch: see conv3x3_split_kernel)
  const int b_reg = nrowsA * rowB;
  const int bufsz = b_reg + T * 32 * RB + BPAD;
  const int tpi = g.Hv / g.TH;
  const int NCH = (MODE == 1 ? g.Kw : g.Cin) >> 4;
  const int CCH = g.Cin >> 4;
  const int ntn = g.Cout >> 5;
  const int item0 = split_vblock(g.xcd) * items_per_wg;
  const int my_items = (item0 + items_per_wg <= n_items) ? items_per_wg : n_items - item0;
  const int nst = my_items * NCH;

  // zero halo columns of both buffers (all waves)
  for (int e = tid; e < 2 * g.NI * g.IHt * 2 * 6; e += 64 * NWC + NPT) {
    const int q = e % 6, side = (e / 6) & 1, row = (e / 12) % (g.NI * g.IHt), bufi = (e / 12) / (g.NI * g.IHt);
    const u32x4 zero4 = {0u, 0u, 0u, 0u};
    *reinterpret_cast<u32x4*>(smem + (size_t)bufi * bufsz + (size_t)row * rowB + (size_t)(side ? g.IWt - 1 : 0) * RB + 16 * q) = zero4;
  }
  char* bufc = smem;               // buffer the MFMAs read
  char* bufn = smem + bufsz;       // buffer being filled

  if (producer) {
    // =============================== producer waves: staging only ===============================
    const int pt = tid - 64 * NWC, pw = wave - NWC;         // producer thread / wave index
#if PIDM_WS_PRODUCER_PRIO
    __builtin_amdgcn_s_setprio(PIDM_WS_PRODUCER_PRIO);      // (experiment: the producers' vector instructions win the issue arbitration)
#endif
    const int SEG = g.NI * g.IHt * g.Wv;                    // staged pixels per tile
    const int hh = pt & 1;
    int a_lds[KA], a_im[KA], a_hy[KA];
    unsigned a_vo[KA], a_ro[KA];
#pragma unroll
    for (int k = 0; k < KA; ++k) {
      int sp = (pt + NPT * k) >> 1;
      if (sp >= SEG) sp = SEG - 1;                          // surplus slots repeat the last unit (same data to the same place)
      const int sr = sp >> g.wsh, x = sp & (g.Wv - 1);
      const int img = fast_div(sr, g.IHt, g.mIHt), hy = sr - img * g.IHt;
      a_lds[k] = (img * g.IHt + hy) * rowB + (x + 1) * RB + 48 * hh;
      a_vo[k] = (unsigned)((MODE == 1 ? 2 * x : x) * g.ld0 + 8 * hh) * 4u;
      a_im[k] = img;
      a_hy[k] = hy;
      a_ro[k] = a_vo[k] + (unsigned)(img * g.Hi + (MODE == 1 ? 2 * hy : hy)) * (unsigned)(g.Wi * g.ld0 * 4);
    }
    const unsigned b_lane = 16u * lane;
    // two register sets: the activations of stage k travel in set k & 1 - fetched during stage k-2 (a whole stage before they are
    // needed: with a single set the fetch latency sat between every barrier and the first LDS write of the next stage, and the
    // producers, not the matrix pipe, set the pace), split and written to LDS during stage k-1
    f32x4 ra[2][KA][2];
    float akeep[2][KA];
    const char* l_sp = reinterpret_cast<const char*>(src0);
    const char* l_wn = reinterpret_cast<const char*>(ws);
    int l_b0 = 0, l_iy0 = 0, l_py = 0;
    long long l_rb = 0;
    SplitCursor lc;               // the stage whose activations are being fetched
    split_cursor_init<MODE>(lc, item0, nst, g.tiles_m, tpi, ntn);
    // PIDM_STREAM_TRACE=1: cycle stamps of workgroup 0 - slot 0: consumer wave 0 [stage top, taps done, epilogue done, past barrier],
    // slot 1: producer wave 0 [iteration top, set W landed, staging written, past barrier]
    const int tr_p = (trace && blockIdx.x == 0 && lane == 0 && pw == 0) ? 128 : -1;
#define PIDM_WS_STAGE()                                                                                            \
  {                                                                                                                \
    const int ch__ = lc.ch, tq__ = lc.tq;                                                                          \
    const int ph__ = (MODE == 1) ? lc.ph : 0;                                                                      \
    const int c0__ = ((MODE == 1) ? lc.cc : lc.ch) * 16;                                                           \
    l_b0 = lc.bi * g.NI;                                                                                           \
    l_iy0 = lc.rt * g.TH - 1;                                                                                      \
    l_py = (MODE == 1) ? g.ph_oy[ph__] : 0;                                                                        \
    l_rb = (long long)(l_b0 * g.Hi + (MODE == 1 ? 2 * l_iy0 + l_py : l_iy0)) * (long long)(g.Wi * g.ld0 * 4);      \
    l_sp = reinterpret_cast<const char*>(((c0__ < g.C0) ? src0 + c0__ : src1 + (c0__ - g.C0)) +                    \
                                         ((MODE == 1) ? g.ph_ox[ph__] * g.ld0 : 0));                               \
    l_wn = reinterpret_cast<const char*>(ws) + ((size_t)tq__ * NCH + ch__) * SLAB;                                 \
  }
#define PIDM_WS_LOAD_A(set_, k_)                                                                                   \
  {                                                                                                                \
    const int b__ = l_b0 + a_im[k_], iy__ = l_iy0 + a_hy[k_];                                                      \
    const bool ok__ = (b__ < g.B) & (iy__ >= 0) & (iy__ < (MODE == 1 ? g.Hv : g.Hi));                              \
    const char* p__ = ok__ ? l_sp + l_rb + a_ro[k_] : l_sp + a_vo[k_];                                             \
    if (kWsLeaveFetch) {                                                                                           \
      PIDM_UNTRACKED_LOAD_F32X4(ra[set_][k_][0], p__, 0);                                                          \
      PIDM_UNTRACKED_LOAD_F32X4(ra[set_][k_][1], p__, 16);                                                         \
    } else {                                                                                                       \
      ra[set_][k_][0] = reinterpret_cast<const f32x4*>(p__)[0];                                                    \
      ra[set_][k_][1] = reinterpret_cast<const f32x4*>(p__)[1];                                                    \
    }                                                                                                              \
    akeep[set_][k_] = ok__ ? 1.f : 0.f;                                                                            \
  }
#define PIDM_WS_COPY_B(k_, wn_, buf_)                                                                              \
  {                                                                                                                \
    if (kWsLeaveFetch) pidm_glds_b128_untracked((wn_) + 1024 * (pw + NPW * (k_)) + b_lane, (buf_) + b_reg + 1024 * (pw + NPW * (k_))); \
    else pidm_glds_b128((wn_) + 1024 * (pw + NPW * (k_)) + b_lane, (buf_) + b_reg + 1024 * (pw + NPW * (k_)));      \
  }
#define PIDM_WS_WRITE_A(set_, k_, buf_)                                                                            \
  {                                                                                                                \
    const f32x4 v0__ = ra[set_][k_][0] * akeep[set_][k_], v1__ = ra[set_][k_][1] * akeep[set_][k_];                \
    unsigned q0__[4], q1__[4], q2__[4];                                                                            \
    pidm_split3_pk(v0__[0], v0__[1], q0__[0], q1__[0], q2__[0]);                                                   \
    pidm_split3_pk(v0__[2], v0__[3], q0__[1], q1__[1], q2__[1]);                                                   \
    pidm_split3_pk(v1__[0], v1__[1], q0__[2], q1__[2], q2__[2]);                                                   \
    pidm_split3_pk(v1__[2], v1__[3], q0__[3], q1__[3], q2__[3]);                                                   \
    u32x4* d__ = reinterpret_cast<u32x4*>((buf_) + a_lds[k_]);                                                     \
    d__[0] = u32x4{q0__[0], q0__[1], q0__[2], q0__[3]};                                                            \
    d__[1] = u32x4{q1__[0], q1__[1], q1__[2], q1__[3]};                                                            \
    d__[2] = u32x4{q2__[0], q2__[1], q2__[2], q2__[3]};                                                            \
  }
    // one producer iteration = consumer stage s_: weight slab of stage s_+1 (LDS-direct), fetch of stage s_+2 into set (s_ & 1)
    // (free since the previous iteration), then stage s_+1 from set ((s_+1) & 1) into the other buffer
#define PIDM_WS_ITER(s_, setL_, setW_)                                                                             \
  {                                                                                                                \
    const char* wn1__ = l_wn;                                                                                      \
    if (tr_p >= 0 && (s_) < 32) g_stream_trace[tr_p + 4 * (s_) + 0] = clock64();                                   \
    _Pragma("unroll") for (int k = 0; k < NB; ++k) PIDM_WS_COPY_B(k, wn1__, bufn)                                  \
    __builtin_amdgcn_sched_barrier(0);   /* (issue order: NB copies, then the 2 KA loads of the fetch - the waits count on it) */ \
    split_cursor_next<MODE>(lc, NCH, CCH, g.tiles_m, tpi, ntn);                                                    \
    PIDM_WS_STAGE()                                                                                                \
    _Pragma("unroll") for (int k = 0; k < KA; ++k) PIDM_WS_LOAD_A(setL_, k)                                        \
    /* set setW_ (fetched by the previous iteration, left in flight over its barrier) has landed once at most this iteration's */ \
    /* NB copies + 2 KA loads are outstanding */                                                                   \
    if (kWsLeaveFetch) { PIDM_WAIT_VMEM_LEAVE(NB + 2 * KA); __builtin_amdgcn_sched_barrier(0); }                    \
    if (tr_p >= 0 && (s_) < 32) g_stream_trace[tr_p + 4 * (s_) + 1] = clock64();                                   \
    _Pragma("unroll") for (int k = 0; k < KA; ++k) PIDM_WS_WRITE_A(setW_, k, bufn)                                 \
    if (tr_p >= 0 && (s_) < 32) g_stream_trace[tr_p + 4 * (s_) + 2] = clock64();                                   \
    /* the LDS-direct copies have landed; the 2 KA loads of the fetch (stage s+2, consumed by the NEXT iteration) stay in flight */ \
    if (kWsLeaveFetch) PIDM_WAIT_VMEM_LEAVE(2 * KA); else PIDM_WAIT_VMEM();                                        \
    __syncthreads();                 /* buffer (s+1)&1 complete (this wave's part), buffer s&1 free */             \
    if (tr_p >= 0 && (s_) < 32) g_stream_trace[tr_p + 4 * (s_) + 3] = clock64();                                   \
    char* tswap__ = bufc; bufc = bufn; bufn = tswap__;                                                             \
  }
    // prologue: stage 0 into bufc; the activations of stage 1 stay in registers (set 1)
    PIDM_WS_STAGE()
#pragma unroll
    for (int k = 0; k < KA; ++k) PIDM_WS_LOAD_A(0, k)
#pragma unroll
    for (int k = 0; k < NB; ++k) PIDM_WS_COPY_B(k, l_wn, bufc)
    if (kWsLeaveFetch) { PIDM_WAIT_VMEM_LEAVE(NB); __builtin_amdgcn_sched_barrier(0); }   // set 0 has landed (the copies are behind it)
#pragma unroll
    for (int k = 0; k < KA; ++k) PIDM_WS_WRITE_A(0, k, bufc)
    split_cursor_next<MODE>(lc, NCH, CCH, g.tiles_m, tpi, ntn);
    PIDM_WS_STAGE()
#pragma unroll
    for (int k = 0; k < KA; ++k) PIDM_WS_LOAD_A(1, k)
    if (kWsLeaveFetch) PIDM_WAIT_VMEM_LEAVE(2 * KA); else PIDM_WAIT_VMEM();
    __syncthreads();
    for (int s = 0; s < nst; s += 2) {
      PIDM_WS_ITER(s, 0, 1)
      if (s + 1 < nst) PIDM_WS_ITER(s + 1, 1, 0)
    }
#undef PIDM_WS_ITER
#undef PIDM_WS_STAGE
#undef PIDM_WS_LOAD_A
#undef PIDM_WS_COPY_B
#undef PIDM_WS_WRITE_A
    return;
  }

  // =============================== consumer waves: fragments, MFMAs, epilogue ===============================
  // MS sub-tiles (32 pixels x 32 output channels, one accumulator pair each) per consumer wave: wave w owns sub-tiles w MS .. w MS + MS - 1.
  // MS = 2 (round 6): a tap's three weight fragments are read from LDS once for both sub-tiles - 9 instead of 12 KB of fragment reads per
  // 12 MFMAs (the LDS port is the co-critical resource of these kernels: 65-73 % busy with one sub-tile per wave) - and the 256-pixel
  // tile runs on ONE consumer wave per SIMD: no second wave's MFMAs in the pipe, a stage long enough (108 MFMAs) for the producers.
  int a_frag[MS];
#pragma unroll
  for (int ms = 0; ms < MS; ++ms) {
    const int pm = (wave * MS + ms) * 32 + l31;
    const int a_tx = pm & (g.Wv - 1), a_ty = (pm >> g.wsh) & (g.TH - 1), a_img = pm >> (g.wsh + g.tsh);
    a_frag[ms] = ((a_img < g.NI) ? (a_img * g.IHt + a_ty) * rowB + a_tx * RB : 0) + 48 * half;
  }
  const int b_frag = b_reg + l31 * RB + 48 * half;
  SplitCursor cs;                    // the stage being computed
  split_cursor_init<MODE>(cs, item0, nst, g.tiles_m, tpi, ntn);
  __syncthreads();                   // stage 0 is in bufc
  f32x16 acc[MS], accb[MS];
#pragma unroll
  for (int ms = 0; ms < MS; ++ms)
    for (int r = 0; r < 16; ++r) { acc[ms][r] = 0.f; accb[ms][r] = 0.f; }
  const int tr_c = (trace && blockIdx.x == 0 && lane == 0 && wave == 0) ? 0 : -1;
  for (int s = 0; s < nst; ++s) {
    if (tr_c >= 0 && s < 32) g_stream_trace[tr_c + 4 * s + 0] = clock64();
    const float bv_pre = (bias ? bias : reinterpret_cast<const float*>(ws))[cs.tn * 32 + l31];
    int oy0 = 0, ox0 = 0;
    if (MODE == 1) {
      oy0 = 1 - g.ph_pad_y[cs.ph]; ox0 = 1 - g.ph_pad_x[cs.ph];
    } else if (MODE == 2) {
      oy0 = 1 - g.pad_y[cs.zz]; ox0 = 1 - g.pad_x[cs.zz];
    }
    const char* afp = bufc + ((MODE == 0) ? 0 : oy0 * rowB + ox0 * RB);
    const char* bfp = bufc + b_frag;
    // PF = taps the fragment reads run ahead of the MFMAs that use them (PF + 1 register sets).  One consumer wave per SIMD has
    // nobody to hide an LDS round trip behind: with PF = 1 a read has the 192 cycles of one tap's six MFMAs, with PF = 2 twice that
    constexpr int PF = (MS == 1) ? PIDM_WS_PF : 1, NS = PF + 1;
    u32x4 fa[NS][MS][3], fb[NS][3];
#define PIDM_WS_FRAGS(set_, t_)                                                                                    \
  {                                                                                                                \
    const size_t to__ = (size_t)((MODE == 0) ? ((t_) / 3) * rowB + ((t_) % 3) * RB : ((t_) >> 1) * rowB + ((t_) & 1) * RB); \
    const u32x4* br__ = reinterpret_cast<const u32x4*>(bfp + (size_t)((t_)*32) * RB);                              \
    _Pragma("unroll") for (int p = 0; p < 3; ++p) fb[set_][p] = br__[p];                                           \
    _Pragma("unroll") for (int ms = 0; ms < MS; ++ms) {                                                            \
      const u32x4* ar__ = reinterpret_cast<const u32x4*>(afp + a_frag[ms] + to__);                                 \
      _Pragma("unroll") for (int p = 0; p < 3; ++p) fa[set_][ms][p] = ar__[p];                                     \
    }                                                                                                              \
  }
    PIDM_WS_FRAGS(0, 0)
    if (PF == 2) PIDM_WS_FRAGS(1, 1)
#pragma unroll
    for (int t = 0; t < T; ++t) {
      const int cur = t % NS;
      if (t + PF < T) PIDM_WS_FRAGS((t + PF) % NS, t + PF)
      // the six terms of a product smallest first; even taps -> acc, odd taps -> accb (two chains per sub-tile, summed in the epilogue:
      // conv3x3_split_kernel); the sub-tiles of a wave take turns term by term - consecutive MFMAs never wait for each other's result
#pragma unroll
      for (int q = 0; q < 6; ++q) {
        constexpr int ia[6] = {2, 0, 1, 1, 0, 0}, ib[6] = {0, 2, 1, 0, 1, 0};
#pragma unroll
        for (int ms = 0; ms < MS; ++ms) {
          if (t & 1) accb[ms] = pidm_mfma_bf16_32x32x16(fa[cur][ms][ia[q]], fb[cur][ib[q]], accb[ms]);
          else acc[ms] = pidm_mfma_bf16_32x32x16(fa[cur][ms][ia[q]], fb[cur][ib[q]], acc[ms]);
        }
      }
      // one fragment read per MFMA: hipcc left to itself puts the tap's 6-8 ds_read_b128 into ONE gap between two MFMAs of the same
      // accumulator chain - more than the ~5 instructions a 32-cycle MFMA hides, with nobody else on the SIMD to fill the pipe
#if PIDM_WS_INTERLEAVE
#pragma unroll
      for (int q = 0; q < 6 * MS; ++q) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                   // one MFMA
        if (t + PF < T && q < 3 + 3 * MS) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);  // one LDS read
      }
#endif
      __builtin_amdgcn_sched_barrier(0);
    }
#undef PIDM_WS_FRAGS
    if (tr_c >= 0 && s < 32) g_stream_trace[tr_c + 4 * s + 1] = clock64();
    if (cs.ch == NCH - 1) {
      const int zz = (MODE == 2) ? cs.zz : 0;
      const int b0 = cs.bi * g.NI, vy0 = cs.rt * g.TH, n0 = cs.tn * 32;
      const int c = n0 + l31;
      const float bv = bias ? bv_pre : 0.f;
#pragma unroll
      for (int ms = 0; ms < MS; ++ms) {
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[ms][r] += accb[ms][r];
      const int p0 = (wave * MS + ms) * 32;
      const int tx0 = p0 & (g.Wv - 1), ty0 = (p0 >> g.wsh) & (g.TH - 1), img0 = p0 >> (g.wsh + g.tsh);
      const int b = b0 + img0;
      if (b < g.B && img0 < g.NI) {        // wave-uniform
        const int pin = (vy0 + ty0) * g.Wv + tx0;
        float v[16];
        float gs1 = 0.f, gs2 = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          v[r] = acc[ms][r] + bv;
          gs1 += v[r];
          gs2 += v[r] * v[r];
        }
        if (g.gn_part) PIDM_GN_PARTIAL(gs1, gs2, b, pin, c)
        if (g.bn_part) {
          const float* xrow = g.bn_x + ((size_t)b * g.Ho * g.Wo + pin) * g.Cout + c;
          PIDM_BN_PARTIAL(acc[ms], bv, b, pin, c, xrow, g.Cout, (g.bn_res && residual != nullptr), residual + ((size_t)b * g.Ho * g.Wo + pin) * g.ldr + c, g.ldr)
        }
        const bool odd1 = (l31 & 1) != 0, odd2 = (l31 & 2) != 0;
        const size_t opix = (size_t)b * g.sob + (size_t)pin * g.sox + n0 + 4 * (l31 >> 2);
        const size_t rpix = ((size_t)b * g.Ho * g.Wo + pin) * g.ldr + n0 + 4 * (l31 >> 2);
#pragma unroll
        for (int q4 = 0; q4 < 4; ++q4) {
          float x0 = v[4 * q4], x1 = v[4 * q4 + 1], x2 = v[4 * q4 + 2], x3 = v[4 * q4 + 3];
          const float r01 = pidm_quad_xor1(odd1 ? x0 : x1), r23 = pidm_quad_xor1(odd1 ? x2 : x3);
          x0 = odd1 ? r01 : x0; x1 = odd1 ? x1 : r01;
          x2 = odd1 ? r23 : x2; x3 = odd1 ? x3 : r23;
          const float r02 = pidm_quad_xor2(odd2 ? x0 : x2), r13 = pidm_quad_xor2(odd2 ? x1 : x3);
          x0 = odd2 ? r02 : x0; x2 = odd2 ? x2 : r02;
          x1 = odd2 ? r13 : x1; x3 = odd2 ? x3 : r13;
          const int prow = 8 * q4 + 4 * half + (l31 & 3);
          f32x4 o = {x0, x1, x2, x3};
          if (MODE == 2) {
            const int pp = p0 + prow, vx = pp & (g.Wv - 1), vy = vy0 + ((pp >> g.wsh) & (g.TH - 1));
            const size_t opx = (size_t)(2 * vy + g.ooy[zz]) * g.Wo + 2 * vx + g.oox[zz];
            if (residual) o += *reinterpret_cast<const f32x4*>(residual + ((size_t)b * g.Ho * g.Wo + opx) * g.ldr + n0 + 4 * (l31 >> 2));
            *reinterpret_cast<f32x4*>(out + (size_t)b * g.sob + opx * g.sox + n0 + 4 * (l31 >> 2)) = o;
          } else {
            if (residual) o += *reinterpret_cast<const f32x4*>(residual + rpix + (size_t)prow * g.ldr);
            *reinterpret_cast<f32x4*>(out + opix + (size_t)prow * g.sox) = o;
          }
        }
      }
      for (int r = 0; r < 16; ++r) { acc[ms][r] = 0.f; accb[ms][r] = 0.f; }
      }
    }
    if (tr_c >= 0 && s < 32) g_stream_trace[tr_c + 4 * s + 2] = clock64();
    split_cursor_next<MODE>(cs, NCH, CCH, g.tiles_m, tpi, ntn);
    __syncthreads();               // buffer (s+1)&1 complete, buffer s&1 free
    if (tr_c >= 0 && s < 32) g_stream_trace[tr_c + 4 * s + 3] = clock64();
    char* tswap = bufc; bufc = bufn; bufn = tswap;
  }
}

// ---------------------------------------------------------------------------------------------------
// 1x1 / stride-1 convolution = a GEMM [pixels] x [Cin] x [Cout] in the split form (to_qkv / to_out of the qkv-form attention levels,
// res_conv, the linears of wide models - and their input gradients, which are 1x1 convolutions with the transposed weight).  The fp32
// implicit-GEMM kernel ran these at 40-65 TFLOP/s where they are compute bound (16x16 128 -> 768: 49 us).
// Nothing is shared between neighbouring pixels here, so the reuse has to come from the OUTPUT width: a workgroup owns 128 pixels x
// NTG * 32 output channels, each of the 4 consumer waves (one per SIMD) 32 pixels x NTG accumulator tiles, and an activation
// fragment is read from LDS once per NTG * 6 MFMAs.  A stage = 32 input channels (two k-steps): 12 NTG MFMAs per consumer wave;
// the 4 producer waves stage the 128 x 32 activation block (fetched a stage ahead, split, [k-step][pixel] rows of 112 bytes) and
// copy the stage's pre-split weight slab ([k-step][sub-tile][32 rows], pack layout of 1x1 tensors) with global_load_lds, as in
// conv3x3_split_ws_kernel.  Two accumulator chains per tile (even / odd stages), summed in the epilogue.
// Pixels are addressed flat (input p * ld, output p * sox): contiguous channels-last tensors, pixel count a multiple of 128.
// ---------------------------------------------------------------------------------------------------
// NTGP = tiles per group in the PACKED piece layout (4 or 2), NTG <= NTGP the tiles a workgroup takes: launches with fewer than one
// item per CU at NTG = 4 run with NTG = 2 on the same packing (a stage's weights are then two runs of 7 KB, one per k-step).
// NPW = producer waves (4 or 8: see conv3x3_split_ws_kernel; a 1x1 stage has only 12 NTG MFMAs per consumer wave to hide them under).
template <int NTG, int NTGP, int NPW>
__global__ void __launch_bounds__(256 + 64 * NPW) conv1x1_split_kernel(ConvGeom g, const float* __restrict__ src0, const float* __restrict__ src1,
                                                            const unsigned short* __restrict__ ws, const float* __restrict__ bias,
                                                            const float* __restrict__ residual, float* __restrict__ out, int n_items,
                                                            int items_per_wg, int tiles_m) {
  constexpr int RB = kSplitRow, NW = 4, NPT = 64 * NPW, KS = 2;
  constexpr int UA = 512 / NPT;                         // staging units (8 channels of a pixel; 512 per stage) per producer thread
  constexpr int ABYTES = KS * 128 * RB;                 // activation block of a stage
  constexpr int SLAB = KS * NTG * 32 * RB;              // weights of a stage in LDS: [k-step][tile][32 rows]
  constexpr int KRUN = NTG * 32 * RB;                   // ... of one k-step: a whole number of KB (14 / 7), contiguous in the packing
  constexpr int NPIECE = KRUN / 1024;
  constexpr int NCOPY = KS * ((NPIECE + NPW - 1) / NPW);   // LDS-direct copies per producer wave and stage
  static_assert(NTG == 2 || NTG == 4, "a k-step's run must be a whole number of KB");
  static_assert(NTGP % NTG == 0, "a workgroup's tiles lie inside one packed group");
  constexpr int BUFSZ = ABYTES + SLAB;
  HIP_DYNAMIC_SHARED(float, smemf)
  char* smem = reinterpret_cast<char*>(smemf);
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int half = lane >> 5, l31 = lane & 31;
  const int NCH = g.Cin >> 5;                           // stages per item
  const int item0 = split_vblock(g.xcd) * items_per_wg;
  const int my_items = (item0 + items_per_wg <= n_items) ? items_per_wg : n_items - item0;
  const int nst = my_items * NCH;
  char* bufc = smem;
  char* bufn = smem + BUFSZ;
  // position of a stage: chunk ch of item (n-group tq, pixel tile tm), advanced incrementally
  int c_ch = 0, c_tq = item0 / tiles_m, c_tm = item0 - c_tq * tiles_m, c_left = nst - 1;
#define PIDM_G1_NEXT()                                                                                             \
  if (c_left > 0) {                                                                                                \
    --c_left;                                                                                                      \
    if (++c_ch == NCH) {                                                                                           \
      c_ch = 0;                                                                                                    \
      if (++c_tm == tiles_m) { c_tm = 0; ++c_tq; }                                                                 \
    }                                                                                                              \
  }
  if (wave >= NW) {
    // =============================== producer waves ===============================
    const int pt = tid - 64 * NW, pw = wave - NW;
    // unit u = pt + 256 k: 8 channels (group q of the stage's 32) of pixel px
    int a_lds[UA];
    unsigned a_go[UA];
#pragma unroll
    for (int k = 0; k < UA; ++k) {
      const int u = pt + NPT * k, px = u >> 2, q = u & 3;
      a_lds[k] = ((q >> 1) * 128 + px) * RB + 48 * (q & 1);
      a_go[k] = (unsigned)(px * g.ld0 + 8 * q) * 4u;
    }
    const unsigned b_lane = 16u * lane;
    f32x4 ra[2][UA][2];
    const char* l_sp = reinterpret_cast<const char*>(src0);
    const char* l_wn = reinterpret_cast<const char*>(ws);
#define PIDM_G1_STAGE()                                                                                            \
  {                                                                                                                \
    const int c0__ = c_ch * 32;                                                                                    \
    l_sp = reinterpret_cast<const char*>((c0__ < g.C0) ? src0 + c0__ : src1 + (c0__ - g.C0)) +                     \
           (size_t)c_tm * 128 * (size_t)g.ld0 * 4;                                                                 \
    /* k-step 0 of the stage: packed group (c_tq NTG) / NTGP, first tile (c_tq NTG) % NTGP; k-step 1 is NTGP * 32 rows further */ \
    l_wn = reinterpret_cast<const char*>(ws) +                                                                     \
           ((((size_t)((c_tq * NTG) / NTGP) * (2 * NCH) + 2 * c_ch) * NTGP + (c_tq * NTG) % NTGP) * 32) * RB;       \
  }
#define PIDM_G1_LOAD_A(set_)                                                                                       \
  _Pragma("unroll") for (int k = 0; k < UA; ++k) {                                                                  \
    const char* p__ = l_sp + a_go[k];                                                                              \
    if (kWsLeaveFetch) {                                                                                           \
      PIDM_UNTRACKED_LOAD_F32X4(ra[set_][k][0], p__, 0);                                                           \
      PIDM_UNTRACKED_LOAD_F32X4(ra[set_][k][1], p__, 16);                                                          \
    } else {                                                                                                       \
      ra[set_][k][0] = reinterpret_cast<const f32x4*>(p__)[0];                                                     \
      ra[set_][k][1] = reinterpret_cast<const f32x4*>(p__)[1];                                                     \
    }                                                                                                              \
  }
  /* every producer wave issues the SAME number of copies (NCOPY: the waits count them): a wave whose last piece index would  */ \
  /* lie beyond the run repeats the run's last piece - the same bytes to the same place                                         */ \
#define PIDM_G1_COPY_B(wn_, buf_)                                                                                  \
  _Pragma("unroll") for (int ks__ = 0; ks__ < KS; ++ks__)                                                          \
    _Pragma("unroll") for (int k = 0; k < (NPIECE + NPW - 1) / NPW; ++k) {                                         \
      const int pc__ = (pw + NPW * k < NPIECE) ? pw + NPW * k : NPIECE - 1;                                        \
      if (kWsLeaveFetch)                                                                                           \
        pidm_glds_b128_untracked((wn_) + (size_t)ks__ * NTGP * 32 * RB + 1024 * pc__ + b_lane,                     \
                                 (buf_) + ABYTES + ks__ * KRUN + 1024 * pc__);                                     \
      else if (pw + NPW * k < NPIECE)                                                                              \
        pidm_glds_b128((wn_) + (size_t)ks__ * NTGP * 32 * RB + 1024 * pc__ + b_lane,                               \
                       (buf_) + ABYTES + ks__ * KRUN + 1024 * pc__);                                               \
    }
#define PIDM_G1_WRITE_A(set_, buf_)                                                                                \
  _Pragma("unroll") for (int k = 0; k < UA; ++k) {                                                                  \
    const f32x4 v0__ = ra[set_][k][0], v1__ = ra[set_][k][1];                                                      \
    unsigned q0__[4], q1__[4], q2__[4];                                                                            \
    pidm_split3_pk(v0__[0], v0__[1], q0__[0], q1__[0], q2__[0]);                                                   \
    pidm_split3_pk(v0__[2], v0__[3], q0__[1], q1__[1], q2__[1]);                                                   \
    pidm_split3_pk(v1__[0], v1__[1], q0__[2], q1__[2], q2__[2]);                                                   \
    pidm_split3_pk(v1__[2], v1__[3], q0__[3], q1__[3], q2__[3]);                                                   \
    u32x4* d__ = reinterpret_cast<u32x4*>((buf_) + a_lds[k]);                                                      \
    d__[0] = u32x4{q0__[0], q0__[1], q0__[2], q0__[3]};                                                            \
    d__[1] = u32x4{q1__[0], q1__[1], q1__[2], q1__[3]};                                                            \
    d__[2] = u32x4{q2__[0], q2__[1], q2__[2], q2__[3]};                                                            \
  }
    // one producer iteration = consumer stage s: weight slab of stage s+1, fetch of stage s+2 into set s & 1, stage s+1 from the other set
#define PIDM_G1_ITER(setL_, setW_)                                                                                 \
  {                                                                                                                \
    const char* wn1__ = l_wn;                                                                                      \
    PIDM_G1_COPY_B(wn1__, bufn)                                                                                    \
    __builtin_amdgcn_sched_barrier(0);   /* (issue order: NCOPY copies, then the 2 UA loads of the fetch - as in conv3x3_split_ws_kernel) */ \
    PIDM_G1_NEXT()                                                                                                 \
    PIDM_G1_STAGE()                                                                                                \
    PIDM_G1_LOAD_A(setL_)                                                                                          \
    if (kWsLeaveFetch) { PIDM_WAIT_VMEM_LEAVE(NCOPY + 2 * UA); __builtin_amdgcn_sched_barrier(0); }   /* set setW_ has landed */ \
    PIDM_G1_WRITE_A(setW_, bufn)                                                                                   \
    if (kWsLeaveFetch) PIDM_WAIT_VMEM_LEAVE(2 * UA); else PIDM_WAIT_VMEM();   /* the copies have landed, the fetch stays in flight */ \
    __syncthreads();                                                                                               \
    char* tswap__ = bufc; bufc = bufn; bufn = tswap__;                                                             \
  }
    PIDM_G1_STAGE()
    PIDM_G1_LOAD_A(0)
    PIDM_G1_COPY_B(l_wn, bufc)
    if (kWsLeaveFetch) { PIDM_WAIT_VMEM_LEAVE(NCOPY); __builtin_amdgcn_sched_barrier(0); }
    PIDM_G1_WRITE_A(0, bufc)
    PIDM_G1_NEXT()
    PIDM_G1_STAGE()
    PIDM_G1_LOAD_A(1)
    if (kWsLeaveFetch) PIDM_WAIT_VMEM_LEAVE(2 * UA); else PIDM_WAIT_VMEM();
    __syncthreads();
    for (int s = 0; s < nst; s += 2) {
      PIDM_G1_ITER(0, 1)
      if (s + 1 < nst) PIDM_G1_ITER(1, 0)
    }
#undef PIDM_G1_STAGE
#undef PIDM_G1_LOAD_A
#undef PIDM_G1_COPY_B
#undef PIDM_G1_WRITE_A
#undef PIDM_G1_ITER
    return;
  }
  // =============================== consumer waves ===============================
  const int a_frag = (wave * 32 + l31) * RB + 48 * half;
  const int b_frag = ABYTES + l31 * RB + 48 * half;
  __syncthreads();                   // stage 0 is in bufc
  f32x16 acc[NTG][2];
#pragma unroll
  for (int j = 0; j < NTG; ++j)
    for (int r = 0; r < 16; ++r) { acc[j][0][r] = 0.f; acc[j][1][r] = 0.f; }
  for (int s = 0; s < nst; ++s) {
    // fragments one (k-step, tile) pair ahead of the MFMAs that use them
    u32x4 fa[2][3], fb[2][3];
#define PIDM_G1_FRAG_A(set_, ks_)                                                                                  \
  {                                                                                                                \
    const u32x4* ar__ = reinterpret_cast<const u32x4*>(bufc + a_frag + (ks_) * 128 * RB);                          \
    _Pragma("unroll") for (int p = 0; p < 3; ++p) fa[set_][p] = ar__[p];                                           \
  }
#define PIDM_G1_FRAG_B(set_, i_)                                                                                   \
  {                                                                                                                \
    const u32x4* br__ = reinterpret_cast<const u32x4*>(bufc + b_frag + ((i_) * 32) * RB);                          \
    _Pragma("unroll") for (int p = 0; p < 3; ++p) fb[set_][p] = br__[p];                                           \
  }
    PIDM_G1_FRAG_A(0, 0)
    PIDM_G1_FRAG_B(0, 0)
#pragma unroll
    for (int i = 0; i < KS * NTG; ++i) {          // i = ks * NTG + j
      const int ks = i / NTG, j = i - ks * NTG, cb = i & 1, ca = ks & 1;
      if (i + 1 < KS * NTG) {
        PIDM_G1_FRAG_B(cb ^ 1, i + 1)
        if ((i + 1) % NTG == 0) PIDM_G1_FRAG_A(ca ^ 1, ks + 1)
      }
      // small terms first (as in conv3x3_split_kernel); chain = stage parity
      if (s & 1) {
        acc[j][1] = pidm_mfma_bf16_32x32x16(fa[ca][2], fb[cb][0], acc[j][1]);
        acc[j][1] = pidm_mfma_bf16_32x32x16(fa[ca][0], fb[cb][2], acc[j][1]);
        acc[j][1] = pidm_mfma_bf16_32x32x16(fa[ca][1], fb[cb][1], acc[j][1]);
        acc[j][1] = pidm_mfma_bf16_32x32x16(fa[ca][1], fb[cb][0], acc[j][1]);
        acc[j][1] = pidm_mfma_bf16_32x32x16(fa[ca][0], fb[cb][1], acc[j][1]);
        acc[j][1] = pidm_mfma_bf16_32x32x16(fa[ca][0], fb[cb][0], acc[j][1]);
      } else {
        acc[j][0] = pidm_mfma_bf16_32x32x16(fa[ca][2], fb[cb][0], acc[j][0]);
        acc[j][0] = pidm_mfma_bf16_32x32x16(fa[ca][0], fb[cb][2], acc[j][0]);
        acc[j][0] = pidm_mfma_bf16_32x32x16(fa[ca][1], fb[cb][1], acc[j][0]);
        acc[j][0] = pidm_mfma_bf16_32x32x16(fa[ca][1], fb[cb][0], acc[j][0]);
        acc[j][0] = pidm_mfma_bf16_32x32x16(fa[ca][0], fb[cb][1], acc[j][0]);
        acc[j][0] = pidm_mfma_bf16_32x32x16(fa[ca][0], fb[cb][0], acc[j][0]);
      }
#if PIDM_WS_INTERLEAVE
      // one fragment read per MFMA (see conv3x3_split_ws_kernel): 3 reads of the next tile's weights, 3 more when the k-step changes
#pragma unroll
      for (int q = 0; q < 6; ++q) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        if (i + 1 < KS * NTG && (q < 3 || (i + 1) % NTG == 0)) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
      }
#endif
      __builtin_amdgcn_sched_barrier(0);
    }
#undef PIDM_G1_FRAG_A
#undef PIDM_G1_FRAG_B
    if (c_ch == NCH - 1) {
      // ---- item done: bias, 4x4 register transposes, residual, 16-byte stores (as in conv3x3_split_kernel).  All residual rows of
      //      the wave's NTG tiles are requested first: one memory latency per item instead of one per tile ----
      const size_t pbase = (size_t)c_tm * 128 + wave * 32;
      const bool odd1 = (l31 & 1) != 0, odd2 = (l31 & 2) != 0;
      f32x4 rres[NTG][4];
      if (residual) {
#pragma unroll
        for (int j = 0; j < NTG; ++j)
#pragma unroll
          for (int q4 = 0; q4 < 4; ++q4)
            rres[j][q4] = *reinterpret_cast<const f32x4*>(residual + (pbase + 8 * q4 + 4 * half + (l31 & 3)) * g.ldr +
                                                          (c_tq * NTG + j) * 32 + 4 * (l31 >> 2));
      }
#pragma unroll
      for (int j = 0; j < NTG; ++j) {
        const int n0 = (c_tq * NTG + j) * 32;
        const float bv = bias ? bias[n0 + l31] : 0.f;
        float v[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) v[r] = (acc[j][0][r] + acc[j][1][r]) + bv;
#pragma unroll
        for (int q4 = 0; q4 < 4; ++q4) {
          float x0 = v[4 * q4], x1 = v[4 * q4 + 1], x2 = v[4 * q4 + 2], x3 = v[4 * q4 + 3];
          const float r01 = pidm_quad_xor1(odd1 ? x0 : x1), r23 = pidm_quad_xor1(odd1 ? x2 : x3);
          x0 = odd1 ? r01 : x0; x1 = odd1 ? x1 : r01;
          x2 = odd1 ? r23 : x2; x3 = odd1 ? x3 : r23;
          const float r02 = pidm_quad_xor2(odd2 ? x0 : x2), r13 = pidm_quad_xor2(odd2 ? x1 : x3);
          x0 = odd2 ? r02 : x0; x2 = odd2 ? x2 : r02;
          x1 = odd2 ? r13 : x1; x3 = odd2 ? x3 : r13;
          const size_t p = pbase + 8 * q4 + 4 * half + (l31 & 3);
          f32x4 o = {x0, x1, x2, x3};
          if (residual) o += rres[j][q4];
          *reinterpret_cast<f32x4*>(out + p * g.sox + n0 + 4 * (l31 >> 2)) = o;
        }
        for (int r = 0; r < 16; ++r) { acc[j][0][r] = 0.f; acc[j][1][r] = 0.f; }
      }
    }
    PIDM_G1_NEXT()
    __syncthreads();               // buffer (s+1)&1 complete, buffer s&1 free
    char* tswap = bufc; bufc = bufn; bufn = tswap;
  }
#undef PIDM_G1_NEXT
}

// ---------------------------------------------------------------------------------------------------
// 7x7 / stride 1 / pad 3 convolution with very few input channels: the UNet's init_conv (reference src/unet_model.py:453,568;
// Cin = 2, or 4 with self-conditioning).  The implicit-GEMM kernels pad Cin to 8 channels PER TAP (49 taps x 8 = 392 columns for
// 98 real ones) and ran this layer at 16 TFLOP/s - 100 us for 1.6 GFLOP.  Here (kx, channel) of one kernel row is the
// contraction index: in a channels-last image the 7 x Cin values a pixel needs from input row y + ky - 3 are CONTIGUOUS, so one
// kernel row is one (Cin = 2: 14 -> 16 wide) or two (Cin = 4: 28 -> 32) k-steps of the 32x32x16 bf16 MFMA, 3-piece split operands
// as in conv3x3_split_kernel (same accuracy).  A workgroup = 8 waves = 256 output pixels x 32 output channels; the (TH + 6) input
// rows of the tile sit in LDS with a zero halo; the weights of the 32 channels live in registers, pre-split once per workgroup.
// ---------------------------------------------------------------------------------------------------
template <int CIN>
__global__ void __launch_bounds__(512) conv7x7_split_kernel(ConvGeom g, const float* __restrict__ src, const float* __restrict__ wp, int Kp,
                                                           const float* __restrict__ bias, const float* __restrict__ residual,
                                                           float* __restrict__ out, int n_tiles, int tiles_per_wg) {
  constexpr int RK = 7 * CIN, KS = (RK + 15) / 16, NK = 7 * KS;
  HIP_DYNAMIC_SHARED(float, smem)
  const int tid = threadIdx.x, lane = tid & 63, half = lane >> 5, l31 = lane & 31;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int W = g.Wv, H = g.Hv, TH = 256 / W, tpi = H / TH;
  const int RL = ((W + 6) * CIN + (16 * KS - RK) + 1) & ~1;     // floats per staged input row (reads run past the last real tap)
  const int n0 = blockIdx.y * 32;
  // ---- this lane's B fragments: output channel n0 + l31, k = 16 s + 8 half + e of kernel row ky.  The 32 channels' packed
  //      weights ([49 taps][Kp] floats each) come in through LDS with coalesced loads (56 dependent global loads per lane cost
  //      more than the whole tile loop), rows padded to an odd length so that the 32 lanes of a read hit 32 banks ----
  const int WR = (49 * Kp) | 1;
  float* wl = smem + (size_t)(TH + 6) * RL;
  for (int e = tid; e < 32 * 49 * Kp; e += 512) {
    const int n = e / (49 * Kp), r = e - n * (49 * Kp);
    wl[(size_t)n * WR + r] = wp[(size_t)(n0 + n) * 49 * Kp + r];
  }
  __syncthreads();
  u32x4 fb[NK][3];
#pragma unroll
  for (int ks = 0; ks < NK; ++ks) {
    const int ky = ks / KS, sk = ks - ky * KS;
    float wv[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int k = 16 * sk + 8 * half + e, kx = k / CIN, c = k - kx * CIN;
      wv[e] = (k < RK) ? wl[(size_t)l31 * WR + (ky * 7 + (k < RK ? kx : 0)) * Kp + c] : 0.f;
    }
    unsigned q0[4], q1[4], q2[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) pidm_split3_pk(wv[2 * e], wv[2 * e + 1], q0[e], q1[e], q2[e]);
    fb[ks][0] = u32x4{q0[0], q0[1], q0[2], q0[3]};
    fb[ks][1] = u32x4{q1[0], q1[1], q1[2], q1[3]};
    fb[ks][2] = u32x4{q2[0], q2[1], q2[2], q2[3]};
  }
  const float bv = bias ? bias[n0 + l31] : 0.f;
  const int pm = wave * 32 + l31;                   // this lane's pixel of the tile (A row)
  const int a_ty = pm / W, a_tx = pm - a_ty * W;
  const int t_first = blockIdx.x * tiles_per_wg;
  const int t_last = (t_first + tiles_per_wg < n_tiles) ? t_first + tiles_per_wg : n_tiles;
  for (int tile = t_first; tile < t_last; ++tile) {
    const int b = tile / tpi, y0 = (tile - b * tpi) * TH;
    // ---- stage rows y0 - 3 .. y0 + TH + 2 with a 3-pixel zero halo left and right (and zeros behind the last pixel) ----
    __syncthreads();                                // the previous tile's reads are done
    // (thread -> (row, float pair): RL is even and the rows are 8-byte aligned in the source: Cin = 2 or 4, ld0 = Cin)
    for (int e = tid; e < (TH + 6) * (RL >> 1); e += 512) {
      const int r = e / (RL >> 1), q = 2 * (e - r * (RL >> 1));
      const int y = y0 - 3 + r, xc = q - 3 * CIN;   // float index inside the image row
      f32x2 v = {0.f, 0.f};
      if (y >= 0 && y < H && xc >= 0 && xc < W * CIN) v = *reinterpret_cast<const f32x2*>(src + ((size_t)b * H + y) * (size_t)(W * CIN) + xc);
      *reinterpret_cast<f32x2*>(smem + (size_t)r * RL + q) = v;
    }
    __syncthreads();
    f32x16 acc, accb;
    for (int r = 0; r < 16; ++r) { acc[r] = 0.f; accb[r] = 0.f; }
#pragma unroll
    for (int ks = 0; ks < NK; ++ks) {
      const int ky = ks / KS, sk = ks - ky * KS;
      const float* ap = smem + (size_t)(a_ty + ky) * RL + a_tx * CIN + 16 * sk + 8 * half;   // 8-byte aligned
      float av[8];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const f32x2 v2 = *reinterpret_cast<const f32x2*>(ap + 2 * e);
        av[2 * e] = v2[0];
        av[2 * e + 1] = v2[1];
      }
      unsigned q0[4], q1[4], q2[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) pidm_split3_pk(av[2 * e], av[2 * e + 1], q0[e], q1[e], q2[e]);
      const u32x4 fa0 = {q0[0], q0[1], q0[2], q0[3]}, fa1 = {q1[0], q1[1], q1[2], q1[3]}, fa2 = {q2[0], q2[1], q2[2], q2[3]};
      // small terms first; two accumulators (even / odd k-steps) as in conv3x3_split_kernel
      if (ks & 1) {
        accb = pidm_mfma_bf16_32x32x16(fa2, fb[ks][0], accb);
        accb = pidm_mfma_bf16_32x32x16(fa0, fb[ks][2], accb);
        accb = pidm_mfma_bf16_32x32x16(fa1, fb[ks][1], accb);
        accb = pidm_mfma_bf16_32x32x16(fa1, fb[ks][0], accb);
        accb = pidm_mfma_bf16_32x32x16(fa0, fb[ks][1], accb);
        accb = pidm_mfma_bf16_32x32x16(fa0, fb[ks][0], accb);
      } else {
        acc = pidm_mfma_bf16_32x32x16(fa2, fb[ks][0], acc);
        acc = pidm_mfma_bf16_32x32x16(fa0, fb[ks][2], acc);
        acc = pidm_mfma_bf16_32x32x16(fa1, fb[ks][1], acc);
        acc = pidm_mfma_bf16_32x32x16(fa1, fb[ks][0], acc);
        acc = pidm_mfma_bf16_32x32x16(fa0, fb[ks][1], acc);
        acc = pidm_mfma_bf16_32x32x16(fa0, fb[ks][0], acc);
      }
    }
    // ---- epilogue: bias, 4x4 register transposes (DPP) so that a lane stores 4 consecutive channels of one pixel ----
    const bool odd1 = (l31 & 1) != 0, odd2 = (l31 & 2) != 0;
    const int p0 = wave * 32;
#pragma unroll
    for (int q4 = 0; q4 < 4; ++q4) {
      float x0 = acc[4 * q4] + accb[4 * q4] + bv, x1 = acc[4 * q4 + 1] + accb[4 * q4 + 1] + bv;
      float x2 = acc[4 * q4 + 2] + accb[4 * q4 + 2] + bv, x3 = acc[4 * q4 + 3] + accb[4 * q4 + 3] + bv;
      const float r01 = pidm_quad_xor1(odd1 ? x0 : x1), r23 = pidm_quad_xor1(odd1 ? x2 : x3);
      x0 = odd1 ? r01 : x0; x1 = odd1 ? x1 : r01;
      x2 = odd1 ? r23 : x2; x3 = odd1 ? x3 : r23;
      const float r02 = pidm_quad_xor2(odd2 ? x0 : x2), r13 = pidm_quad_xor2(odd2 ? x1 : x3);
      x0 = odd2 ? r02 : x0; x2 = odd2 ? x2 : r02;
      x1 = odd2 ? r13 : x1; x3 = odd2 ? x3 : r13;
      const int pp = p0 + 8 * q4 + 4 * half + (l31 & 3);
      const int py = pp / W, px = pp - py * W;
      f32x4 o = {x0, x1, x2, x3};
      const size_t pin = (size_t)(y0 + py) * W + px;
      if (residual) o += *reinterpret_cast<const f32x4*>(residual + ((size_t)b * H * W + pin) * g.ldr + n0 + 4 * (l31 >> 2));
      *reinterpret_cast<f32x4*>(out + (size_t)b * g.sob + pin * g.sox + n0 + 4 * (l31 >> 2)) = o;
    }
  }
}

// Which form of the split convolution kernel takes a launch (PIDM_SPLIT_WS, read per launch): unset = the warp-specialised form
// for the 4-wave / 128-pixel tile only (one consumer wave per SIMD cannot hide its own staging: 15-19 % faster there, 8x8 level
// 22.4 -> 18.4 us, 38.3 -> 31.5, 64.1 -> 51.9), the one-role form for the 8-wave tile (two waves per SIMD already overlap each
// other and the extra producer waves cost 2-3 %: 63.8 -> 65.7 us; profiles/r03_ws_conv.txt); 1 / 0 force one form everywhere.
// sub-tiles per consumer wave of the 256-pixel tile (PIDM_SPLIT_MS = 1 / 2, read per launch; the 128-pixel tile has one)
// (default 1: with two sub-tiles per wave the 256-pixel tile measured 2 % SLOWER per step than the two-role 8-wave kernel - the
// consumers reach 80-86 % of the pipe, but the producers' staging arithmetic gets one instruction per ~20 cycles beside them and
// the stage is theirs again: profiles/r06_split_conv_stage_stamps.txt)
static int split_ms() {
  const char* e = knob("PIDM_SPLIT_MS");
  return (e && atoi(e) == 2) ? 2 : 1;
}
// PIDM_SPLIT_XCD: item order of the split-form kernels follows the XCDs (split_vblock); read per launch
static int split_xcd_order() {
  const char* e = knob("PIDM_SPLIT_XCD");
  return e ? atoi(e) : 0;
}
// producer waves of the warp-specialised kernels (PIDM_SPLIT_NPW = 4 / 8, read per launch; see conv3x3_split_ws_kernel)
// default: 8 where the consumers are one wave per SIMD (4-wave tiles, the 1x1 kernel with two tiles per wave); the 8-wave tile and
// the four-tile 1x1 item keep 4 (16 waves of 128 registers / 12 of 168 would spill their accumulators)
static int split_npw(bool one_consumer_per_simd) {
  const char* e = knob("PIDM_SPLIT_NPW");
  if (!e) return one_consumer_per_simd ? 8 : 4;
  return atoi(e) == 4 ? 4 : 8;
}
template <int NW, int MODE>
static void launch_split_ws(int wgs, size_t lds, hipStream_t st, const ConvGeom& gs, const float* src0, const float* src1,
                            const unsigned short* wsplit, const float* bias, const float* residual, float* out, int n_items, int ipw,
                            int trace) {
  static bool attr = false;
  if (!attr) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3_split_ws_kernel<NW, MODE, 4, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 256);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3_split_ws_kernel<NW, MODE, 8, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 256);
    if (NW == 8) {
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3_split_ws_kernel<8, MODE, 4, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 256);
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3_split_ws_kernel<8, MODE, 8, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 256);
    }
    attr = true;
  }
  const int ms = (NW == 8) ? split_ms() : 1;
  const int npw = split_npw(NW == 4 || ms == 2);
#define PIDM_WS_LAUNCH(NW_, NPW_, MS_)                                                                                               \
  hipLaunchKernelGGL(HIP_KERNEL_NAME(conv3x3_split_ws_kernel<NW_, MODE, NPW_, MS_>), dim3(wgs), dim3(64 * (NW_ / MS_) + 64 * NPW_), lds, st, gs, src0, \
                     src1, wsplit, bias, residual, out, n_items, ipw, trace)
  if (NW == 8 && ms == 2) {
    if (npw == 8) PIDM_WS_LAUNCH(8, 8, 2); else PIDM_WS_LAUNCH(8, 4, 2);
  } else {
    if (npw == 8) PIDM_WS_LAUNCH(NW, 8, 1); else PIDM_WS_LAUNCH(NW, 4, 1);
  }
#undef PIDM_WS_LAUNCH
}
static bool split_ws_on(int nw) {
  const char* e = knob("PIDM_SPLIT_WS");
  if (!e) return nw == 4 || split_ms() == 2;   // (round 6: the 256-pixel tile as well, with two sub-tiles per consumer wave)
  return atoi(e) != 0;
}
// pre-split weights of a 3x3 convolution: [Cout/32][Cin/16][9 taps][32 rows][2 halves][3 pieces][8 channels] bf16, behind the
// fp32 packing of the same tensor (packed_floats counts both).  Shape-only condition: the launcher may still take another kernel.
static bool split_shape_ok(const ConvGeom& g) {
  return g.KH == 3 && g.KW == 3 && g.stride == 1 && g.nz == 1 && g.nph == 1 && g.os == 1 && (g.Cin % 32 == 0) && (g.Cout % 32 == 0);
}
// the 4x4 / stride-2 family as 2x2 taps: 4 K-phases (nph = 4, one matrix with K = 4 Cin) or 4 output parities (nz = 4, four slabs)
static bool split_shape_ok2(const ConvGeom& g) {
  return g.KH == 2 && g.KW == 2 && g.stride == 1 && ((g.nph == 4 && g.nz == 1) || (g.nz == 4 && g.nph == 1)) && (g.Cin % 32 == 0) &&
         (g.Cout % 32 == 0);
}
// 1x1 / stride-1 tensors for conv1x1_split_kernel: whole 32-channel chunks in, groups of 4 / 2 / 1 32-channel tiles out
static int split_ntg(const ConvGeom& g) {
  // (not the linears - 1x1 "images": the concatenated FiLM linear is packed in sub-blocks that carry no pieces, and a batch is too
  // few pixels for this tile anyway)
  if (!(g.KH == 1 && g.KW == 1 && g.stride == 1 && g.nz == 1 && g.nph == 1 && g.os == 1 && (g.Cin % 32 == 0) && (g.Cout % 64 == 0) &&
        g.Cin >= 64 && g.Hv * g.Wv > 1))
    return 0;
  const bool on = [] { const char* e = knob("PIDM_CONV1X1_SPLIT"); return !(e && !atoi(e)); }();
  if (!on) return 0;
  return (g.Cout % 128 == 0) ? 4 : 2;
}
// pieces of packed element (parity slab z, row n, tap t, column k): [z][n / 32][k / 16][T taps][32 rows][2 halves][3 pieces][8 + pad];
// ntg > 1 (1x1 tensors, T = 1): the 32-row tiles of a group of ntg are adjacent - [n / (32 ntg)][k / 16][tile in group][32 rows]...
__device__ __forceinline__ size_t split_row_index(int nch, int T, int ntn, int ntg, int z, int ntile, int chunk, int t) {
  return ((((size_t)z * (ntn / ntg) + ntile / ntg) * nch + chunk) * T + t) * ntg + (ntile % ntg);
}
__device__ __forceinline__ void split_store(unsigned short* ws, int nch, int T, int ntn, int ntg, int z, int n, int t, int k, float v) {
  unsigned p0, p1, p2;
  pidm_split3_pk(v, 0.f, p0, p1, p2);
  const size_t o = (split_row_index(nch, T, ntn, ntg, z, n >> 5, k >> 4, t) * 32 + (n & 31)) * (kSplitRow / 2) + ((k >> 3) & 1) * 24 + (k & 7);
  ws[o] = (unsigned short)(p0 & 0xffffu);
  ws[o + 8] = (unsigned short)(p1 & 0xffffu);
  ws[o + 16] = (unsigned short)(p2 & 0xffffu);
}

// ---------------------------------------------------------------------------------------------------
// weight packing: reference layout -> [nz][Np][T][Kp] (zero padded)
//   kind 0: fwd, normal conv        src [N=Cout][K=Cin][KH][KW]
//   kind 1: fwd, transposed 4x4s2   src [K=Cin][N=Cout][4][4], 4 parity classes of 2x2 taps
//   kind 2: dgrad of normal s1      src [K=Cout][N=Cin][KH][KW]  (flip taps)
//   kind 3: dgrad of normal 4x4 s2  src [K=Cout][N=Cin][4][4], 4 parity classes of 2x2 taps
//   kind 4: dgrad of transposed     src [N=Cin][K=Cout][4][4]   (stride-2 conv over dy)
// ---------------------------------------------------------------------------------------------------
__device__ __forceinline__ int parity_tap(int par, int j) { return par == 0 ? 3 - 2 * j : 2 - 2 * j; }

// iterates over the SOURCE-valid elements (n < N, k < K) only; padding is zero-filled once by the caller.
// (n_off, k_off) place a source tensor inside a larger packed matrix (concatenated time-MLP linears).
__global__ void pack_kernel(const float* __restrict__ src, float* __restrict__ dst, int kind, int nz, int N, int K,
                            int Np, int Kp, int KH, int KW, int T, int n_off, int k_off, unsigned short* __restrict__ split, int nch, int ntn,
                            int ntg) {
  const size_t total = (size_t)nz * N * T * K;
  for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
    const int k = (int)(idx % K);
    const int t = (int)((idx / K) % T);
    const int n = (int)((idx / ((size_t)K * T)) % N);
    const int z = (int)(idx / ((size_t)K * T * N));
    float v;
    if (kind == 5 || kind == 6) {
      // phased 4x4s2 (z = phase (py,px)): tap j of phase parity p is source tap (p == 0 ? 1 + 2j : 2j); one packed matrix
      const int py = z >> 1, px = z & 1, jy = t >> 1, jx = t & 1;
      const int ky = py == 0 ? 1 + 2 * jy : 2 * jy, kx = px == 0 ? 1 + 2 * jx : 2 * jx;
      v = (kind == 5) ? src[(((size_t)n * K + k) * 4 + ky) * 4 + kx]     // strided conv  W[n][k][ky][kx]
                      : src[(((size_t)n * K + k) * 4 + ky) * 4 + kx];    // dgrad of convT Wt[n][k][ky][kx]
      dst[(((size_t)(n_off + n)) * T + t) * Kp + k_off + z * K + k] = v;
      if (split) split_store(split, nch, T, ntn, ntg, 0, n_off + n, t, k_off + z * K + k, v);
      continue;
    }
    if (kind == 0) {
      v = src[((size_t)n * K + k) * T + t];
    } else if (kind == 2) {
      const int ky = KH - 1 - t / KW, kx = KW - 1 - t % KW;
      v = src[(((size_t)k * N + n) * KH + ky) * KW + kx];
    } else if (kind == 4) {
      v = src[((size_t)n * K + k) * T + t];
    } else {  // parity kinds: T == 4 (2x2), source taps 4x4
      const int ky = parity_tap(z >> 1, t >> 1), kx = parity_tap(z & 1, t & 1);
      v = src[(((size_t)k * N + n) * 4 + ky) * 4 + kx];
    }
    dst[(((size_t)z * Np + n_off + n) * T + t) * Kp + k_off + k] = v;
    if (split) split_store(split, nch, T, ntn, ntg, z, n_off + n, t, k_off + k, v);
  }
}

// all weight tensors of the model in ONE launch: a device-side descriptor table.
//
// TILED tensors (PackDesc::tiled: kinds 0 / 2 / 4 with 1 or 9 taps - the 3x3 and 1x1 convolutions and the linears, > 80 % of the
// weights): a block owns 32 packed rows x 32 packed columns x all taps.  The source is read as 32 contiguous runs of 32 T floats
// (rows of the [N][K][T] tensor for the forward packing, of the [K][N][T] tensor for the input-gradient packing, whose taps are
// flipped on their way into LDS), the packing leaves as whole 128-byte lines and the bf16 pieces as whole 3.5 KB images of 32 LDS
// rows.  The element-wise form below read 4 bytes per lane at a stride of 36 bytes (forward) or 36 KB (input gradient) and
// scattered the pieces as 2-byte stores: 0.9 TB/s on the 130 M-parameter mechanics model (2.3 ms per step).
//
// Everything else (4x4 / stride-2 tensors in their parity / phase forms, the 7x7 init convolution): element-wise, each block
// 2048 elements, index arithmetic in 32 bits and hierarchical (three unsigned divisions per element instead of four 64-bit
// divisions and three remainders), two consecutive k per thread where K is even.
__device__ __forceinline__ size_t pack_src_index(const PackDesc& d, unsigned z, unsigned n, unsigned t, unsigned k) {
  if (d.kind == 5 || d.kind == 6) {
    const unsigned py = z >> 1, px = z & 1, jy = t >> 1, jx = t & 1;
    const unsigned ky = py == 0 ? 1 + 2 * jy : 2 * jy, kx = px == 0 ? 1 + 2 * jx : 2 * jx;
    return (((size_t)n * d.K + k) * 4 + ky) * 4 + kx;
  }
  if (d.kind == 0 || d.kind == 4) return ((size_t)n * d.K + k) * d.T + t;
  if (d.kind == 2) {
    const unsigned ky = d.KH - 1 - t / d.KW, kx = d.KW - 1 - t % d.KW;
    return (((size_t)k * d.N + n) * d.KH + ky) * d.KW + kx;
  }
  const int ky = parity_tap(z >> 1, t >> 1), kx = parity_tap(z & 1, t & 1);
  return (((size_t)k * d.N + n) * 4 + ky) * 4 + kx;
}
// source stride between k and k + 1
__device__ __forceinline__ size_t pack_src_kstride(const PackDesc& d) {
  if (d.kind == 5 || d.kind == 6) return 16;
  if (d.kind == 0 || d.kind == 4) return (size_t)d.T;
  if (d.kind == 2) return (size_t)d.N * d.KH * d.KW;
  return (size_t)d.N * 16;
}

static constexpr int kPackTileT = 9;       // most taps a tiled tensor has (LDS: 32 x 33 x 9 floats)
__device__ __forceinline__ void pack_tile(const PackDesc& d, unsigned tile_id, float* __restrict__ tile) {
  const unsigned T = (unsigned)d.T, N = (unsigned)d.N, K = (unsigned)d.K;
  const unsigned ktiles = (K + 31u) >> 5;
  const unsigned nt = tile_id / ktiles, kt = tile_id - nt * ktiles;
  const unsigned n0 = nt * 32u, k0 = kt * 32u;
  const unsigned tid = threadIdx.x, RUN = 32u * T;
  const unsigned magicT = (T > 1) ? (unsigned)((0x100000000ULL + T - 1) / T) : 0u;     // e / T == __umulhi(e, magicT) for e * T < 2^32
  const unsigned magicR = (unsigned)((0x100000000ULL + RUN - 1) / RUN);
  // Whole tiles of 16-byte aligned tensors move four floats per lane (round 6: the scalar loops below were 36 + 36 dependent
  // iterations per thread, the kernel ran at 1.3 TB/s and cost the mechanics step 1.4 ms); edge tiles and odd shapes keep them.
  const bool whole = n0 + 32u <= N && k0 + 32u <= K;
  const bool vec_src = whole && ((reinterpret_cast<size_t>(d.src) & 15) == 0) && ((((d.kind == 2) ? N : K) * T) & 3u) == 0;
  const bool vec_dst = whole && ((reinterpret_cast<size_t>(d.dst) & 15) == 0) && ((unsigned)d.Kp & 3u) == 0 && ((unsigned)d.k_off & 3u) == 0;
  // ---- source tile -> LDS as tile[(nl * 33 + kl) * T + t] (33: the piece pass reads 32 rows nl at one kl) ----
  if (vec_src && d.kind == 2) {
    for (unsigned e = 4u * tid; e < 32u * RUN; e += 1024u) {
      const unsigned kl = __umulhi(e, magicR), r = e - kl * RUN;
      const f32x4 v = *reinterpret_cast<const f32x4*>(d.src + ((size_t)(k0 + kl) * N + n0) * T + r);
#pragma unroll
      for (unsigned j = 0; j < 4u; ++j) {
        const unsigned rj = r + j, nl = (T > 1) ? __umulhi(rj, magicT) : rj, ts = rj - nl * T;
        tile[(nl * 33u + kl) * T + (T - 1u - ts)] = v[j];
      }
    }
  } else if (vec_src) {
    for (unsigned e = 4u * tid; e < 32u * RUN; e += 1024u) {
      const unsigned nl = __umulhi(e, magicR), r = e - nl * RUN;
      const f32x4 v = *reinterpret_cast<const f32x4*>(d.src + ((size_t)(n0 + nl) * K + k0) * T + r);
      float* o = tile + nl * 33u * T + r;
      o[0] = v[0]; o[1] = v[1]; o[2] = v[2]; o[3] = v[3];
    }
  } else if (d.kind == 2) {
    // [K][N][T] tensor: run kl = the 32 T floats of rows n0 .. n0 + 31 of column k0 + kl; tap t' of the source is tap T - 1 - t'
    for (unsigned e = tid; e < 32u * RUN; e += 256u) {
      const unsigned kl = __umulhi(e, magicR), r = e - kl * RUN;
      const unsigned nl = (T > 1) ? __umulhi(r, magicT) : r, ts = r - nl * T;
      float v = 0.f;
      if (k0 + kl < K && n0 + nl < N) v = d.src[((size_t)(k0 + kl) * N + n0) * T + r];
      tile[(nl * 33u + kl) * T + (T - 1u - ts)] = v;
    }
  } else {
    // [N][K][T] tensor: run nl = the 32 T floats of columns k0 .. k0 + 31 of row n0 + nl, already in the tile's order
    for (unsigned e = tid; e < 32u * RUN; e += 256u) {
      const unsigned nl = __umulhi(e, magicR), r = e - nl * RUN;
      const unsigned kl = (T > 1) ? __umulhi(r, magicT) : r;
      float v = 0.f;
      if (n0 + nl < N && k0 + kl < K) v = d.src[((size_t)(n0 + nl) * K + k0) * T + r];
      tile[nl * 33u * T + r] = v;
    }
  }
  __syncthreads();
  // ---- fp32 packing: row (n, t) = 32 consecutive columns = one 128-byte line; 8 rows per sweep ----
  if (vec_dst) {
    const unsigned kl = 4u * (tid & 7u);
    for (unsigned row = tid >> 3; row < RUN; row += 32u) {         // row = nl * T + t; 8 lanes x 16 bytes = the row's 128-byte line
      const unsigned nl = (T > 1) ? __umulhi(row, magicT) : row, t = row - nl * T;
      const float* ti = tile + (nl * 33u + kl) * T + t;
      *reinterpret_cast<f32x4*>(d.dst + ((size_t)(d.n_off + n0 + nl) * T + t) * d.Kp + d.k_off + k0 + kl) = f32x4{ti[0], ti[T], ti[2u * T], ti[3u * T]};
    }
  } else {
    const unsigned kl = tid & 31u;
    for (unsigned row = tid >> 5; row < RUN; row += 8u) {          // row = nl * T + t
      const unsigned nl = (T > 1) ? __umulhi(row, magicT) : row, t = row - nl * T;
      if (n0 + nl < N && k0 + kl < K)
        d.dst[((size_t)(d.n_off + n0 + nl) * T + t) * d.Kp + d.k_off + k0 + kl] = tile[(nl * 33u + kl) * T + t];
    }
  }
  // ---- bf16 pieces: unit (t, 16-column chunk c) = the 32 LDS rows of the n-tile, one wave each: lane (row, half) splits its 8 ----
  if (d.split) {
    const unsigned lane = tid & 63u, wv = tid >> 6, nrow = lane >> 1, half = lane & 1u;
    const unsigned ntile = ((unsigned)d.n_off + n0) >> 5, chunk0 = ((unsigned)d.k_off + k0) >> 4;
    for (unsigned u = wv; u < 2u * T; u += 4u) {
      const unsigned c = (u >= T) ? 1u : 0u, t = u - c * T;
      if (chunk0 + c >= (unsigned)d.nch) continue;                 // (wave-uniform: K % 32 == 16 leaves the last tile one chunk)
      float v[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = tile[(nrow * 33u + c * 16u + half * 8u + j) * T + t];
      unsigned q0[4], q1[4], q2[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) pidm_split3_pk(v[2 * j], v[2 * j + 1], q0[j], q1[j], q2[j]);
      unsigned short* o = d.split + (split_row_index(d.nch, (int)T, d.ntn, d.ntg, 0, (int)ntile, (int)(chunk0 + c), (int)t) * 32u + nrow) * (kSplitRow / 2) + half * 24u;
      *reinterpret_cast<u32x4*>(o) = u32x4{q0[0], q0[1], q0[2], q0[3]};
      *reinterpret_cast<u32x4*>(o + 8) = u32x4{q1[0], q1[1], q1[2], q1[3]};
      *reinterpret_cast<u32x4*>(o + 16) = u32x4{q2[0], q2[1], q2[2], q2[3]};
      // the row's 16 bytes of padding too (nobody reads them: kSplitRow is an LDS stride): with them the wave's 32 rows are 28 WHOLE
      // 128-byte lines instead of 28 lines with a hole each
      if (half) *reinterpret_cast<u32x4*>(o + 24) = u32x4{0u, 0u, 0u, 0u};
    }
  }
}

__global__ void __launch_bounds__(256) pack_multi_kernel(const PackDesc* __restrict__ table, int ndesc) {
  __shared__ float tile[32 * 33 * kPackTileT];
  int lo = 0, hi = ndesc - 1;
  const unsigned bid = blockIdx.x;
  while (lo < hi) {  // last descriptor with blk0 <= bid
    const int mid = (lo + hi + 1) >> 1;
    if (table[mid].blk0 <= bid) lo = mid; else hi = mid - 1;
  }
  const PackDesc d = table[lo];
  if (d.tiled) {
    pack_tile(d, bid - d.blk0, tile);
    return;
  }
  const unsigned K = (unsigned)d.K, T = (unsigned)d.T, N = (unsigned)d.N;
  const unsigned total = (unsigned)d.nz * N * T * K;
  const unsigned base = (bid - d.blk0) * 2048u;
  const bool phased = d.kind == 5 || d.kind == 6;
  const size_t ks = pack_src_kstride(d);
  if ((K & 1u) == 0) {
    // pairs (k, k + 1): k is even, and so is every packed column below when its offset is (checked per element: the concatenated
    // time-MLP linears have arbitrary offsets)
#pragma unroll 2
    for (int it = 0; it < 4; ++it) {
      const unsigned idx = base + it * 512u + 2u * threadIdx.x;
      if (idx >= total) break;
      const unsigned q1 = idx / K, k = idx - q1 * K;
      const unsigned q2 = q1 / T, t = q1 - q2 * T;
      const unsigned z = q2 / N, n = q2 - z * N;
      const size_t si = pack_src_index(d, z, n, t, k);
      const float v0 = d.src[si], v1 = d.src[si + ks];
      const unsigned zz = phased ? 0u : z;
      const unsigned col = (unsigned)d.k_off + (phased ? z * K : 0u) + k;
      float* o = d.dst + (((size_t)zz * d.Np + d.n_off + n) * T + t) * d.Kp + col;
      if ((col & 1u) == 0) {
        *reinterpret_cast<f32x2*>(o) = f32x2{v0, v1};
      } else {
        o[0] = v0;
        o[1] = v1;
      }
      if (d.split) {
        const unsigned nn = (unsigned)d.n_off + n;
        if ((col & 1u) == 0) {
          unsigned p0, p1, p2;
          pidm_split3_pk(v0, v1, p0, p1, p2);
          const size_t so = (split_row_index(d.nch, (int)T, d.ntn, d.ntg, (int)zz, (int)(nn >> 5), (int)(col >> 4), (int)t) * 32 + (nn & 31)) * (kSplitRow / 2) + ((col >> 3) & 1) * 24 + (col & 7);
          *reinterpret_cast<unsigned*>(d.split + so) = p0;
          *reinterpret_cast<unsigned*>(d.split + so + 8) = p1;
          *reinterpret_cast<unsigned*>(d.split + so + 16) = p2;
        } else {           // (never for a tensor with pieces: they exist for whole, un-offset matrices only)
          split_store(d.split, d.nch, d.T, d.ntn, d.ntg, (int)zz, (int)nn, (int)t, (int)col, v0);
          split_store(d.split, d.nch, d.T, d.ntn, d.ntg, (int)zz, (int)nn, (int)t, (int)col + 1, v1);
        }
      }
    }
    return;
  }
  for (int it = 0; it < 8; ++it) {
    const unsigned idx = base + it * 256u + threadIdx.x;
    if (idx >= total) break;
    const unsigned q1 = idx / K, k = idx - q1 * K;
    const unsigned q2 = q1 / T, t = q1 - q2 * T;
    const unsigned z = q2 / N, n = q2 - z * N;
    const float v = d.src[pack_src_index(d, z, n, t, k)];
    const unsigned zz = phased ? 0u : z;
    const unsigned col = (unsigned)d.k_off + (phased ? z * K : 0u) + k;
    d.dst[(((size_t)zz * d.Np + d.n_off + n) * T + t) * d.Kp + col] = v;
    if (d.split) split_store(d.split, d.nch, d.T, d.ntn, d.ntg, (int)zz, d.n_off + (int)n, (int)t, (int)col, v);
  }
}

// ---------------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------------
static bool is_pow2(int x) { return x > 0 && (x & (x - 1)) == 0; }

// kind: 0 normal conv (any KHxKW/stride/pad), 1 "transposed type" (4 parity classes of 2x2 taps, os=2)
int make_geom(ConvGeom* g, int kind, int B, int Hi, int Wi, int C0, int C1, int ld0, int ld1, int Cout, int KH, int KW,
              int stride, int pad, int out_nchw, int ldo, int ldr) {
  memset(g, 0, sizeof(*g));
  g->B = B; g->Hi = Hi; g->Wi = Wi; g->C0 = C0; g->C1 = C1; g->ld0 = ld0; g->ld1 = ld1 > 0 ? ld1 : 4;
  g->Cin = C0 + C1; g->Cout = Cout;
  g->nph = 1; g->in_step = 1; g->Kw = g->Cin;
  if (kind == 0 && KH == 4 && KW == 4 && stride == 2 && pad == 1 && (Hi % 2 == 0) && (Wi % 2 == 0) && (C0 % 8 == 0) &&
      (C1 % 8 == 0) && (ld0 % 4 == 0) && (C1 == 0 || ld1 % 4 == 0)) {
    // phased form of the 4x4/s2/p1 convolution: input row 2y + ky - 1 = 2*sy + py with (py = 0: sy = y + j, ky = 1 + 2j)
    // and (py = 1: sy = y + j - 1, ky = 2j), j in {0,1}; likewise in x.  K = 4 phases x Cin, taps 2x2, stride 1.
    g->KH = g->KW = 2; g->stride = 1; g->os = 1; g->nz = 1;
    g->nph = 4; g->in_step = 2; g->Kw = 4 * g->Cin;
    g->Hv = g->Ho = Hi / 2; g->Wv = g->Wo = Wi / 2;
    for (int ph = 0; ph < 4; ++ph) {
      const int py = ph >> 1, px = ph & 1;
      g->ph_oy[ph] = py; g->ph_ox[ph] = px;
      g->ph_pad_y[ph] = py; g->ph_pad_x[ph] = px;   // py = 0 -> pad 0, py = 1 -> pad 1
    }
  } else if (kind == 0) {
    g->KH = KH; g->KW = KW; g->stride = stride; g->os = 1; g->nz = 1;
    g->Hv = g->Ho = (Hi + 2 * pad - KH) / stride + 1;
    g->Wv = g->Wo = (Wi + 2 * pad - KW) / stride + 1;
    g->pad_y[0] = g->pad_x[0] = pad;
  } else {
    g->KH = g->KW = 2; g->stride = 1; g->os = 2; g->nz = 4;
    g->Hv = Hi; g->Wv = Wi; g->Ho = 2 * Hi; g->Wo = 2 * Wi;
    for (int z = 0; z < 4; ++z) {
      const int py = z >> 1, px = z & 1;
      g->pad_y[z] = py == 0 ? 1 : 0; g->pad_x[z] = px == 0 ? 1 : 0;
      g->ooy[z] = py; g->oox[z] = px;
    }
  }
  if (!is_pow2(g->Wv) || g->Wv > kBM || g->Hv < 1) return fail("conv: output width %d must be a power of two <= %d", g->Wv, kBM);
  g->TH = kBM / g->Wv < g->Hv ? kBM / g->Wv : g->Hv;
  if (g->Hv % g->TH) return fail("conv: output height %d not divisible by tile rows %d", g->Hv, g->TH);
  g->NI = kBM / (g->Wv * g->TH);
  for (g->wsh = 0; (1 << g->wsh) < g->Wv; ++g->wsh) {}
  for (g->tsh = 0; (1 << g->tsh) < g->TH; ++g->tsh) {}
  if ((1 << g->tsh) != g->TH) return fail("conv: tile rows %d must be a power of two", g->TH);
  g->IHt = (g->TH - 1) * g->stride + g->KH;
  g->IWt = (g->Wv - 1) * g->stride + g->KW;
  g->mIWt = g->IWt > 1 ? (unsigned)((0x100000000ULL + g->IWt - 1) / g->IWt) : 0;
  g->mIHt = g->IHt > 1 ? (unsigned)((0x100000000ULL + g->IHt - 1) / g->IHt) : 0;
  // tiny strided images: the halo blow-up (e.g. 36 input pixels per 2x2 output) must still fit in LDS
  while (g->NI > 1 && g->NI * g->IHt * g->IWt > 768) g->NI >>= 1;
  const int tpi = g->Hv / g->TH;
  g->tiles_m = (g->NI > 1) ? cdiv(B, g->NI) : B * tpi;
  if (out_nchw) { g->sob = (long)Cout * g->Ho * g->Wo; g->soc = (long)g->Ho * g->Wo; g->soy = g->Wo; g->sox = 1; }
  else { g->sob = (long)g->Ho * g->Wo * ldo; g->soy = (long)g->Wo * ldo; g->sox = ldo; g->soc = 1; }
  g->ldr = ldr;
  return 0;
}

int pick_kc(int Cin) {
  // PIDM_KC=8 forces the 8-channel chunk (experiments); default: 16 when the channel count allows it
  static int forced = -1;
  if (forced < 0) {
    const char* e = knob("PIDM_KC");
    forced = e ? atoi(e) : 0;
  }
  if (forced == 8) return 8;
  return (Cin % 16 == 0) ? 16 : 8;
}
// n-tiles per workgroup: 64 output channels per workgroup unless that leaves the chip under-filled
int pick_nt(int Cout, int tiles_m) {
  if (Cout <= 32) return 1;
  return (tiles_m * cdiv(Cout, 64) >= 384) ? 2 : 1;
}
// packed weights always pad Cout to a multiple of 64 so that either tile width can read them
int packed_np(int Cout) { return cdiv(Cout, 64) * 64; }

// K columns of one packed weight row (the leading dimension of the fp32 packing)
int packed_kp(const ConvGeom& g) {
  const int KC = pick_kc(g.Cin);
  return cdiv(g.Kw, KC) * KC;
}
static size_t packed_fp32_floats(const ConvGeom& g) {
  const int KC = pick_kc(g.Cin);
  const size_t Np = (size_t)packed_np(g.Cout), Kp = (size_t)cdiv(g.Kw, KC) * KC;
  return (size_t)g.nz * Np * g.KH * g.KW * Kp;
}
// fp32 packing, followed by the bf16 pieces (3 x 2 bytes per weight) where conv3x3_split_kernel can take the tensor
size_t packed_floats(const ConvGeom& g) {
  if (split_shape_ok(g)) return packed_fp32_floats(g) + (size_t)(g.Cout / 32) * (g.Cin / 16) * (kSplitSlab / 4) + 128;
  if (split_shape_ok2(g)) return packed_fp32_floats(g) + (size_t)g.nz * (g.Cout / 32) * (g.Kw / 16) * (4 * 32 * kSplitRow / 4) + 512;
  if (split_ntg(g)) return packed_fp32_floats(g) + (size_t)(g.Cout / 32) * (g.Cin / 16) * (32 * kSplitRow / 4) + 128;
  return packed_fp32_floats(g);
}
static unsigned short* split_part(const ConvGeom& g, float* w_packed) {
  return (split_shape_ok(g) || split_shape_ok2(g) || split_ntg(g)) ? reinterpret_cast<unsigned short*>(w_packed + packed_fp32_floats(g)) : nullptr;
}

// w_ref -> packed.  The packed matrix is sized by g (g.Cout rows, g.Cin columns); the source tensor covers rows
// [n_off, n_off+n_src) and columns [k_off, k_off+k_src) of it (n_src/k_src <= 0: the whole matrix).
// Padding must have been zero-filled by the caller.
int launch_pack(const ConvGeom& g, int kind, const float* w_ref, float* w_packed, int srcKH, int srcKW, int n_off, int k_off,
                int n_src, int k_src, hipStream_t st) {
  const int KC = pick_kc(g.Cin);
  const int Np = packed_np(g.Cout), Kp = cdiv(g.Kw, KC) * KC, T = g.KH * g.KW;
  const int N = n_src > 0 ? n_src : g.Cout, K = k_src > 0 ? k_src : g.Cin;
  if (g.nph > 1) kind = (kind == 4) ? 6 : 5;   // phased 4x4s2: forward of a strided conv (5) / dgrad of a transposed conv (6)
  const size_t total = (size_t)g.nz * N * T * K * g.nph;
  int blocks = (int)((total + 255) / 256);
  if (blocks > 4096) blocks = 4096;
  unsigned short* split = (n_off == 0 && k_off == 0 && n_src <= 0 && k_src <= 0) ? split_part(g, w_packed) : nullptr;
  hipLaunchKernelGGL(pack_kernel, dim3(blocks), dim3(256), 0, st, w_ref, w_packed, kind, g.nph > 1 ? 4 : g.nz, N, K, Np, Kp, srcKH,
                     srcKW, T, n_off, k_off, split, Kp / 16, g.Cout / 32, split_ntg(g) ? split_ntg(g) : 1);
  PIDM_CHECK_LAUNCH("pack_kernel");
  return 0;
}

// fill a descriptor for the multi-tensor pack (same arguments as launch_pack); returns its block count
unsigned make_pack_desc(const ConvGeom& g, int kind, const float* w_ref, float* w_packed, int srcKH, int srcKW, int n_off,
                        int k_off, int n_src, int k_src, PackDesc* d) {
  const int KC = pick_kc(g.Cin);
  d->src = w_ref; d->dst = w_packed; d->kind = kind; d->nz = g.nz;
  d->N = n_src > 0 ? n_src : g.Cout; d->K = k_src > 0 ? k_src : g.Cin;
  d->Np = packed_np(g.Cout); d->Kp = cdiv(g.Kw, KC) * KC; d->KH = srcKH; d->KW = srcKW; d->T = g.KH * g.KW;
  d->n_off = n_off; d->k_off = k_off;
  d->split = (n_off == 0 && k_off == 0 && n_src <= 0 && k_src <= 0) ? split_part(g, w_packed) : nullptr;
  d->nch = d->Kp / 16;
  d->ntn = g.Cout / 32;
  d->ntg = split_ntg(g) ? split_ntg(g) : 1;
  if (g.nph > 1) { d->kind = (kind == 4) ? 6 : 5; d->nz = 4; }   // nz doubles as the phase count for kinds 5/6
  const size_t total = (size_t)d->nz * d->N * d->T * d->K;
  if (total >= ((size_t)1 << 31)) {          // pack_multi_kernel indexes a tensor in 32 bits
    fail("weight re-pack: a tensor of %zu elements exceeds the 2^31 limit of the packing kernel", total);
    d->nblk = 0;
    return 0;
  }
  // tiled form (32 rows x 32 columns x all taps per block, transposed through LDS): the un-phased single-matrix kinds with 1 or 9
  // taps whose pieces (if any) start on tile borders; PIDM_PACK_TILED=0 / 1: element-wise everywhere / tiled where eligible
  // (read per table build, not cached: the host emulator's build defaults to the element-wise form - its fibers make the tile's
  // barrier expensive - and a test switches the tiled form on for one model)
#ifndef PIDM_PACK_TILED_DEFAULT
#define PIDM_PACK_TILED_DEFAULT 1
#endif
  const char* te = knob("PIDM_PACK_TILED");
  const bool tiled_on = te ? atoi(te) != 0 : PIDM_PACK_TILED_DEFAULT != 0;
  d->tiled = (tiled_on && d->nz == 1 && (d->kind == 0 || d->kind == 2 || d->kind == 4) && (d->T == 1 || d->T == kPackTileT) &&
              (!d->split || ((d->n_off & 31) == 0 && (d->k_off & 31) == 0))) ? 1 : 0;
  d->nblk = d->tiled ? (unsigned)(cdiv(d->N, 32) * cdiv(d->K, 32)) : (unsigned)((total + 2047) / 2048);
  return d->nblk;
}
int launch_pack_multi(const PackDesc* table_dev, int ndesc, unsigned nblocks, hipStream_t st) {
  hipLaunchKernelGGL(pack_multi_kernel, dim3(nblocks), dim3(256), 0, st, table_dev, ndesc);
  PIDM_CHECK_LAUNCH("pack_multi_kernel");
  return 0;
}

int launch_conv(const ConvGeom& g, const float* src0, const float* src1, const float* wp, const float* bias,
                const float* residual, float* out, int sigmoid_last, hipStream_t st) {
  const int KC = pick_kc(g.Cin), NT = pick_nt(g.Cout, g.tiles_m * g.nz);
  const bool nt4 = conv_nt4_ok(g, KC);
  if (prof_enabled()) {
    char lab[160];
    snprintf(lab, sizeof(lab), "conv B%d %dx%d Cin%d Cout%d k%dx%d nph%d nz%d%s%s%s", g.B, g.Hv, g.Wv, g.Cin, g.Cout, g.KH, g.KW, g.nph,
             g.nz, g.gn_part ? " +gnstats" : "", g.bn_part ? " +bnsums" : "", residual ? " +res" : "");
    prof_set_label(lab);
  }
  if (knob("PIDM_TRACE_CONV"))   // debugging aid: which tile configuration a launch takes
    fprintf(stderr, "[pidm] conv B=%d %dx%d Cin=%d Cout=%d k=%dx%d nph=%d%s%s%s -> KC=%d NT=%d\n", g.B, g.Hv, g.Wv, g.Cin, g.Cout, g.KH,
            g.KW, g.nph, g.gn_part ? " +gnstats" : "", g.bn_part ? (g.bn_res ? " +bnsums(res)" : " +bnsums") : "", residual ? " +res" : "", KC,
            nt4 ? 4 : NT);
  {
    // 1x1 / stride-1 convolutions as a split-form GEMM (conv1x1_split_kernel): contiguous channels-last input and output, pixel
    // count a multiple of 128, whole 32-channel chunks from either source
    const char* se = knob("PIDM_CONV_SPLIT");
    const int ntg = (se && !atoi(se)) ? 0 : split_ntg(g);
    const long npix = (long)g.B * g.Hv * g.Wv;
    // Measured (tools/bench_conv1x1.py, batch 64): with two stages per item (Cin = 64) the un-overlapped epilogue dominates - the
    // memory-bound 64x64 layers and 64 -> 768 are slower than on the fp32 kernel, 64 -> 256 at 16x16 is twice as fast
    const bool shape_pays = g.Cin >= 128 || (g.Cout <= 256 && npix <= 65536);
    if (ntg && shape_pays && g.Hv == g.Hi && g.Wv == g.Wi && g.Ho == g.Hv && g.Wo == g.Wv && g.pad_y[0] == 0 && g.pad_x[0] == 0 && (npix % 128) == 0 &&
        g.soc == 1 && g.soy == (long)g.Wo * g.sox && g.sob == (long)g.Ho * g.Wo * g.sox && (g.sox & 3) == 0 && (g.ld0 & 3) == 0 &&
        (g.C0 % 32) == 0 && (g.C1 == 0 || (g.ld1 == g.ld0 && src1)) && (size_t)npix * g.ld0 * 4 < ((size_t)1 << 32) && !sigmoid_last &&
        !g.gn_part && !g.bn_part && (!residual || ((g.ldr & 3) == 0 && (reinterpret_cast<size_t>(residual) & 15) == 0)) &&
        (reinterpret_cast<size_t>(src0) & 15) == 0 && (!src1 || (reinterpret_cast<size_t>(src1) & 15) == 0) &&
        (reinterpret_cast<size_t>(out) & 15) == 0) {
      const int tiles_m = (int)(npix / 128);
      const char* ce1 = knob("PIDM_STREAM_WGS");           // (the unit tests lower it: several items per workgroup)
      int n_cu1 = ce1 ? atoi(ce1) : 256;
      if (n_cu1 < 1) n_cu1 = 256;
      // tiles per workgroup: the packing's 4 unless that leaves CUs without an item
      const int ntr = (ntg == 4 && tiles_m * (g.Cout / 128) < n_cu1) ? 2 : ntg;
      const int n_items = tiles_m * (g.Cout / (32 * ntr));
      const int ipw = cdiv(n_items, n_cu1), wgs = cdiv(n_items, ipw);
      const size_t lds = (size_t)2 * (2 * 128 * kSplitRow + 2 * ntr * 32 * kSplitRow);
      const unsigned short* wsplit = reinterpret_cast<const unsigned short*>(wp + packed_fp32_floats(g));
      static bool attr_1 = false;
      if (!attr_1) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv1x1_split_kernel<4, 4, 4>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 256);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv1x1_split_kernel<2, 4, 4>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 256);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv1x1_split_kernel<2, 2, 4>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 256);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv1x1_split_kernel<4, 4, 8>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 256);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv1x1_split_kernel<2, 4, 8>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 256);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv1x1_split_kernel<2, 2, 8>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 256);
        attr_1 = true;
      }
      if (knob("PIDM_TRACE_CONV")) fprintf(stderr, "[pidm]   -> conv1x1_split_kernel<%d, %d>, %d items over %d workgroups, %zu B LDS\n", ntr, ntg, n_items, wgs, lds);
      const bool prof = prof_enabled();
      if (prof) prof_begin_launch(2, 2.0 * (double)npix * (double)g.Cout * g.Cin, st);
      const float* s1 = src1 ? src1 : src0;
      ConvGeom g1 = g;
      g1.xcd = split_xcd_order();
      if (split_npw(ntr == 2) == 8) {
        if (ntr == 4) hipLaunchKernelGGL(HIP_KERNEL_NAME(conv1x1_split_kernel<4, 4, 8>), dim3(wgs), dim3(768), lds, st, g1, src0, s1, wsplit, bias, residual, out, n_items, ipw, tiles_m);
        else if (ntg == 4) hipLaunchKernelGGL(HIP_KERNEL_NAME(conv1x1_split_kernel<2, 4, 8>), dim3(wgs), dim3(768), lds, st, g1, src0, s1, wsplit, bias, residual, out, n_items, ipw, tiles_m);
        else hipLaunchKernelGGL(HIP_KERNEL_NAME(conv1x1_split_kernel<2, 2, 8>), dim3(wgs), dim3(768), lds, st, g1, src0, s1, wsplit, bias, residual, out, n_items, ipw, tiles_m);
      } else {
        if (ntr == 4) hipLaunchKernelGGL(HIP_KERNEL_NAME(conv1x1_split_kernel<4, 4, 4>), dim3(wgs), dim3(512), lds, st, g1, src0, s1, wsplit, bias, residual, out, n_items, ipw, tiles_m);
        else if (ntg == 4) hipLaunchKernelGGL(HIP_KERNEL_NAME(conv1x1_split_kernel<2, 4, 4>), dim3(wgs), dim3(512), lds, st, g1, src0, s1, wsplit, bias, residual, out, n_items, ipw, tiles_m);
        else hipLaunchKernelGGL(HIP_KERNEL_NAME(conv1x1_split_kernel<2, 2, 4>), dim3(wgs), dim3(512), lds, st, g1, src0, s1, wsplit, bias, residual, out, n_items, ipw, tiles_m);
      }
      if (prof) prof_end_launch(st);
      PIDM_CHECK_LAUNCH("conv1x1_split_kernel");
      return 0;
    }
  }
  {
    // the 7x7 init convolution with (kx, channel) flattened into the contraction index, split form (conv7x7_split_kernel)
    const char* se = knob("PIDM_CONV_SPLIT");
    const bool on = !(se && !atoi(se));
    if (on && g.KH == 7 && g.KW == 7 && g.stride == 1 && g.nz == 1 && g.nph == 1 && g.os == 1 && g.pad_y[0] == 3 && g.pad_x[0] == 3 &&
        g.C1 == 0 && (g.Cin == 2 || g.Cin == 4) && g.ld0 == g.Cin && (g.Cout % 32) == 0 && g.soc == 1 && (g.sox & 3) == 0 &&
        g.Wv == g.Wi && g.Hv == g.Hi && g.Wv >= 8 && g.Wv <= 256 && (256 % g.Wv) == 0 && (g.Hv % (256 / g.Wv)) == 0 &&
        (!residual || ((g.ldr & 3) == 0 && (reinterpret_cast<size_t>(residual) & 15) == 0)) && !sigmoid_last && !g.gn_part && !g.bn_part && (reinterpret_cast<size_t>(out) & 15) == 0) {
      const int KCp = pick_kc(g.Cin), Kp = cdiv(g.Kw, KCp) * KCp;
      const int TH = 256 / g.Wv, n_tiles = g.B * (g.Hv / TH);
      const int KSt = (7 * g.Cin + 15) / 16, RL = ((g.Wv + 6) * g.Cin + (16 * KSt - 7 * g.Cin) + 1) & ~1;
      const size_t lds = ((size_t)(TH + 6) * RL + (size_t)32 * ((49 * Kp) | 1)) * sizeof(float);
      const char* ce7 = knob("PIDM_STREAM_WGS");              // (the unit tests lower it: several tiles per workgroup)
      int n_wg7 = ce7 ? atoi(ce7) : 256;                      // 170 registers x 512 threads: one workgroup per CU
      if (n_wg7 < 1) n_wg7 = 256;
      const int tpw = cdiv(n_tiles, n_wg7);                     // persistent workgroups: the weight prologue is paid once per workgroup
      const dim3 grid(cdiv(n_tiles, tpw), g.Cout / 32, 1);
      const bool prof = prof_enabled();
      if (knob("PIDM_TRACE_CONV")) fprintf(stderr, "[pidm]   -> conv7x7_split_kernel<%d>, %d tiles, %d per workgroup, %zu B LDS\n", g.Cin, n_tiles, tpw, lds);
      if (prof) prof_begin_launch(2, 2.0 * g.B * g.Hv * g.Wv * (double)g.Cout * g.Cin * 49, st);
      if (g.Cin == 2) hipLaunchKernelGGL(HIP_KERNEL_NAME(conv7x7_split_kernel<2>), grid, dim3(512), lds, st, g, src0, wp, Kp, bias, residual, out, n_tiles, tpw);
      else hipLaunchKernelGGL(HIP_KERNEL_NAME(conv7x7_split_kernel<4>), grid, dim3(512), lds, st, g, src0, wp, Kp, bias, residual, out, n_tiles, tpw);
      if (prof) prof_end_launch(st);
      PIDM_CHECK_LAUNCH("conv7x7_split_kernel");
      return 0;
    }
  }
  {
    // the 4x4 / stride-2 family (2x2 taps as 4 K-phases or 4 output parities) in the split form
    const char* se = knob("PIDM_CONV_SPLIT");
    const bool on = !(se && !atoi(se));
    const int mode = (g.nph == 4) ? 1 : 2;
    if (on && split_shape_ok2(g) && g.soc == 1 && (g.C0 % 16 == 0) && ((g.ld0 | g.ld1) & 3) == 0 && (g.C1 == 0 || g.ld1 == g.ld0) &&
        g.Wv >= 8 && (mode == 1 ? (g.in_step == 2 && 2 * g.Wv == g.Wi && 2 * g.Hv == g.Hi) : (g.os == 2 && g.Wv == g.Wi && g.Hv == g.Hi)) &&
        !sigmoid_last && !g.gn_part && !g.bn_part && (g.sox & 3) == 0 && (reinterpret_cast<size_t>(out) & 15) == 0 &&
        (reinterpret_cast<size_t>(src0) & 15) == 0 && (!src1 || (reinterpret_cast<size_t>(src1) & 15) == 0) &&
        (!residual || ((g.ldr & 3) == 0 && (reinterpret_cast<size_t>(residual) & 15) == 0)) &&
        (double)g.B * g.Hi * g.Wi * g.ld0 * 4.0 < 4.0e9) {
      const char* ce = knob("PIDM_STREAM_WGS");
      int n_cu = ce ? atoi(ce) : 256;
      if (n_cu < 1) n_cu = 256;
      const char* fe = knob("PIDM_SPLIT_NW");
      const int force = fe ? atoi(fe) : 0;
      const int mult = (mode == 2 ? 4 : 1) * (g.Cout / 32);
      for (int nw = 8; nw >= 4; nw >>= 1) {
        if (force && force != nw) continue;
        ConvGeom gs = g;
        // tile of 32 nw pixels with the 3x3 halo layout (one halo row / column on either side)
        const int bm = 32 * nw;
        if (gs.Wv > bm) continue;
        const int TH = bm / gs.Wv < gs.Hv ? bm / gs.Wv : gs.Hv;
        int tsh = 0;
        while ((1 << tsh) < TH) ++tsh;
        if (gs.Hv % TH || (1 << tsh) != TH) continue;
        const int NI = bm / (gs.Wv * TH);
        if (NI * gs.Wv * TH != bm) continue;
        gs.TH = TH; gs.tsh = tsh; gs.NI = NI; gs.IHt = TH + 2; gs.IWt = gs.Wv + 2;
        gs.mIHt = (unsigned)((0x100000000ULL + gs.IHt - 1) / gs.IHt);
        gs.mIWt = (unsigned)((0x100000000ULL + gs.IWt - 1) / gs.IWt);
        gs.tiles_m = (NI > 1) ? cdiv(gs.B, NI) : gs.B * (gs.Hv / TH);
        const int npixA = gs.NI * gs.IHt * gs.IWt, SEG = gs.NI * gs.IHt * gs.Wv;
        gs.rpad = split_row_pad(gs.Wv);
        gs.xcd = split_xcd_order();
        const size_t lds = (size_t)2 * ((npixA + 4 * 32) * kSplitRow + gs.NI * gs.IHt * gs.rpad + 2048);
        if (!(SEG % 32 == 0 && 2 * SEG >= 64 * nw && 2 * SEG <= 128 * nw && lds <= 160 * 1024 - 256)) continue;
        const int n_items = gs.tiles_m * mult;
        if (nw == 8 && !force && n_items < n_cu) continue;        // more, smaller items fill the chip better (the 4-wave tile follows)
        static bool attr_q = false;
        if (!attr_q) {
          (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3_split_kernel<8, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 256);
          (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3_split_kernel<4, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 256);
          (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3_split_kernel<8, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 256);
          (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3_split_kernel<4, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 256);
          attr_q = true;
        }
        const int ipw = cdiv(n_items, n_cu), wgs = cdiv(n_items, ipw);
        const unsigned short* wsplit = reinterpret_cast<const unsigned short*>(wp + packed_fp32_floats(g));
        const float* s1 = src1 ? src1 : src0;
        if (knob("PIDM_TRACE_CONV")) fprintf(stderr, "[pidm]   -> conv3x3_split_kernel<%d, %d>, %d items over %d workgroups, %zu B LDS\n", nw, mode, n_items, wgs, lds);
        const bool prof = prof_enabled();
        if (prof) prof_begin_launch(2, 2.0 * g.B * g.Hv * g.Wv * (double)g.Cout * g.Kw * 4 * g.nz, st);
        const dim3 bd(64 * nw);
        if (split_ws_on(nw)) {
          PIDM_PROF_NAME(nw == 8 ? "conv3x3_split_ws_kernel<8, 2x2>" : "conv3x3_split_ws_kernel<4, 2x2>");
          if (nw == 8 && mode == 1) launch_split_ws<8, 1>(wgs, lds, st, gs, src0, s1, wsplit, bias, residual, out, n_items, ipw, 0);
          else if (nw == 4 && mode == 1) launch_split_ws<4, 1>(wgs, lds, st, gs, src0, s1, wsplit, bias, residual, out, n_items, ipw, 0);
          else if (nw == 8) launch_split_ws<8, 2>(wgs, lds, st, gs, src0, s1, wsplit, bias, residual, out, n_items, ipw, 0);
          else launch_split_ws<4, 2>(wgs, lds, st, gs, src0, s1, wsplit, bias, residual, out, n_items, ipw, 0);
        } else {
        PIDM_PROF_NAME(nw == 8 ? "conv3x3_split_kernel<8, 2x2>" : "conv3x3_split_kernel<4, 2x2>");
        if (nw == 8 && mode == 1) hipLaunchKernelGGL(HIP_KERNEL_NAME(conv3x3_split_kernel<8, 1>), dim3(wgs), bd, lds, st, gs, src0, s1, wsplit, bias, residual, out, n_items, ipw, 0);
        else if (nw == 4 && mode == 1) hipLaunchKernelGGL(HIP_KERNEL_NAME(conv3x3_split_kernel<4, 1>), dim3(wgs), bd, lds, st, gs, src0, s1, wsplit, bias, residual, out, n_items, ipw, 0);
        else if (nw == 8) hipLaunchKernelGGL(HIP_KERNEL_NAME(conv3x3_split_kernel<8, 2>), dim3(wgs), bd, lds, st, gs, src0, s1, wsplit, bias, residual, out, n_items, ipw, 0);
        else hipLaunchKernelGGL(HIP_KERNEL_NAME(conv3x3_split_kernel<4, 2>), dim3(wgs), bd, lds, st, gs, src0, s1, wsplit, bias, residual, out, n_items, ipw, 0);
        }
        if (prof) prof_end_launch(st);
        PIDM_CHECK_LAUNCH("conv3x3_split_kernel(2x2)");
        return 0;
      }
    }
  }
  {
    // bf16 matrix pipe, fp32-faithful split operands (PIDM_CONV_SPLIT=0: off -> the fp32-MFMA kernels below; read per launch)
    const char* se = knob("PIDM_CONV_SPLIT");
    const bool on = !(se && !atoi(se));
    ConvGeom gs = g;
    if (on && split_shape_ok(g) && g.soc == 1 && (g.C0 % 16 == 0) && ((g.ld0 | g.ld1) & 3) == 0 && (g.C1 == 0 || g.ld1 == g.ld0) &&
        g.Wv >= 8 && g.Wv == g.Wi && g.pad_y[0] == 1 && g.pad_x[0] == 1 && !sigmoid_last && (g.sox & 3) == 0 &&
        (reinterpret_cast<size_t>(out) & 15) == 0 && (reinterpret_cast<size_t>(src0) & 15) == 0 &&
        (!src1 || (reinterpret_cast<size_t>(src1) & 15) == 0) &&
        (!residual || ((g.ldr & 3) == 0 && (reinterpret_cast<size_t>(residual) & 15) == 0)) &&
        (double)g.B * g.Hi * g.Wi * g.ld0 * 4.0 < 4.0e9) {
      // rows of 32 / 64 pixels with 32 / 64 input channels: the row-streaming kernel (k_conv_rs.hip)
      if (!knob("PIDM_STREAM_TRACE") && !knob("PIDM_SPLIT_NW")) {
        const int rc = launch_conv_rs(g, src0, src1, reinterpret_cast<const unsigned short*>(wp + packed_fp32_floats(g)), bias, residual, out, st);
        if (rc <= 0) return rc;
      }
      // 8 waves on a 256-pixel tile, or - when that leaves CUs without a work item - 4 waves on 128 pixels (PIDM_SPLIT_NW forces one)
      const char* fe = knob("PIDM_SPLIT_NW");
      const int force = fe ? atoi(fe) : 0;
      ConvGeom g8 = g;
      const char* ce = knob("PIDM_STREAM_WGS");        // persistent workgroups (default: one per CU of an MI355X); read per launch
      int n_cu = ce ? atoi(ce) : 256;
      if (n_cu < 1) n_cu = 256;
      const bool small = !retile_bm(&g8, 256) || g8.tiles_m * (g.Cout / 32) < n_cu;
      for (int pass = 0; pass < 2; ++pass) {
        const int nw = (small != (pass == 1)) ? 4 : 8;
        if (force && force != nw) continue;
        gs = g;
        if (!retile_bm(&gs, 32 * nw)) continue;
        const int npixA = gs.NI * gs.IHt * gs.IWt, SEG = gs.NI * gs.IHt * gs.Wv;
        gs.rpad = split_row_pad(gs.Wv);
        gs.xcd = split_xcd_order();
        const size_t lds = (size_t)2 * ((npixA + 9 * 32) * kSplitRow + gs.NI * gs.IHt * gs.rpad + 512);
        if (!(SEG % 32 == 0 && 2 * SEG >= 64 * nw && 2 * SEG <= 128 * nw && lds <= 160 * 1024 - 256)) continue;
        const int n_items = gs.tiles_m * (g.Cout / 32);
        const bool prof = prof_enabled();
        static bool attr_p = false;
        if (!attr_p) {
          (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3_split_kernel<8, 0>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 256);
          (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3_split_kernel<4, 0>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 256);
          attr_p = true;
        }
        gs.w_off[0] = 0;
        const int ipw = cdiv(n_items, n_cu), wgs = cdiv(n_items, ipw);
        const unsigned short* wsplit = reinterpret_cast<const unsigned short*>(wp + packed_fp32_floats(g));
        const int trace = knob("PIDM_STREAM_TRACE") ? 1 : 0;
        if (knob("PIDM_TRACE_CONV")) fprintf(stderr, "[pidm]   -> conv3x3_split_kernel<%d>, %d items over %d workgroups, %zu B LDS\n", nw, n_items, wgs, lds);
        if (prof) prof_begin_launch(2, 2.0 * g.B * g.Hv * g.Wv * (double)g.Cout * g.Kw * 9, st);
        if (split_ws_on(nw)) {
          PIDM_PROF_NAME(nw == 8 ? "conv3x3_split_ws_kernel<8, 0>" : "conv3x3_split_ws_kernel<4, 0>");
          if (nw == 8) launch_split_ws<8, 0>(wgs, lds, st, gs, src0, src1 ? src1 : src0, wsplit, bias, residual, out, n_items, ipw, trace);
          else launch_split_ws<4, 0>(wgs, lds, st, gs, src0, src1 ? src1 : src0, wsplit, bias, residual, out, n_items, ipw, trace);
        } else {
        PIDM_PROF_NAME(nw == 8 ? "conv3x3_split_kernel<8, 0>" : "conv3x3_split_kernel<4, 0>");
        if (nw == 8)
          hipLaunchKernelGGL(HIP_KERNEL_NAME(conv3x3_split_kernel<8, 0>), dim3(wgs), dim3(512), lds, st, gs, src0, src1 ? src1 : src0, wsplit, bias, residual, out,
                             n_items, ipw, trace);
        else
          hipLaunchKernelGGL(HIP_KERNEL_NAME(conv3x3_split_kernel<4, 0>), dim3(wgs), dim3(256), lds, st, gs, src0, src1 ? src1 : src0, wsplit, bias, residual, out,
                             n_items, ipw, trace);
        }
        if (prof) prof_end_launch(st);
        PIDM_CHECK_LAUNCH("conv3x3_split_kernel");
        return 0;
      }
    }
  }
  if (g.bn_part && g.bn_res) {
    // GroupNorm-backward sums of (result + residual): only the split-form 3x3 epilogues above add the residual before summing.
    // Any other kernel runs the convolution without the sums and says so (1 = sums NOT produced: the caller runs its own pass).
    ConvGeom g2 = g;
    g2.bn_part = nullptr;
    const int rc = launch_conv(g2, src0, src1, wp, bias, residual, out, sigmoid_last, st);
    return rc < 0 ? rc : 1;
  }
  return launch_conv_fp32(g, src0, src1, wp, bias, residual, out, sigmoid_last, st, KC, NT, nt4);
}

// geometry of the forward op described by a public pidm_conv_desc
int geom_fwd(const pidm_conv_desc* d, ConvGeom* g) {
  if (d->transposed) {
    if (d->KH != 4 || d->KW != 4 || d->stride != 2 || d->pad != 1) return fail("transposed conv: only 4x4 s2 p1");
    return make_geom(g, 1, d->B, d->Hi, d->Wi, d->C0, d->C1, d->ld0, d->ld1, d->Cout, 4, 4, 2, 1, d->out_nchw, d->ldo, d->ldo);
  }
  return make_geom(g, 0, d->B, d->Hi, d->Wi, d->C0, d->C1, d->ld0, d->ld1, d->Cout, d->KH, d->KW, d->stride, d->pad,
                   d->out_nchw, d->ldo, d->ldo);
}

// geometry of the adjoint (dgrad) problem: input = dy [B,Ho,Wo,Cout] (stride ld_dy), output = dx [B,Hi,Wi,Cin] (stride ld_dx)
int geom_dgrad(const pidm_conv_desc* d, int ld_dy, int ld_dx, ConvGeom* g, int* pack_kind) {
  const int Cin = d->C0 + d->C1;
  if (d->transposed) {
    *pack_kind = 4;
    return make_geom(g, 0, d->B, 2 * d->Hi, 2 * d->Wi, d->Cout, 0, ld_dy, 0, Cin, 4, 4, 2, 1, 0, ld_dx, ld_dx);
  }
  const int Ho = (d->Hi + 2 * d->pad - d->KH) / d->stride + 1, Wo = (d->Wi + 2 * d->pad - d->KW) / d->stride + 1;
  if (d->stride == 1) {
    *pack_kind = 2;
    return make_geom(g, 0, d->B, Ho, Wo, d->Cout, 0, ld_dy, 0, Cin, d->KH, d->KW, 1, d->KH - 1 - d->pad, 0, ld_dx, ld_dx);
  }
  if (d->stride == 2 && d->KH == 4 && d->KW == 4 && d->pad == 1) {
    *pack_kind = 3;
    return make_geom(g, 1, d->B, Ho, Wo, d->Cout, 0, ld_dy, 0, Cin, 4, 4, 2, 1, 0, ld_dx, ld_dx);
  }
  return fail("dgrad: unsupported conv geometry");
}

}  // namespace pidm

using namespace pidm;

extern "C" size_t pidm_conv_packed_weight_floats(const pidm_conv_desc* d) {
  ConvGeom g;
  if (geom_fwd(d, &g)) return 0;
  return packed_floats(g);
}

extern "C" int pidm_conv_pack_weights(const pidm_conv_desc* d, const float* w_ref, float* w_packed, int mode, void* stream) {
  ConvGeom g;
  int kind = 0;
  if (mode == 0) {
    if (geom_fwd(d, &g)) return -1;
    kind = d->transposed ? 1 : 0;
  } else if (geom_dgrad(d, d->Cout, d->C0 + d->C1, &g, &kind)) {
    return -1;
  }
  if (hipMemsetAsync(w_packed, 0, packed_floats(g) * sizeof(float), as_stream(stream)) != hipSuccess)
    return fail("pack: memset failed");
  return launch_pack(g, kind, w_ref, w_packed, d->KH, d->KW, 0, 0, 0, 0, as_stream(stream));
}

extern "C" size_t pidm_conv_dgrad_packed_weight_floats(const pidm_conv_desc* d) {
  ConvGeom g;
  int kind;
  if (geom_dgrad(d, d->Cout, d->C0 + d->C1, &g, &kind)) return 0;
  return packed_floats(g);
}

extern "C" int pidm_conv_forward(const pidm_conv_desc* d, const float* src0, const float* src1, const float* w_packed,
                                 const float* bias, const float* residual, float* out, void* stream) {
  ConvGeom g;
  if (geom_fwd(d, &g)) return -1;
  return launch_conv(g, src0, src1, w_packed, bias, residual, out, 0, as_stream(stream));
}

// forward convolution that also leaves GroupNorm partial statistics of its output (ConvGeom::gn_part): returns the number of
// chunks per image written to `partial` [B][chunks][groups][2] doubles (32-pixel chunks behind the tile kernels, whole strips
// behind the row-streaming kernel: ConvGeom::part_chunks_out), 0 if this shape / kernel has no statistics
// epilogue (the convolution itself is done either way), < 0 on error
extern "C" int pidm_conv_forward_gn_partials(const pidm_conv_desc* d, const float* src0, const float* src1, const float* w_packed,
                                             const float* bias, float* out, int groups, double* partial, void* stream) {
  ConvGeom g;
  if (geom_fwd(d, &g)) return -1;
  const int HW = g.Ho * g.Wo, cpg = (groups > 0 && g.Cout % groups == 0) ? g.Cout / groups : 0;
  int chunks = HW / 32;
  const bool ok = partial && g.Cout % 32 == 0 && cpg >= 4 && cpg <= 32 && (cpg & (cpg - 1)) == 0 && HW % 32 == 0 && g.nz == 1 &&
                  g.nph == 1 && g.os == 1 && g.soc == 1 && g.KH == 3;
  if (ok) {
    g.gn_part = partial;
    g.gn_cpg = cpg;
    g.gn_G = groups;
    g.gn_nchunk = HW / 32;
    g.part_chunks_out = &chunks;
  }
  const int rc = launch_conv(g, src0, src1, w_packed, bias, nullptr, out, 0, as_stream(stream));
  if (rc < 0) return rc;
  return (ok && rc == 0) ? chunks : 0;
}

extern "C" int pidm_conv_dgrad(const pidm_conv_desc* d, const float* dy, int ld_dy, const float* w_packed_dgrad,
                               const float* residual, float* dx, int ld_dx, void* stream) {
  ConvGeom g;
  int kind;
  if (geom_dgrad(d, ld_dy, ld_dx, &g, &kind)) return -1;
  return launch_conv(g, dy, nullptr, w_packed_dgrad, nullptr, residual, dx, 0, as_stream(stream));
}

// measurement aid: the cycle stamps conv3x3_stream_kernel left (PIDM_STREAM_TRACE=1): 4 per stage, up to 64 stages
extern "C" int pidm_debug_stream_trace(unsigned long long* out256) {
  return hipMemcpyFromSymbol(out256, HIP_SYMBOL(pidm::g_stream_trace), sizeof(unsigned long long) * 256) == hipSuccess ? 0 : -1;
}
