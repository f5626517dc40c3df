// 3x3 / stride-1 weight gradient on the bf16 matrix pipe, "row-streaming" form (round 4): no LDS and no barrier in the main loop.
//
// Replaces the backward of every 3x3 convolution's weight (`Block.proj`, /root/reference/src/unet_model.py:227 through
// loss.backward(), main.py:164):  dW[m][ky][kx][n] = sum_p dY[p][m] X[p + (ky-1, kx-1)][n].
//
// The contraction index of a weight gradient is the PIXEL, so the 32x32x16 bf16 MFMA wants, per lane, 8 consecutive k values =
// 8 pixels of ONE channel - while the tensors are channels-last.  conv_wgrad_split_kernel (k_conv.hip) transposes through
// registers + LDS and then takes turns between staging and k-steps (matrix pipe 0.10-0.29 busy, profiles/r03_pmc_split_kernels.txt).
// Here the transpose is the LOAD: lane l of a wave reads channel l & 31 of pixel x0 + j for j = 0..7 with eight dword loads -
// each is two whole 128-byte lines (lanes 0-31: one pixel of strip A, lanes 32-63: one pixel of strip B) - and then owns exactly
// the MFMA operand: 8 consecutive pixels of a row for its channel.  The values are split into their three bf16 pieces in
// registers (pidm_common.h) and go straight into the MFMA.
//
//   work item  = a pair of "strips": 8 pixels wide, R rows high (R a power of two, chosen by the launcher so that every wave of
//                the chip has an item), one strip per wave half; the wave walks DOWN the rows: a k-step = one row of both
//                strips = 16 pixels, and the 3x3 window needs the X rows y-1, y, y+1 - of which y-1 and y are the previous
//                k-steps' rows.  They stay in registers as pieces (rolling window of three rows), so each k-step loads and
//                splits ONE new X row and ONE dY row: ~110 vector instructions next to 54 MFMAs (9 taps x 6 split terms).
//   kx = 0 / 2 = the centre row's registers moved by one bf16 (v_alignbit with the piece of the pixel before / after, which is
//                loaded and split with the row: 10 loads per X row).
//   padding    = raw buffer loads: a row above / below the image, the pixel left of x = 0 / right of x = W-1 and the missing
//                second strip of an odd tail are read with an out-of-range offset, which returns 0 without a memory access -
//                no branches, no selects on the data.
//   wave       = all 9 taps of a 32 (dY channels) x 32 (X channels) block: 144 accumulator registers, 108 registers of row
//                pieces; one wave per SIMD (4 waves per workgroup and CU, 512-register budget), loads two k-steps ahead.
//   workgroup  = (split, block): its 4 waves take consecutive items of the split's range; at the end the four accumulator sets
//                are summed through LDS (fixed order) and leave as the split's partial slab, exactly the layout
//                conv_wgrad_split_kernel writes - the deferred fixed-order reduction (reduce_multi_kernel) is unchanged, results
//                stay bit-identical run to run.
// Measured and not kept (profiles/r04_b_wgrad_rs_variants.txt, r04_c_*): a fourth row buffer so that every MFMA of an iteration
// reads pieces finished an iteration earlier, the iteration laid out by hand as 18 fenced groups of 3 MFMAs with a slice of the
// vector work each (the ISA then shows 1-7 vector instructions between consecutive MFMAs instead of one batch of ~140), the same
// with sched_group_barrier, and -fno-slp-vectorize (no v_pk_add_f32 beside MFMAs): all within +-2 % of this form at batch 64
// and 256 - like the forward kernels the launch runs at the rate the chip sustains for a bf16 MFMA stream on real data
// (~1.05 PFLOP/s of bf16 terms = 175 TFLOP/s fp32-equivalent at batch 256), not at an issue limit of its own.
// Arithmetic per product: identical to conv_wgrad_split_kernel (same pieces, same six terms smallest first); only the order in
// which pixels meet an accumulator differs.
#include <stdio.h>
#include <stdlib.h>

#include "pidm_launch.h"

namespace pidm {

namespace {

struct RsRow {        // the bf16 pieces of one X row segment: centre, moved right by one pixel (kx = 0), moved left (kx = 2)
  u32x4 c[3], l[3], r[3];
};

__device__ __forceinline__ void rs_split_row(const float (&v)[10], RsRow& o) {
  // v[0] = pixel before, v[1..8] = the 8 pixels, v[9] = pixel after
  unsigned e[3];
  pidm_split3_pk(v[9], v[0], e[0], e[1], e[2]);      // low half: after, high half: before
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    unsigned p0, p1, p2;
    pidm_split3_pk(v[1 + 2 * k], v[2 + 2 * k], p0, p1, p2);
    o.c[0][k] = p0; o.c[1][k] = p1; o.c[2][k] = p2;
  }
#pragma unroll
  for (int pc = 0; pc < 3; ++pc) {
    const unsigned m01 = __builtin_amdgcn_alignbit(o.c[pc][1], o.c[pc][0], 16), m12 = __builtin_amdgcn_alignbit(o.c[pc][2], o.c[pc][1], 16),
                   m23 = __builtin_amdgcn_alignbit(o.c[pc][3], o.c[pc][2], 16);
    o.l[pc] = u32x4{__builtin_amdgcn_alignbit(o.c[pc][0], e[pc], 16), m01, m12, m23};
    o.r[pc] = u32x4{m01, m12, m23, __builtin_amdgcn_alignbit(e[pc], o.c[pc][3], 16)};
  }
}

__device__ __forceinline__ float rs_split_dy(const float (&v)[8], u32x4 (&ya)[3]) {
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    unsigned p0, p1, p2;
    pidm_split3_pk(v[2 * k], v[2 * k + 1], p0, p1, p2);
    ya[0][k] = p0; ya[1][k] = p1; ya[2][k] = p2;
  }
  return ((v[0] + v[1]) + (v[2] + v[3])) + ((v[4] + v[5]) + (v[6] + v[7]));
}

// six bf16 terms of one fp32 product block, smallest first (pidm_common.h)
#define PIDM_RS_SIX(acc_, ya_, xb_)                                    \
  acc_ = pidm_mfma_bf16_32x32x16(ya_[2], xb_[0], acc_);               \
  acc_ = pidm_mfma_bf16_32x32x16(ya_[0], xb_[2], acc_);               \
  acc_ = pidm_mfma_bf16_32x32x16(ya_[1], xb_[1], acc_);               \
  acc_ = pidm_mfma_bf16_32x32x16(ya_[1], xb_[0], acc_);               \
  acc_ = pidm_mfma_bf16_32x32x16(ya_[0], xb_[1], acc_);               \
  acc_ = pidm_mfma_bf16_32x32x16(ya_[0], xb_[0], acc_);

constexpr unsigned kRsInv = 0x80000000u;   // a byte offset no tensor reaches (the launcher checks): reads as 0

}  // namespace

// The body of the kernel: problem `wg` (by value in scalar registers: kernel arguments of the single-problem kernel, a row of the
// device table of the grouped one), workgroup (split, by) of its splits x blocks grid.
__device__ __forceinline__ void conv_wgrad_rs_body(const WgradItem& wg, const int split, const int by, float* __restrict__ red) {
  const float* __restrict__ src0 = wg.src0;
  const float* __restrict__ src1 = wg.src1;
  const float* __restrict__ dy = wg.dy;
  float* __restrict__ partial = wg.partial;
  float* __restrict__ bias_partial = wg.bias_partial;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int half = lane >> 5, l31 = lane & 31;
  const int ntn = wg.NP / 32;
  const int tn = by % ntn, tm = by / ntn;
  const int m0 = tm * 32, n0 = tn * 32;
  const int H = wg.Hi, W = wg.Wi, R = wg.rs_R;
  const int wsh = wg.wsh;                                  // log2 W
  const unsigned ldxb = (unsigned)wg.ld0 * 4u, ldyb = (unsigned)wg.ld_dy * 4u;  // bytes between pixels
  const unsigned rowxb = ldxb << wsh, rowyb = ldyb << wsh;                      // bytes between rows
  const unsigned tot_x = (unsigned)wg.B * (unsigned)H * rowxb, tot_y = (unsigned)wg.B * (unsigned)H * rowyb;
  // the n-tile lies in one source of a concatenation (C0 % 32 == 0, equal channel strides: checked by the launcher)
  const float* xsrc = (n0 < wg.C0) ? src0 + n0 : src1 + (n0 - wg.C0);
  const pidm_rsrc rx = pidm_make_rsrc(xsrc, tot_x), ry = pidm_make_rsrc(dy + m0, tot_y);
  const bool do_bias = (bias_partial != nullptr) && (tn == 0);

  f32x16 acc[9];
#pragma unroll
  for (int i = 0; i < 9; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  float bacc = 0.f;

  const int wv = split * 4 + wave;                         // this wave among the 4 * nsplit waves of the block
  const int npairs = (wg.rs_S + 1) >> 1;
  int pair_lo = wv * wg.rs_ppw, pair_hi = pair_lo + wg.rs_ppw;
  if (pair_hi > npairs) pair_hi = npairs;

  for (int pair = pair_lo; pair < pair_hi; ++pair) {
    // ---- this lane's strip: s = ((b * (H / R) + chunk) * (W / 8) + xs) ----
    const int s = 2 * pair + half;
    const bool s_ok = s < wg.rs_S;
    const int xs = s & ((1 << wg.rs_xsh) - 1), t1 = s >> wg.rs_xsh;
    const int chunk = t1 & ((1 << wg.rs_csh) - 1), b = t1 >> wg.rs_csh;
    const int x0 = xs << 3, r0 = chunk * R;
    const bool left_ok = x0 > 0, right_ok = x0 + 8 < W;
    // byte offset of (image b, row r0 - 1, pixel x0, this lane's channel) - may be "negative" for the row above the tensor,
    // which is never dereferenced (its loads get kRsInv)
    unsigned offx = ((unsigned)(b * H + r0 - 1) << wsh) * ldxb + (unsigned)x0 * ldxb + (unsigned)l31 * 4u;
    unsigned offy = ((unsigned)(b * H + r0) << wsh) * ldyb + (unsigned)x0 * ldyb + (unsigned)l31 * 4u;
    int yx = r0 - 1;          // image row the next X load reads
    int ny = 0;               // dY rows of the chunk loaded so far

    float rawx[2][10], rawy[2][8];
#define PIDM_RS_LOAD_X(dst_)                                                                                        \
  {                                                                                                                 \
    const bool ok__ = s_ok & (yx >= 0) & (yx < H) & (yx <= r0 + R);                                                 \
    const unsigned vc__ = ok__ ? offx : kRsInv, vp__ = (ok__ & left_ok) ? offx - ldxb : kRsInv,                     \
                   vn__ = (ok__ & right_ok) ? offx : kRsInv;                                                        \
    dst_[0] = pidm_buf_load_f32(rx, vp__, 0u);                                                                      \
    _Pragma("unroll") for (int j__ = 0; j__ < 8; ++j__) dst_[1 + j__] = pidm_buf_load_f32(rx, vc__, (unsigned)j__ * ldxb); \
    dst_[9] = pidm_buf_load_f32(rx, vn__, 8u * ldxb);                                                               \
    offx += rowxb;                                                                                                  \
    ++yx;                                                                                                           \
  }
#define PIDM_RS_LOAD_Y(dst_)                                                                                        \
  {                                                                                                                 \
    const unsigned vy__ = (s_ok & (ny < R)) ? offy : kRsInv;                                                        \
    _Pragma("unroll") for (int j__ = 0; j__ < 8; ++j__) dst_[j__] = pidm_buf_load_f32(ry, vy__, (unsigned)j__ * ldyb); \
    offy += rowyb;                                                                                                  \
    ++ny;                                                                                                           \
  }
    RsRow rows[3];
    u32x4 ya[2][3];
    {
      // prologue: rows r0 - 1 and r0 of X, row r0 of dY as pieces; X rows r0 + 1, r0 + 2 and dY rows r0 + 1, r0 + 2 in flight
      float ta[10], tb[10], ty[8];
      PIDM_RS_LOAD_X(ta)
      PIDM_RS_LOAD_X(tb)
      PIDM_RS_LOAD_Y(ty)
      PIDM_RS_LOAD_X(rawx[0])
      PIDM_RS_LOAD_Y(rawy[1])
      PIDM_RS_LOAD_X(rawx[1])
      PIDM_RS_LOAD_Y(rawy[0])
      rs_split_row(ta, rows[0]);
      rs_split_row(tb, rows[1]);
      const float sb = rs_split_dy(ty, ya[0]);
      if (do_bias) bacc += sb;
    }
    // ---- the rows of the chunk: iteration i uses X rows i - 1, i, i + 1 (pieces in rows[(i + 0 / 1 / 2) % 3]) and dY row i ----
    int i = 0;
    while (i < R) {
#pragma unroll
      for (int u = 0; u < 6; ++u) {
        if (i >= R) break;         // wave-uniform
        RsRow& ra = rows[u % 3];
        RsRow& rb = rows[(u + 1) % 3];
        RsRow& rc = rows[(u + 2) % 3];
        u32x4(&yc)[3] = ya[u & 1];
        u32x4(&yn)[3] = ya[(u + 1) & 1];
        rs_split_row(rawx[u & 1], rc);               // X row i + 1
        PIDM_RS_LOAD_X(rawx[u & 1])                  // X row i + 3
        PIDM_RS_SIX(acc[0], yc, ra.l)
        PIDM_RS_SIX(acc[1], yc, ra.c)
        PIDM_RS_SIX(acc[2], yc, ra.r)
        PIDM_RS_SIX(acc[3], yc, rb.l)
        PIDM_RS_SIX(acc[4], yc, rb.c)
        PIDM_RS_SIX(acc[5], yc, rb.r)
        const float sb = rs_split_dy(rawy[(u + 1) & 1], yn);   // dY row i + 1 (zeros past the chunk)
        PIDM_RS_LOAD_Y(rawy[(u + 1) & 1])            // dY row i + 3
        if (do_bias) bacc += sb;
        PIDM_RS_SIX(acc[6], yc, rc.l)
        PIDM_RS_SIX(acc[7], yc, rc.c)
        PIDM_RS_SIX(acc[8], yc, rc.r)
        ++i;
      }
    }
#undef PIDM_RS_LOAD_X
#undef PIDM_RS_LOAD_Y
  }

  // ---- sum of the 4 waves through LDS (fixed order), then the split's partial slab [split][m][tap][n] ----
#pragma unroll
  for (int tap = 0; tap < 9; ++tap)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = (r & 3) + 8 * (r >> 2) + 4 * half;
      red[((wave * 9 + tap) * 32 + row) * 32 + l31] = acc[tap][r];
    }
  __syncthreads();
  {
    const int mrow = tid >> 3, c4 = (tid & 7) * 4;
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
      const float* rp = red + (tap * 32 + mrow) * 32 + c4;
      const f32x4 a0 = *reinterpret_cast<const f32x4*>(rp), a1 = *reinterpret_cast<const f32x4*>(rp + 9 * 1024);
      const f32x4 a2 = *reinterpret_cast<const f32x4*>(rp + 18 * 1024), a3 = *reinterpret_cast<const f32x4*>(rp + 27 * 1024);
      const f32x4 sv = (a0 + a1) + (a2 + a3);
      *reinterpret_cast<f32x4*>(partial + (((size_t)split * wg.MP + (m0 + mrow)) * 9 + tap) * wg.NP + n0 + c4) = sv;
    }
  }
  if (do_bias) {
    __syncthreads();
    red[tid] = bacc;
    __syncthreads();
    if (tid < 32) {
      float sb = 0.f;
      for (int k = 0; k < 8; ++k) sb += red[k * 32 + tid];      // (wave, half) in fixed order
      bias_partial[(size_t)split * wg.MP + m0 + tid] = sb;
    }
  }
}

__global__ void __launch_bounds__(256) conv_wgrad_rs_kernel(WgradItem wg) {
  HIP_DYNAMIC_SHARED(float, red)      // epilogue only: [4 waves][9 taps][32 dY channels][32 X channels]
  conv_wgrad_rs_body(wg, (int)blockIdx.x, (int)blockIdx.y, red);
}

// The same for a TABLE of problems in one launch (round 5; WgradQueue, pidm_launch.h): workgroup blockIdx.x + blk_base belongs to the
// problem whose [blk0, blk0 + gx * gy) contains it - one load per lane and a ballot (the table has at most 64 rows) - and runs that
// problem's workgroup (local % gx, local / gx).  Consecutive workgroups of one problem still read neighbouring splits; the
// launch has no boundary between problems, so the chip drains once per flush instead of once per problem.
__global__ void __launch_bounds__(256) conv_wgrad_rs_multi_kernel(const WgradItem* __restrict__ table, int n, unsigned blk_base) {
  HIP_DYNAMIC_SHARED(float, red)
  const unsigned bid = blockIdx.x + blk_base;
  // (n <= 64, the queue's flush limit: lane l looks at row l, the rows' first workgroups ascend)
  const int lane_ = threadIdx.x & 63;
  const unsigned first_ = lane_ < n ? table[lane_].blk0 : 0xffffffffu;
  const int p = __builtin_amdgcn_readfirstlane(__popcll(__ballot(bid >= first_)) - 1);
  const WgradItem wg = table[p];
  const unsigned local = bid - wg.blk0;
  conv_wgrad_rs_body(wg, (int)(local % wg.gx), (int)(local / wg.gx), red);
}

// ---------------------------------------------------------------------------------------------------------------------
// The 4x4 / stride-2 / pad-1 layers (`Downsample`, src/unet_model.py:197; and - with the operands swapped by the caller - the
// transposed `Upsample`, :163) in the same row-streaming form:  dW[m][ky][kx][n] = sum dY[b,y,x][m] X[b, 2y + ky - 1, 2x + kx - 1][n].
// A k-step is again one row of 8 + 8 OUTPUT pixels; the 8 input pixels a tap needs lie two apart, so a row of X is loaded as its
// even-column fragment E[j] = X[r][2(x0 + j)], j = 0..8, and its odd-column fragment O[j] = X[r][2(x0 + j) + 1], j = -1..7 (18 dword
// loads, each two whole 128-byte lines), and   kx = 0: O[j - 1]   kx = 1: E[j]   kx = 2: O[j]   kx = 3: E[j + 1]
// - the shifted forms are the same registers moved by one bf16, as for the 3x3 taps.  A wave owns the 8 taps of two kernel rows
// (128 accumulator registers: ky = 2 kp, 2 kp + 1 read the input rows 2y + 2kp - 1 and 2y + 2kp); waves 0 / 1 of a workgroup take
// kp = 0 / 1 of the same strips, waves 2 / 3 the same for the workgroup's other strips; the two partners of a kernel-row pair
// are summed through LDS.  No rows are shared between consecutive k-steps of a wave (stride 2), so there is no rolling window:
// ~330 vector instructions per 48 MFMAs - vector-bound, and still 2x the fp32-MFMA kernel it replaces (conv_wgrad_pipe_kernel<2,2,..>).
// ---------------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ void conv_wgrad_rs4_body(const WgradItem& wg, const int split, const int by, float* __restrict__ red) {
  const WgradItem& g = wg;            // (the geometry fields the kernel reads live in the item itself)
  const float* __restrict__ src0 = wg.src0;
  const float* __restrict__ src1 = wg.src1;
  const float* __restrict__ dy = wg.dy;
  float* __restrict__ partial = wg.partial;
  float* __restrict__ bias_partial = wg.bias_partial;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int half = lane >> 5, l31 = lane & 31;
  const int ntn = wg.NP / 32;
  const int tn = by % ntn, tm = by / ntn;
  const int m0 = tm * 32, n0 = tn * 32;
  const int kp = wave & 1, pw = wave >> 1;                 // kernel-row pair, pixel share of the workgroup
  const int Ho = g.Hv, Wo = g.Wv, Hi = g.Hi, R = wg.rs_R;
  const int wsh = g.wsh;                                   // log2 Wo
  const unsigned ldxb = (unsigned)g.ld0 * 4u, ldyb = (unsigned)wg.ld_dy * 4u;
  const unsigned rowxb = ldxb << (wsh + 1), rowyb = ldyb << wsh;              // bytes between rows of X (2 Wo pixels) / of dY
  const unsigned tot_x = (unsigned)g.B * (unsigned)Hi * rowxb, tot_y = (unsigned)g.B * (unsigned)Ho * rowyb;
  const float* xsrc = (n0 < g.C0) ? src0 + n0 : src1 + (n0 - g.C0);
  const pidm_rsrc rx = pidm_make_rsrc(xsrc, tot_x), ry = pidm_make_rsrc(dy + m0, tot_y);
  const bool do_bias = (bias_partial != nullptr) && (tn == 0) && (kp == 0);

  f32x16 acc[8];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  float bacc = 0.f;

  const int wv = split * 2 + pw;                           // this pixel share among the 2 * nsplit of the block
  const int npairs = (wg.rs_S + 1) >> 1;
  int pair_lo = wv * wg.rs_ppw, pair_hi = pair_lo + wg.rs_ppw;
  if (pair_hi > npairs) pair_hi = npairs;

  for (int pair = pair_lo; pair < pair_hi; ++pair) {
    const int s = 2 * pair + half;
    const bool s_ok = s < wg.rs_S;
    const int xs = s & ((1 << wg.rs_xsh) - 1), t1 = s >> wg.rs_xsh;
    const int chunk = t1 & ((1 << wg.rs_csh) - 1), b = t1 >> wg.rs_csh;
    const int x0 = xs << 3, y0 = chunk * R;
    const bool left_ok = x0 > 0, right_ok = x0 + 8 < Wo;
    // X: (image b, input row 2 y0 + 2 kp - 1, input column 2 x0, this lane's channel); dY: (b, y0, x0, channel)
    int yx = 2 * y0 + 2 * kp - 1;
    unsigned offx = ((unsigned)(b * Hi + yx) << (wsh + 1)) * ldxb + (unsigned)(2 * x0) * ldxb + (unsigned)l31 * 4u;
    unsigned offy = ((unsigned)(b * Ho + y0) << wsh) * ldyb + (unsigned)x0 * ldyb + (unsigned)l31 * 4u;

    // one input row: E (even columns, + the one after) and O (odd columns, + the one before), as rs_split_row operands
    float re[2][2][10], ro[2][2][10], rawy[2][8];           // [buffer][row of the pair][...]
#define PIDM_RS4_LOAD_ROW(e_, o_, row_, voff_)                                                                      \
  {                                                                                                                 \
    const bool ok__ = s_ok & ((row_) >= 0) & ((row_) < Hi);                                                         \
    const unsigned vc__ = ok__ ? (voff_) : kRsInv, vp__ = (ok__ & left_ok) ? (voff_) - ldxb : kRsInv,               \
                   vn__ = (ok__ & right_ok) ? (voff_) : kRsInv;                                                     \
    e_[0] = 0.f;                                                                                                    \
    _Pragma("unroll") for (int j__ = 0; j__ < 8; ++j__) e_[1 + j__] = pidm_buf_load_f32(rx, vc__, (unsigned)(2 * j__) * ldxb); \
    e_[9] = pidm_buf_load_f32(rx, vn__, 16u * ldxb);                                                                \
    o_[0] = pidm_buf_load_f32(rx, vp__, 0u);                                                                        \
    _Pragma("unroll") for (int j__ = 0; j__ < 8; ++j__) o_[1 + j__] = pidm_buf_load_f32(rx, vc__, (unsigned)(2 * j__ + 1) * ldxb); \
    o_[9] = 0.f;                                                                                                    \
  }
#define PIDM_RS4_LOAD(buf_)                                                                                         \
  {                                                                                                                 \
    PIDM_RS4_LOAD_ROW(re[buf_][0], ro[buf_][0], yx, offx)                                                           \
    PIDM_RS4_LOAD_ROW(re[buf_][1], ro[buf_][1], yx + 1, offx + rowxb)                                               \
    const unsigned vy__ = (s_ok & (ny < R)) ? offy : kRsInv;                                                        \
    _Pragma("unroll") for (int j__ = 0; j__ < 8; ++j__) rawy[buf_][j__] = pidm_buf_load_f32(ry, vy__, (unsigned)j__ * ldyb); \
    offx += 2u * rowxb;                                                                                             \
    yx += 2;                                                                                                        \
    offy += rowyb;                                                                                                  \
    ++ny;                                                                                                           \
  }
    int ny = 0;
    PIDM_RS4_LOAD(0)
    int i = 0;
    while (i < R) {
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        if (i >= R) break;         // wave-uniform
        PIDM_RS4_LOAD(u ^ 1)                          // the next output row's operands (zeros past the chunk)
        RsRow e0, o0, e1, o1;
        u32x4 ya[3];
        rs_split_row(re[u][0], e0);
        rs_split_row(ro[u][0], o0);
        rs_split_row(re[u][1], e1);
        rs_split_row(ro[u][1], o1);
        const float sb = rs_split_dy(rawy[u], ya);
        if (do_bias) bacc += sb;
        PIDM_RS_SIX(acc[0], ya, o0.l)                // ky = 2 kp:     kx = 0 .. 3
        PIDM_RS_SIX(acc[1], ya, e0.c)
        PIDM_RS_SIX(acc[2], ya, o0.c)
        PIDM_RS_SIX(acc[3], ya, e0.r)
        PIDM_RS_SIX(acc[4], ya, o1.l)                // ky = 2 kp + 1
        PIDM_RS_SIX(acc[5], ya, e1.c)
        PIDM_RS_SIX(acc[6], ya, o1.c)
        PIDM_RS_SIX(acc[7], ya, e1.r)
        ++i;
      }
    }
#undef PIDM_RS4_LOAD
#undef PIDM_RS4_LOAD_ROW
  }

  // ---- the two pixel shares of each kernel-row pair summed through LDS, then the split's partial slab [split][m][16 taps][n] ----
#pragma unroll
  for (int t = 0; t < 8; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = (r & 3) + 8 * (r >> 2) + 4 * half;
      red[((wave * 8 + t) * 32 + row) * 32 + l31] = acc[t][r];
    }
  __syncthreads();
  {
    const int mrow = tid >> 3, c4 = (tid & 7) * 4;
#pragma unroll
    for (int tap = 0; tap < 16; ++tap) {
      const int kpt = tap >> 3, t = tap & 7;           // tap = ky * 4 + kx = (2 kp + (t >> 2)) * 4 + (t & 3) = 8 kp + t
      const float* rp = red + (((kpt) * 8 + t) * 32 + mrow) * 32 + c4;
      const f32x4 a0 = *reinterpret_cast<const f32x4*>(rp), a1 = *reinterpret_cast<const f32x4*>(rp + 2 * 8 * 1024);
      const f32x4 sv = a0 + a1;
      *reinterpret_cast<f32x4*>(partial + (((size_t)split * wg.MP + (m0 + mrow)) * 16 + tap) * wg.NP + n0 + c4) = sv;
    }
  }
  if ((bias_partial != nullptr) && (tn == 0)) {
    __syncthreads();
    red[tid] = bacc;               // zero in the kp = 1 waves
    __syncthreads();
    if (tid < 32) {
      float sb = 0.f;
      for (int k = 0; k < 8; ++k) sb += red[k * 32 + tid];
      bias_partial[(size_t)split * wg.MP + m0 + tid] = sb;
    }
  }
}

__global__ void __launch_bounds__(256) conv_wgrad_rs4_kernel(WgradItem wg) {
  HIP_DYNAMIC_SHARED(float, red)      // epilogue only: [4 waves][8 taps][32][32]
  conv_wgrad_rs4_body(wg, (int)blockIdx.x, (int)blockIdx.y, red);
}
// a table of problems in one launch (see conv_wgrad_rs_multi_kernel)
__global__ void __launch_bounds__(256) conv_wgrad_rs4_multi_kernel(const WgradItem* __restrict__ table, int n, unsigned blk_base) {
  HIP_DYNAMIC_SHARED(float, red)
  const unsigned bid = blockIdx.x + blk_base;
  const int lane_ = threadIdx.x & 63;
  const unsigned first_ = lane_ < n ? table[lane_].blk0 : 0xffffffffu;
  const int p = __builtin_amdgcn_readfirstlane(__popcll(__ballot(bid >= first_)) - 1);
  const WgradItem wg = table[p];
  const unsigned local = bid - wg.blk0;
  conv_wgrad_rs4_body(wg, (int)(local % wg.gx), (int)(local / wg.gx), red);
}

// ---------------------------------------------------------------------------------------------------------------------
// Few input channels, many taps (the 7x7 `init_conv`, src/unet_model.py:453: Cin = 2, or 10 for the mechanics model): the GEMM N
// dimension is the flattened (tap, channel) index, as in conv_wgrad_smallc_kernel (98 columns for 7x7x2 instead of 49 taps x a
// 32-channel tile that is 94 % padding).  Row-streaming form: lane j of an n-tile IS column (ky, kx, c) and loads its own 8 pixels
// X[y + ky - pad][x0 + kx - pad + 0..7][c] (neighbouring lanes read neighbouring floats: a tile's loads are a few contiguous
// runs); rows / columns outside the image are never fetched (per-load offset select).  The 4 waves of a workgroup walk the SAME
// pixels and own different n-tiles (wave w: tiles w, w + 4, ..), so the epilogue needs no cross-wave sum; the dY fragment is
// split by every wave (48 vector instructions).  Vector-bound (~120 instructions per 6 MFMAs and n-tile) - and 3-5x the fp32-MFMA
// LDS kernel it replaces, whose 64 two-deep k-steps per 128 pixels were the cost.
// ---------------------------------------------------------------------------------------------------------------------
template <int MAXN>
__global__ void __launch_bounds__(256) conv_wgrad_rs7_kernel(WgradGeom wg, const float* __restrict__ src0, const float* __restrict__ dy,
                                                             float* __restrict__ partial, float* __restrict__ bias_partial) {
  const ConvGeom& g = wg.g;
  __shared__ float red[256];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int half = lane >> 5, l31 = lane & 31;
  const int T = g.KH * g.KW, Cin = g.Cin, NJ = T * Cin;
  const int m0 = blockIdx.y * 32;
  const int split = blockIdx.x;
  const int H = g.Hi, W = g.Wi, R = wg.rs_R;
  const int wsh = g.wsh;
  const unsigned ldxb = (unsigned)g.ld0 * 4u, ldyb = (unsigned)wg.ld_dy * 4u;
  const unsigned rowxb = ldxb << wsh, rowyb = ldyb << wsh;
  const unsigned tot_x = (unsigned)g.B * (unsigned)H * rowxb, tot_y = (unsigned)g.B * (unsigned)H * rowyb;
  const pidm_rsrc rx = pidm_make_rsrc(src0, tot_x), ry = pidm_make_rsrc(dy + m0, tot_y);
  const bool do_bias = (bias_partial != nullptr) && (wave == 0);

  // this lane's column of each of the wave's n-tiles: (tap row, tap column, channel) -> (dy, dx, c)
  int jdy[MAXN], jdx[MAXN], jt[MAXN], jc[MAXN];
  bool jok[MAXN];
#pragma unroll
  for (int i = 0; i < MAXN; ++i) {
    const int j = (wave + 4 * i) * 32 + l31;
    jok[i] = j < NJ;
    const int t = jok[i] ? j / Cin : 0, c = jok[i] ? j - t * Cin : 0;
    jt[i] = t; jc[i] = c;
    jdy[i] = t / g.KW - g.pad_y[0];
    jdx[i] = t % g.KW - g.pad_x[0];
  }
  f32x16 acc[MAXN];
#pragma unroll
  for (int i = 0; i < MAXN; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  float bacc = 0.f;

  const int npairs = (wg.rs_S + 1) >> 1;
  int pair_lo = split * wg.rs_ppw, pair_hi = pair_lo + wg.rs_ppw;
  if (pair_hi > npairs) pair_hi = npairs;
  for (int pair = pair_lo; pair < pair_hi; ++pair) {
    const int s = 2 * pair + half;
    const bool s_ok = s < wg.rs_S;
    const int xs = s & ((1 << wg.rs_xsh) - 1), t1 = s >> wg.rs_xsh;
    const int chunk = t1 & ((1 << wg.rs_csh) - 1), b = t1 >> wg.rs_csh;
    const int x0 = xs << 3, y0 = chunk * R;
    unsigned offy = ((unsigned)(b * H + y0) << wsh) * ldyb + (unsigned)x0 * ldyb + (unsigned)l31 * 4u;
    unsigned offx[MAXN];
    bool cok[MAXN][8];
#pragma unroll
    for (int i = 0; i < MAXN; ++i) {
      offx[i] = (unsigned)(((b * H + y0 + jdy[i]) << wsh) + x0 + jdx[i]) * ldxb + (unsigned)jc[i] * 4u;
#pragma unroll
      for (int jj = 0; jj < 8; ++jj) cok[i][jj] = s_ok & jok[i] & ((unsigned)(x0 + jdx[i] + jj) < (unsigned)W);
    }
    // rows of the chunk, the next row's operands in flight while the current one is split and multiplied (two register sets)
    float vy[2][8], vx[2][MAXN][8];
    int yl = y0;                   // row the next load reads
#define PIDM_RS7_LOAD(buf_)                                                                                         \
  {                                                                                                                 \
    const unsigned voy__ = (s_ok & (yl < y0 + R)) ? offy : kRsInv;                                                  \
    _Pragma("unroll") for (int jj = 0; jj < 8; ++jj) vy[buf_][jj] = pidm_buf_load_f32(ry, voy__, (unsigned)jj * ldyb); \
    offy += rowyb;                                                                                                  \
    _Pragma("unroll") for (int i = 0; i < MAXN; ++i) {                                                              \
      const bool rok__ = ((unsigned)(yl + jdy[i]) < (unsigned)H) & (yl < y0 + R);                                   \
      _Pragma("unroll") for (int jj = 0; jj < 8; ++jj)                                                              \
          vx[buf_][i][jj] = pidm_buf_load_f32(rx, (rok__ & cok[i][jj]) ? offx[i] + (unsigned)jj * ldxb : kRsInv, 0u); \
      offx[i] += rowxb;                                                                                             \
    }                                                                                                               \
    ++yl;                                                                                                           \
  }
    PIDM_RS7_LOAD(0)
    int y = y0;
    while (y < y0 + R) {
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        if (y >= y0 + R) break;          // wave-uniform
        PIDM_RS7_LOAD(u ^ 1)
        u32x4 ya[3];
        const float sb = rs_split_dy(vy[u], ya);
        if (do_bias) bacc += sb;
#pragma unroll
        for (int i = 0; i < MAXN; ++i) {
          if ((wave + 4 * i) * 32 < NJ) {          // wave-uniform
            u32x4 xb[3];
            rs_split_dy(vx[u][i], xb);
            PIDM_RS_SIX(acc[i], ya, xb)
          }
        }
        ++y;
      }
    }
#undef PIDM_RS7_LOAD
  }
  // results: column j = (t, c) of this lane, rows = 32 dY channels: the split's slab [split][m][T][Cin] (conv_wgrad_smallc_kernel's)
#pragma unroll
  for (int i = 0; i < MAXN; ++i) {
    if (jok[i]) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = (r & 3) + 8 * (r >> 2) + 4 * half;
        partial[(((size_t)split * wg.MP + (m0 + row)) * T + jt[i]) * wg.NP + jc[i]] = acc[i][r];
      }
    }
  }
  if (bias_partial != nullptr) {
    red[tid] = bacc;               // zero outside wave 0
    __syncthreads();
    if (tid < 32) bias_partial[(size_t)split * wg.MP + m0 + tid] = red[tid] + red[32 + tid];
  }
}

// Plan: rows per strip chunk R (power of two <= H) such that the waves of a block (4 per split) get whole items and the longest
// wave - items x (R rows + the three rows of prologue) - is shortest.
static constexpr size_t kRsWgradLds = 4 * 9 * 1024 * sizeof(float);
static constexpr size_t kRs4WgradLds = 4 * 8 * 1024 * sizeof(float);
WgradItem wgrad_item(const WgradGeom& wg, const float* src0, const float* src1, const float* dy, float* partial, float* bias_partial,
                     unsigned gx, unsigned gy, int kind);
int wgrad_group_splitdiv();

static void rs_plan(WgradGeom* wg, int nsplit) {
  const ConvGeom& g = wg->g;
  const int waves = 4 * nsplit;
  int xsh = 0;
  while ((8 << xsh) < g.Wi) ++xsh;
  long best_cost = -1;
  for (int R = g.Hi; R >= 1; R >>= 1) {
    const int cpi = g.Hi / R;
    const long S = (long)g.B * cpi * (g.Wi / 8), pairs = (S + 1) / 2, ppw = (pairs + waves - 1) / waves;
    const long cost = ppw * (R + 3);
    if (best_cost < 0 || cost < best_cost) {
      best_cost = cost;
      int csh = 0;
      while ((1 << csh) < cpi) ++csh;
      wg->rs_R = R; wg->rs_csh = csh; wg->rs_xsh = xsh; wg->rs_S = (int)S; wg->rs_ppw = (int)ppw;
    }
  }
  wg->nsplit = nsplit;
}

static void rs4_plan(WgradGeom* wg, int nsplit) {
  const ConvGeom& g = wg->g;
  const int shares = 2 * nsplit;                 // pixel shares of a block: two per workgroup
  int xsh = 0;
  while ((8 << xsh) < g.Wv) ++xsh;
  long best_cost = -1;
  for (int R = g.Hv; R >= 1; R >>= 1) {
    const int cpi = g.Hv / R;
    const long S = (long)g.B * cpi * (g.Wv / 8), pairs = (S + 1) / 2, ppw = (pairs + shares - 1) / shares;
    const long cost = ppw * (R + 1);
    if (best_cost < 0 || cost < best_cost) {
      best_cost = cost;
      int csh = 0;
      while ((1 << csh) < cpi) ++csh;
      wg->rs_R = R; wg->rs_csh = csh; wg->rs_xsh = xsh; wg->rs_S = (int)S; wg->rs_ppw = (int)ppw;
    }
  }
  wg->nsplit = nsplit;
}

bool wgrad_rs4_eligible(const ConvGeom& g, int ld_dy) {
  auto pow2 = [](int v) { return v > 0 && (v & (v - 1)) == 0; };
  const size_t pix_x = (size_t)g.B * g.Hi * g.Wi, pix_y = (size_t)g.B * g.Hv * g.Wv;
  return g.nph == 4 && g.nz == 1 && g.KH == 2 && g.KW == 2 && g.in_step == 2 && 2 * g.Wv == g.Wi && 2 * g.Hv == g.Hi && g.Wv >= 8 &&
         pow2(g.Wv) && pow2(g.Hv) && (g.Cin % 32 == 0) && (g.C0 % 32 == 0) && (g.Cout % 32 == 0) && (g.C1 == 0 || g.ld1 == g.ld0) &&
         pix_x * (size_t)g.ld0 * 4 < 0x7ff00000ull && pix_y * (size_t)ld_dy * 4 < 0x7ff00000ull;
}

// the 4x4 / stride-2 layers (phased geometry, ConvGeom::nph == 4); same contract as launch_wgrad_rs
bool launch_wgrad_rs4(const WgradGeom& plan, const float* src0, const float* src1, const float* dy, int ld_dy, float* partial,
                      float* bias_partial, hipStream_t st, WgradGeom* used, WgradQueue* wq) {
  const ConvGeom& g = plan.g;
  const char* off = knob("PIDM_WGRAD_RS");
  if (off && !atoi(off)) return false;
  if (!wgrad_rs4_eligible(g, ld_dy) || (reinterpret_cast<size_t>(src0) & 3) || (reinterpret_cast<size_t>(dy) & 3)) return false;
  WgradGeom wg = plan;
  wg.ld_dy = ld_dy;
  int ns = plan.nsplit;
  const char* me = knob("PIDM_WGRAD_SPLIT_MAXNS");
  if (me && atoi(me) > 0 && atoi(me) < ns) ns = atoi(me);
  if (wq) {                                   // grouped: fewer, longer items (see launch_wgrad_rs)
    const int dv = wgrad_group_splitdiv();
    if (dv > 1 && ns / dv >= 1) ns /= dv;
  }
  rs4_plan(&wg, ns);
  const dim3 grid(wg.nsplit, (wg.MP / 32) * (wg.NP / 32), 1);
  static bool attr_ = false;
  if (!attr_) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_wgrad_rs4_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kRs4WgradLds);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_wgrad_rs4_multi_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kRs4WgradLds);
    attr_ = true;
  }
  if (knob("PIDM_TRACE_CONV"))
    fprintf(stderr, "[pidm]   -> conv_wgrad_rs4_kernel%s, %d splits x %d blocks, %d strips of %d rows, %d pairs per pixel share\n",
            wq ? " (queued)" : "", wg.nsplit, grid.y, wg.rs_S, wg.rs_R, wg.rs_ppw);
  const WgradItem it = wgrad_item(wg, src0, src1, dy, partial, bias_partial, grid.x, grid.y, 0);
  *used = wg;
  if (wq) {
    wq->push(kWgFamRs4, it, 2.0 * g.B * g.Hv * g.Wv * (double)g.Cout * g.Cin * 16);
    return true;
  }
  PIDM_PROF_NAME("conv_wgrad_rs4_kernel");
  hipLaunchKernelGGL(conv_wgrad_rs4_kernel, grid, dim3(256), kRs4WgradLds, st, it);
  return true;
}

bool wgrad_rs7_eligible(const ConvGeom& g, int ld_dy) {
  auto pow2 = [](int v) { return v > 0 && (v & (v - 1)) == 0; };
  const size_t pix = (size_t)g.B * g.Hi * g.Wi;
  const int NJ = g.KH * g.KW * g.Cin;
  return g.nph == 1 && g.nz == 1 && g.C1 == 0 && g.stride == 1 && g.Cin <= 16 && g.KH * g.KW > 1 && NJ <= 16 * 32 && g.Wv == g.Wi &&
         g.Hv == g.Hi && g.Wi >= 8 && pow2(g.Wi) && pow2(g.Hi) && (g.Cout % 32 == 0) && pix * (size_t)g.ld0 * 4 < 0x7ff00000ull &&
         pix * (size_t)ld_dy * 4 < 0x7ff00000ull;
}

// few input channels, many taps (wgrad_smallc geometry); same contract as launch_wgrad_rs, partial layout of conv_wgrad_smallc_kernel
bool launch_wgrad_rs7(const WgradGeom& plan, const float* src0, const float* dy, int ld_dy, float* partial, float* bias_partial,
                      hipStream_t st, WgradGeom* used) {
  const ConvGeom& g = plan.g;
  const char* off = knob("PIDM_WGRAD_RS");
  if (off && !atoi(off)) return false;
  if (!wgrad_rs7_eligible(g, ld_dy) || (reinterpret_cast<size_t>(src0) & 3) || (reinterpret_cast<size_t>(dy) & 3)) return false;
  WgradGeom wg = plan;
  wg.ld_dy = ld_dy;
  int ns = plan.nsplit;
  const int blocks = wg.MP / 32;
  const int ntl0 = (g.KH * g.KW * g.Cin + 31) / 32;
  // no software prefetch in this kernel: the load latency is covered by residency - 82 registers with one n-tile per wave
  // (4 workgroups per CU), 173 with four (2 per CU)
  const int slots = ntl0 <= 4 ? 1024 : 512;
  if (ns * blocks > slots) ns = slots / blocks > 0 ? slots / blocks : 1;
  const char* me = knob("PIDM_WGRAD_SPLIT_MAXNS");
  if (me && atoi(me) > 0 && atoi(me) < ns) ns = atoi(me);
  // every wave of a workgroup walks the workgroup's pairs: `shares` = workgroups of a block
  {
    int xsh = 0;
    while ((8 << xsh) < g.Wi) ++xsh;
    long best_cost = -1;
    for (int R = g.Hi; R >= 1; R >>= 1) {
      const int cpi = g.Hi / R;
      const long S = (long)g.B * cpi * (g.Wi / 8), pairs = (S + 1) / 2, ppw = (pairs + ns - 1) / ns;
      const long cost = ppw * (R + 1);
      if (best_cost < 0 || cost < best_cost) {
        best_cost = cost;
        int csh = 0;
        while ((1 << csh) < cpi) ++csh;
        wg.rs_R = R; wg.rs_csh = csh; wg.rs_xsh = xsh; wg.rs_S = (int)S; wg.rs_ppw = (int)ppw;
      }
    }
    wg.nsplit = ns;
  }
  const int ntl = (g.KH * g.KW * g.Cin + 31) / 32, maxn = (ntl + 3) / 4;
  const dim3 grid(wg.nsplit, blocks, 1);
  if (knob("PIDM_TRACE_CONV"))
    fprintf(stderr, "[pidm]   -> conv_wgrad_rs7_kernel<%d>, %d splits x %d blocks, %d strips of %d rows, %d pairs per workgroup\n", maxn <= 1 ? 1 : 4,
            wg.nsplit, grid.y, wg.rs_S, wg.rs_R, wg.rs_ppw);
  PIDM_PROF_NAME("conv_wgrad_rs7_kernel");
  if (maxn <= 1)
    hipLaunchKernelGGL(HIP_KERNEL_NAME(conv_wgrad_rs7_kernel<1>), grid, dim3(256), 0, st, wg, src0, dy, partial, bias_partial);
  else
    hipLaunchKernelGGL(HIP_KERNEL_NAME(conv_wgrad_rs7_kernel<4>), grid, dim3(256), 0, st, wg, src0, dy, partial, bias_partial);
  *used = wg;
  return true;
}

// true: launched (and *used holds the split actually written); false: geometry not eligible (the caller goes on to the LDS-staged
// kernel).  PIDM_WGRAD_RS=0: off (A/B measurements, tests of the older kernel).
WgradItem wgrad_item(const WgradGeom& wg, const float* src0, const float* src1, const float* dy, float* partial, float* bias_partial,
                     unsigned gx, unsigned gy, int kind) {
  const ConvGeom& g = wg.g;
  WgradItem it{};
  it.src0 = src0; it.src1 = src1 ? src1 : src0; it.dy = dy; it.partial = partial; it.bias_partial = bias_partial;
  it.B = g.B; it.Hi = g.Hi; it.Wi = g.Wi; it.wsh = g.wsh; it.Hv = g.Hv; it.Wv = g.Wv;
  it.ld0 = g.ld0; it.ld1 = g.ld1; it.C0 = g.C0; it.Cin = g.Cin; it.Cout = g.Cout; it.ld_dy = wg.ld_dy;
  it.MP = wg.MP; it.NP = wg.NP; it.tiles_per_split = wg.tiles_per_split;
  it.rs_R = wg.rs_R; it.rs_csh = wg.rs_csh; it.rs_xsh = wg.rs_xsh; it.rs_S = wg.rs_S; it.rs_ppw = wg.rs_ppw;
  it.kind = kind;
  it.gx = gx; it.gy = gy;
  return it;
}
int wgrad_group_splitdiv() {
  const char* de = knob("PIDM_WGRAD_GROUP_SPLITDIV");
  const int dv = de ? atoi(de) : 4;
  return dv < 1 ? 1 : dv;
}

// true: launched - or, with `wq`, queued for the grouped launch - (and *used holds the split actually written); false: geometry not
// eligible (the caller goes on to the LDS-staged kernel).  PIDM_WGRAD_RS=0: off (A/B measurements, tests of the older kernel).
bool wgrad_rs_queueable(const ConvGeom& g, const float* src0, const float* dy, int ld_dy) {
  const char* off = knob("PIDM_WGRAD_RS");
  if (off && !atoi(off)) return false;
  auto pow2 = [](int v) { return v > 0 && (v & (v - 1)) == 0; };
  if (!(g.KH == 3 && g.KW == 3 && g.stride == 1 && g.nph == 1 && g.nz == 1 && g.pad_y[0] == 1 && g.pad_x[0] == 1 && g.Wv == g.Wi &&
        g.Hv == g.Hi && g.Wi >= 8 && pow2(g.Wi) && pow2(g.Hi) && (g.Cin % 32 == 0) && (g.C0 % 32 == 0) && (g.Cout % 32 == 0) &&
        (g.C1 == 0 || g.ld1 == g.ld0) && (reinterpret_cast<size_t>(src0) & 3) == 0 && (reinterpret_cast<size_t>(dy) & 3) == 0))
    return false;
  const size_t pix = (size_t)g.B * g.Hi * g.Wi;
  return pix * (size_t)g.ld0 * 4 < 0x7ff00000ull && pix * (size_t)ld_dy * 4 < 0x7ff00000ull;   // 32-bit byte offsets
}

bool launch_wgrad_rs(const WgradGeom& plan, const float* src0, const float* src1, const float* dy, int ld_dy, float* partial,
                     float* bias_partial, hipStream_t st, WgradGeom* used, WgradQueue* wq) {
  const ConvGeom& g = plan.g;
  if (!wgrad_rs_queueable(g, src0, dy, ld_dy)) return false;
  WgradGeom wg = plan;
  wg.ld_dy = ld_dy;
  int ns = plan.nsplit;                     // never more splits than the workspace was sized for
  const char* me = knob("PIDM_WGRAD_SPLIT_MAXNS");   // tests: several items per wave on small problems
  if (me && atoi(me) > 0 && atoi(me) < ns) ns = atoi(me);
  // Grouped launches need not fill the chip problem by problem - the ~40 problems of a pass do it together: PIDM_WGRAD_GROUP_SPLITDIV
  // (default 4) divides the split count: fewer, longer work items (a workgroup's prologue, its 147 KB cross-wave sum and its 36 KB
  // partial slab are per item), proportionally fewer partial slabs for the deferred reduction to read.  Measured per step, same box
  // (profiles/r05_b_wgrad_group_splitdiv.txt): batch 16 5.08 -> 4.80 / 4.74 / 4.80 / 4.85 ms for 2 / 4 / 8 / 16, batch 64 8.28 ->
  // 8.07 / 8.03 / 8.09 / 8.13, batch 256 24.81 -> 24.85 / 24.78 / 25.07 / 25.17 (32: slower everywhere).
  if (wq) {
    const int dv = wgrad_group_splitdiv();
    if (dv > 1 && ns / dv >= 1) ns /= dv;
  }
  rs_plan(&wg, ns);
  const WgradItem it = wgrad_item(wg, src0, src1, dy, partial, bias_partial, (unsigned)ns, (unsigned)((wg.MP / 32) * (wg.NP / 32)), 0);
  static bool attr_ = false;
  if (!attr_) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_wgrad_rs_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kRsWgradLds);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_wgrad_rs_multi_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kRsWgradLds);
    attr_ = true;
  }
  if (knob("PIDM_TRACE_CONV"))
    fprintf(stderr, "[pidm]   -> conv_wgrad_rs_kernel%s, %d splits x %u blocks, %d strips of %d rows, %d pairs per wave\n", wq ? " (queued)" : "",
            wg.nsplit, it.gy, wg.rs_S, wg.rs_R, wg.rs_ppw);
  *used = wg;
  if (wq) {
    wq->push(kWgFamRs, it, 2.0 * g.B * g.Hv * g.Wv * (double)g.Cout * g.Cin * 9);
    return true;
  }
  PIDM_PROF_NAME("conv_wgrad_rs_kernel");
  hipLaunchKernelGGL(conv_wgrad_rs_kernel, dim3(it.gx, it.gy, 1), dim3(256), kRsWgradLds, st, it);
  return true;
}

// rows [first, first + n) of the device table, whose workgroups are [blk_base, blk_base + nblocks) of the queue's flat numbering
int launch_wgrad_rs_multi(const WgradItem* table_dev, int first, int n, unsigned blk_base, unsigned nblocks, hipStream_t st) {
  if (n <= 0 || nblocks == 0) return 0;
  PIDM_PROF_NAME("conv_wgrad_rs_multi_kernel");
  hipLaunchKernelGGL(conv_wgrad_rs_multi_kernel, dim3(nblocks), dim3(256), kRsWgradLds, st, table_dev + first, n, blk_base);
  PIDM_CHECK_LAUNCH("conv_wgrad_rs_multi_kernel");
  return 0;
}

int launch_wgrad_rs4_multi(const WgradItem* table_dev, int first, int n, unsigned blk_base, unsigned nblocks, hipStream_t st) {
  if (n <= 0 || nblocks == 0) return 0;
  PIDM_PROF_NAME("conv_wgrad_rs4_multi_kernel");
  hipLaunchKernelGGL(conv_wgrad_rs4_multi_kernel, dim3(nblocks), dim3(256), kRs4WgradLds, st, table_dev + first, n, blk_base);
  PIDM_CHECK_LAUNCH("conv_wgrad_rs4_multi_kernel");
  return 0;
}

int launch_wgrad_multi(int fam, const WgradItem* table_dev, int first, int n, unsigned blk_base, unsigned nblocks, hipStream_t st) {
  return fam == kWgFamRs    ? launch_wgrad_rs_multi(table_dev, first, n, blk_base, nblocks, st)
         : fam == kWgFamRs4 ? launch_wgrad_rs4_multi(table_dev, first, n, blk_base, nblocks, st)
         : fam == kWgFam1x1Split ? launch_wgrad_1x1_split_multi(table_dev, first, n, blk_base, nblocks, st)
                            : launch_wgrad_1x1_multi(table_dev, first, n, blk_base, nblocks, st);
}

}  // namespace pidm
