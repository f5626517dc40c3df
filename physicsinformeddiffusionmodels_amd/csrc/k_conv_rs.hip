// Row-streaming 3x3 / stride-1 convolution (forward and input gradient) for the wide levels of the UNet (rows of 32 or 64 pixels,
// 32 or 64 input channels): the split-form contraction of conv3x3_split_kernel (k_conv.hip: fp32 operands as three bf16 pieces, six
// v_mfma_f32_32x32x16_bf16 per fp32 product, small terms first, even / odd taps in two accumulator chains) with the ACTIVATIONS
// NEVER PASSING THROUGH LDS.  Replaces the `convolution` ATen op of `Block.proj` at the 64x64 and 32x32 levels
// (/root/reference/src/unet_model.py:227 via :236) and its input gradient.
//
// Why: at these levels a tile of conv3x3_split_kernel has only 2-4 stages of 16 input channels, so its per-stage staging (global ->
// registers -> split -> LDS -> barrier -> fragment reads: ~1950 of a stage's ~5800 cycles, profiles/r02_split_conv_notes.txt) and
// its per-tile epilogue with both waves of a SIMD lined up by the barrier (~2200 cycles every 2-4 stages) keep the matrix pipe at
// 0.32-0.45 occupancy (profiles/r04_pmc_split_kernels.txt).  The operand fragment of the 32x32x16 MFMA is "lane = pixel, 8
// consecutive channels" - in a channels-last image that is 32 contiguous bytes of the lane's own pixel - so a wave can load its
// fragments straight from global memory, split them in registers and keep going: no staging buffer, no barrier, no tile.
//
// A wave owns a strip of 32 pixels x R output rows of one image and walks the R + 2 input rows top to bottom.  For input row r it
// loads the row three times (shifted by -1 / 0 / +1 pixels: the kx taps; the re-reads hit L1), splits each 16-channel chunk into its
// pieces and issues, for ky = 0..2, the six MFMAs of tap (ky, kx) into the accumulators of output row r + 1 - ky: three output rows
// are live (x 2 chains x NT n-tiles x 16 registers), the one that received its last row (ky = 2) is finished, stored and zeroed.
// Zero padding comes from the buffer descriptor: out-of-image columns use an out-of-range offset (reads as 0, pidm_common.h),
// out-of-image rows are skipped.  The pre-split weights of the workgroup's n-tiles (the packing of conv3x3_split_kernel: an image of
// 112-byte LDS rows) are copied to LDS once per workgroup by global_load_lds; a wave reads each tap's fragments once per input row.
// The only barrier of the kernel publishes the weights.  Epilogue per output row = the tile epilogue of conv3x3_split_kernel (bias,
// GroupNorm partial sums, GroupNorm-backward sums, 4x4 register transposes, residual, 16-byte stores).
// Results: same pieces, same products, same two chains per output element as conv3x3_split_kernel, but the chains run tap-major
// instead of chunk-major - equal to it within fp32 rounding of the accumulation order, not bit for bit.
#include <stdio.h>
#include <stdlib.h>

#include "pidm_launch.h"
#include "k_conv_epilogue.h"

namespace pidm {

static constexpr int kRsRow = 112;                  // bytes per LDS weight row (k_conv.hip: kSplitRow)
static constexpr int kRsSlab = 9 * 32 * kRsRow;     // pre-split weights of one (n-tile, 16-channel chunk)
static constexpr unsigned kRsOob = 0x80000000u;     // a byte offset no tensor reaches (the launcher checks): reads as 0

#define PIDM_RSF_MFMA6(acc_, a_, b_)                          \
  acc_ = pidm_mfma_bf16_32x32x16(a_[2], b_[0], acc_);         \
  acc_ = pidm_mfma_bf16_32x32x16(a_[0], b_[2], acc_);         \
  acc_ = pidm_mfma_bf16_32x32x16(a_[1], b_[1], acc_);         \
  acc_ = pidm_mfma_bf16_32x32x16(a_[1], b_[0], acc_);         \
  acc_ = pidm_mfma_bf16_32x32x16(a_[0], b_[1], acc_);         \
  acc_ = pidm_mfma_bf16_32x32x16(a_[0], b_[0], acc_);

// NCH = Cin / 16 (2 or 4), NT = n-tiles of 32 output channels per wave (1 or 2), WPS = waves per SIMD the register budget is cut for,
// RM = R % 3 (1 or 2: R is a power of two >= 4; it fixes which accumulator slot the last rows of a strip use)
template <int NCH, int NT, int WPS, int RM>
__global__ void __launch_bounds__(256) PIDM_WAVES_PER_SIMD(WPS)
conv3x3_rs_kernel(ConvGeom g, const float* __restrict__ src0, const float* __restrict__ src1, const unsigned short* __restrict__ ws,
                  const float* __restrict__ bias, const float* __restrict__ residual, float* __restrict__ out, int R, int n_units,
                  unsigned src_bytes) {
  HIP_DYNAMIC_SHARED(float, smemf)
  char* smem = reinterpret_cast<char*>(smemf);
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int half = lane >> 5, l31 = lane & 31;
  const int ng = blockIdx.y;                         // n-group: n-tiles ng * NT ...
  {
    // the n-group's slabs are contiguous in the packing: [n-tile][chunk][tap][32 rows x 112 bytes]
    const char* wsrc = reinterpret_cast<const char*>(ws) + (size_t)ng * (NT * NCH) * kRsSlab;
    constexpr int NK = NT * NCH * kRsSlab / 1024;
    for (int k = wave; k < NK; k += 4) pidm_glds_b128(wsrc + 1024 * k + 16 * lane, smem + 1024 * k);
  }
  const int unit = blockIdx.x * 4 + wave;
  const bool live = unit < n_units;                  // wave-uniform
  const int nrb = g.Hv / R, nsx = g.Wv >> 5;
  int u = live ? unit : 0;
  const int rb = u % nrb;
  u /= nrb;
  const int sx = u % nsx, b = u / nsx;
  const int y0 = rb * R, x0 = sx * 32;
  const unsigned ldb = (unsigned)g.ld0 * 4u;
  // one descriptor per 16-channel chunk: the chunk's first channel inside whichever source holds it (concatenated inputs)
  pidm_rsrc rsc[NCH];
#pragma unroll
  for (int ch = 0; ch < NCH; ++ch) rsc[ch] = pidm_make_rsrc((ch * 16 < g.C0) ? src0 + ch * 16 : src1 + (ch * 16 - g.C0), src_bytes);
  float bvs[NT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) bvs[nt] = bias ? bias[(ng * NT + nt) * 32 + l31] : 0.f;
  // per-lane byte offset of the lane's 8 channels inside a row, per kx (out-of-image columns: out of range)
  unsigned voff[3];
#pragma unroll
  for (int kx = 0; kx < 3; ++kx) {
    const int x = x0 + l31 + kx - 1;
    voff[kx] = (live && x >= 0 && x < g.Wv) ? (unsigned)x * ldb + 32u * (unsigned)half : kRsOob;
  }
  const unsigned img_off = (unsigned)b * (unsigned)g.Hi * (unsigned)g.Wi * ldb, row_b = (unsigned)g.Wi * ldb;
  const char* bl = smem + l31 * kRsRow + 48 * half;   // this lane's B fragments: + ((nt * NCH + ch) * 9 + tap) * 32 * 112 + 16 * piece

  f32x4 raw[3][NCH][2];
  f32x16 acc[3][NT][2];
#pragma unroll
  for (int s = 0; s < 3; ++s)
#pragma unroll
    for (int n = 0; n < NT; ++n)
#pragma unroll
      for (int r = 0; r < 16; ++r) { acc[s][n][0][r] = 0.f; acc[s][n][1][r] = 0.f; }

  // loads of input row index i_ (image row y0 - 1 + i_) into raw[kx_][ch_]; rows outside the image or past the strip read zeros
#define PIDM_RSF_LOAD(i_, kx_, ch_)                                                                                   \
  {                                                                                                                   \
    const int r__ = y0 - 1 + (i_);                                                                                    \
    const bool ok__ = (r__ >= 0) & (r__ < g.Hi) & ((i_) < R + 2);                                                     \
    const unsigned vo__ = ok__ ? voff[kx_] : kRsOob;                                                                  \
    const unsigned so__ = img_off + (unsigned)(ok__ ? r__ : 0) * row_b;                                               \
    raw[kx_][ch_][0] = pidm_buf_load_f32x4(rsc[ch_], vo__, so__);                                                     \
    raw[kx_][ch_][1] = pidm_buf_load_f32x4(rsc[ch_], vo__, so__ + 16u);                                               \
  }
#pragma unroll
  for (int kx = 0; kx < 3; ++kx)
#pragma unroll
    for (int ch = 0; ch < NCH; ++ch) PIDM_RSF_LOAD(0, kx, ch)
  PIDM_WAIT_VMEM();
  __syncthreads();                                   // the weights are in LDS (the only barrier)
  if (!live) return;

  // the finished output row o_ (slot s_): bias, GroupNorm sums, transposes, residual, stores; the slot restarts at zero
#define PIDM_RSF_EPILOGUE(s_, o_)                                                                                     \
  _Pragma("unroll") for (int nt = 0; nt < NT; ++nt) {                                                                 \
    f32x16 av = acc[s_][nt][0];                                                                                       \
    _Pragma("unroll") for (int r = 0; r < 16; ++r) av[r] += acc[s_][nt][1][r];                                       \
    const int n0 = (ng * NT + nt) * 32, c = n0 + l31;                                                                 \
    const float bv = bvs[nt];                                                                                         \
    const int pin = (y0 + (o_)) * g.Wv + x0;                                                                          \
    float v[16];                                                                                                      \
    float gs1 = 0.f, gs2 = 0.f;                                                                                       \
    _Pragma("unroll") for (int r = 0; r < 16; ++r) {                                                                  \
      v[r] = av[r] + bv;                                                                                              \
      gs1 += v[r];                                                                                                    \
      gs2 += v[r] * v[r];                                                                                             \
    }                                                                                                                 \
    if (g.gn_part) PIDM_GN_PARTIAL(gs1, gs2, b, pin, c)                                                               \
    if (g.bn_part) {                                                                                                  \
      const float* xrow = g.bn_x + ((size_t)b * g.Ho * g.Wo + pin) * g.Cout + c;                                      \
      PIDM_BN_PARTIAL(av, bv, b, pin, c, xrow, g.Cout,                                                                \
                      (g.bn_res && residual) ? residual + ((size_t)b * g.Ho * g.Wo + pin) * g.ldr + c : (const float*)nullptr, g.ldr) \
    }                                                                                                                 \
    const bool odd1 = (l31 & 1) != 0, odd2 = (l31 & 2) != 0;                                                          \
    const size_t opix = (size_t)b * g.sob + (size_t)pin * g.sox + n0 + 4 * (l31 >> 2);                                \
    const size_t rpix = ((size_t)b * g.Ho * g.Wo + pin) * g.ldr + n0 + 4 * (l31 >> 2);                                \
    _Pragma("unroll") for (int q4 = 0; q4 < 4; ++q4) {                                                                \
      float e0 = v[4 * q4], e1 = v[4 * q4 + 1], e2 = v[4 * q4 + 2], e3 = v[4 * q4 + 3];                               \
      const float r01 = pidm_quad_xor1(odd1 ? e0 : e1), r23 = pidm_quad_xor1(odd1 ? e2 : e3);                         \
      e0 = odd1 ? r01 : e0; e1 = odd1 ? e1 : r01;                                                                     \
      e2 = odd1 ? r23 : e2; e3 = odd1 ? e3 : r23;                                                                     \
      const float r02 = pidm_quad_xor2(odd2 ? e0 : e2), r13 = pidm_quad_xor2(odd2 ? e1 : e3);                         \
      e0 = odd2 ? r02 : e0; e2 = odd2 ? e2 : r02;                                                                     \
      e1 = odd2 ? r13 : e1; e3 = odd2 ? e3 : r13;                                                                     \
      const int prow = 8 * q4 + 4 * half + (l31 & 3);                                                                 \
      f32x4 o = {e0, e1, e2, e3};                                                                                     \
      if (residual) o += *reinterpret_cast<const f32x4*>(residual + rpix + (size_t)prow * g.ldr);                     \
      *reinterpret_cast<f32x4*>(out + opix + (size_t)prow * g.sox) = o;                                               \
    }                                                                                                                 \
    _Pragma("unroll") for (int r = 0; r < 16; ++r) { acc[s_][nt][0][r] = 0.f; acc[s_][nt][1][r] = 0.f; }             \
  }

  // One input row (index i_, i_ % 3 == J_): groups of six MFMAs in the order (kx, chunk, ky, n-tile); the output row of tap row ky
  // is o = i_ - ky in slot (J_ + 3 - ky) % 3.  KM_ = the tap rows whose output row exists (bit ky; compile time): 1 and 3 for the
  // first two rows of a strip, 7 in between, 6 and 4 for the last two.  The pipeline runs across groups, steps and rows: a group
  // reads the weight fragments of the NEXT group before its own MFMAs (NKY_ = tap row of the next row's first group), the groups of
  // a (kx, chunk) step split the registers of the next step into its pieces - which are then re-loaded with the next input row -
  // and a scheduling fence closes every group, so that the compiler neither hoists a row's worth of fragment reads nor sinks them.
  // No branch inside a row: the matrix instructions of a group and the vector work around them share a scheduling region.
#define PIDM_RSF_FRAGS(dst_, bl_, kx_, ch_, ky_, nt_)                                                                 \
  {                                                                                                                   \
    const u32x4* bp__ = reinterpret_cast<const u32x4*>((bl_) + (((nt_) * NCH + (ch_)) * 9 + (ky_) * 3 + (kx_)) * (32 * kRsRow)); \
    dst_[0] = bp__[0]; dst_[1] = bp__[1]; dst_[2] = bp__[2];                                                          \
  }
#define PIDM_RSF_ROW(J_, i_, KM_, NKY_)                                                                               \
  {                                                                                                                   \
    constexpr int NK__ = ((KM_) & 1) + (((KM_) >> 1) & 1) + (((KM_) >> 2) & 1);                                       \
    constexpr int KF__ = ((KM_) & 1) ? 0 : (((KM_) & 2) ? 1 : 2);                      /* first tap row of the mask */  \
    _Pragma("unroll") for (int kx = 0; kx < 3; ++kx) {                                                                \
      _Pragma("unroll") for (int ch = 0; ch < NCH; ++ch) {                                                            \
        const int sn = (kx * NCH + ch + 1) % (3 * NCH), kxn = sn / NCH, chn = sn % NCH;   /* next step (of the next row after the last) */ \
        const f32x4 a0 = raw[kxn][chn][0], a1 = raw[kxn][chn][1];                                                     \
        unsigned q0[4], q1[4], q2[4];                                                                                 \
        int z__ = 0;                       /* the fragment reads are loop-invariant: keep them where they are written */ \
        PIDM_OPAQUE_I32(z__);                                                                                         \
        const char* blz = bl + z__;                                                                                   \
        _Pragma("unroll") for (int ky = 0; ky < 3; ++ky) {                                                            \
          if (((KM_) >> ky) & 1) {                                                                                    \
            const int kidx = ((KM_) & ((1 << ky) - 1) & 1) + (((KM_) & ((1 << ky) - 1)) >> 1);                        \
            const int kyn = (ky < 1 && ((KM_) & 2)) ? 1 : ((ky < 2 && ((KM_) & 4)) ? 2 : -1);    /* next tap row of the mask */ \
            _Pragma("unroll") for (int nt = 0; nt < NT; ++nt) {                                                       \
              const int gi = ((kx * NCH + ch) * NK__ + kidx) * NT + nt;                                               \
              if (nt + 1 < NT) PIDM_RSF_FRAGS(fb[(gi + 1) & 1], blz, kx, ch, ky, nt + 1)                              \
              else if (kyn >= 0) PIDM_RSF_FRAGS(fb[(gi + 1) & 1], blz, kx, ch, kyn, 0)                                \
              else if (sn != 0) PIDM_RSF_FRAGS(fb[(gi + 1) & 1], blz, kxn, chn, KF__, 0)                              \
              else PIDM_RSF_FRAGS(fb[(gi + 1) & 1], blz, 0, 0, (NKY_), 0)                                             \
              PIDM_RSF_MFMA6(acc[((J_) + 3 - ky) % 3][nt][(ky * 3 + kx) & 1], pc, fb[gi & 1])                         \
              if (nt == 0) {                                                                                          \
                if (kidx == 0) {                                                                                      \
                  pidm_split3_pk(a0[0], a0[1], q0[0], q1[0], q2[0]);                                                  \
                  pidm_split3_pk(a0[2], a0[3], q0[1], q1[1], q2[1]);                                                  \
                }                                                                                                     \
                if (kidx == (NK__ > 1 ? 1 : 0)) pidm_split3_pk(a1[0], a1[1], q0[2], q1[2], q2[2]);                    \
                if (kidx == NK__ - 1) pidm_split3_pk(a1[2], a1[3], q0[3], q1[3], q2[3]);                              \
              }                                                                                                       \
              if (kidx == NK__ - 1 && nt == NT - 1) {                                                                 \
                pc[0] = u32x4{q0[0], q0[1], q0[2], q0[3]};                                                            \
                pc[1] = u32x4{q1[0], q1[1], q1[2], q1[3]};                                                            \
                pc[2] = u32x4{q2[0], q2[1], q2[2], q2[3]};                                                            \
                PIDM_RSF_LOAD((i_) + (sn == 0 ? 2 : 1), kxn, chn)                                                     \
              }                                                                                                       \
              __builtin_amdgcn_sched_barrier(0);                                                                      \
            }                                                                                                         \
          }                                                                                                           \
        }                                                                                                             \
      }                                                                                                               \
    }                                                                                                                 \
    if ((i_) >= 2) PIDM_RSF_EPILOGUE(((J_) + 1) % 3, (i_) - 2)                                                        \
  }

  // pipeline prologue: pieces of the first step of the first row (its registers go on to the second row), fragments of the first group
  u32x4 pc[3], fb[2][3];
  {
    const f32x4 a0 = raw[0][0][0], a1 = raw[0][0][1];
    unsigned q0[4], q1[4], q2[4];
    pidm_split3_pk(a0[0], a0[1], q0[0], q1[0], q2[0]);
    pidm_split3_pk(a0[2], a0[3], q0[1], q1[1], q2[1]);
    pidm_split3_pk(a1[0], a1[1], q0[2], q1[2], q2[2]);
    pidm_split3_pk(a1[2], a1[3], q0[3], q1[3], q2[3]);
    pc[0] = u32x4{q0[0], q0[1], q0[2], q0[3]};
    pc[1] = u32x4{q1[0], q1[1], q1[2], q1[3]};
    pc[2] = u32x4{q2[0], q2[1], q2[2], q2[3]};
    PIDM_RSF_LOAD(1, 0, 0)
    PIDM_RSF_FRAGS(fb[0], bl, 0, 0, 0, 0)
  }
  // rows 0 and 1 feed one and two output rows, rows 2 .. R - 1 three, rows R and R + 1 two and one (R >= 4, R % 3 == RM)
  PIDM_RSF_ROW(0, 0, 1, 0)
  PIDM_RSF_ROW(1, 1, 3, 0)
  int i0 = 2;
  for (; i0 + 2 < R; i0 += 3) {
    PIDM_RSF_ROW(2, i0, 7, 0)
    PIDM_RSF_ROW(0, i0 + 1, 7, 0)
    PIDM_RSF_ROW(1, i0 + 2, 7, (i0 + 3 < R ? 0 : 1))
  }
  if (RM == 1) {                     // (R - 2) % 3 == 2 rows left
    PIDM_RSF_ROW(2, R - 2, 7, 0)
    PIDM_RSF_ROW(0, R - 1, 7, 1)
    PIDM_RSF_ROW(1, R, 6, 2)
    PIDM_RSF_ROW(2, R + 1, 4, 0)
  } else {
    PIDM_RSF_ROW(2, R, 6, 2)
    PIDM_RSF_ROW(0, R + 1, 4, 0)
  }
#undef PIDM_RSF_FRAGS
#undef PIDM_RSF_ROW
#undef PIDM_RSF_EPILOGUE
#undef PIDM_RSF_LOAD
}

// PIDM_CONV_RS=0: off (conv3x3_split_kernel takes the launch).  PIDM_CONV_RS_WAVES: waves a launch should have at least before rows
// per strip are doubled (default 1024 = one per SIMD of an MI355X).  PIDM_CONV_RS_WPS = 2: the two-waves-per-SIMD register budget.
static int rs_fwd_knob(const char* name, int dflt) {
  const char* e = knob(name);
  return e ? atoi(e) : dflt;
}

// 0: launched; 1: not this kernel's shape (the caller goes on); < 0: error
int launch_conv_rs(const ConvGeom& g, const float* src0, const float* src1, const unsigned short* wsplit, const float* bias,
                   const float* residual, float* out, hipStream_t st) {
  if (!rs_fwd_knob("PIDM_CONV_RS", 1)) return 1;
  if (!(g.KH == 3 && g.KW == 3 && g.stride == 1 && g.nz == 1 && g.nph == 1 && g.os == 1 && g.pad_y[0] == 1 && g.pad_x[0] == 1)) return 1;
  if (!(g.Wv == g.Wi && g.Hv == g.Hi && g.Ho == g.Hv && g.Wo == g.Wv && (g.Wv % 32) == 0 && g.Wv >= 32)) return 1;
  if (!((g.Cin == 32 || g.Cin == 64) && (g.C0 % 16) == 0 && (g.Cout % 32) == 0 && g.soc == 1)) return 1;
  if (!(g.C1 == 0 || (g.ld1 == g.ld0 && src1))) return 1;
  if (!((g.ld0 & 3) == 0 && (g.sox & 3) == 0 && g.soy == (long)g.Wo * g.sox && g.sob == (long)g.Ho * g.Wo * g.sox)) return 1;
  const double bytes = (double)g.B * g.Hi * g.Wi * g.ld0 * 4.0;
  if (bytes >= 2147483648.0) return 1;               // 32-bit offsets, and kRsOob must stay out of range
  const int NCH = g.Cin / 16, ntn = g.Cout / 32;
  const int NT = (NCH == 2 && (ntn % 2) == 0) ? 2 : 1;
  const int ngr = ntn / NT;
  const int want = rs_fwd_knob("PIDM_CONV_RS_WAVES", 1024);
  int R = g.Hv;
  if (R < 4 || (R & (R - 1))) return 1;             // (a power of two: R % 3 is 1 or 2)
  while (R > 4 && (long)g.B * (g.Wv / 32) * (g.Hv / R) * ngr < want) R >>= 1;
  // too little work for strips of 8 rows: conv3x3_split_kernel's 256-pixel tiles fill the chip better (measured at batch 64: the
  // 32x32 level 20-30 us there, 28-33 us here with strips of 4 rows, where every row is an edge row)
  if (R < rs_fwd_knob("PIDM_CONV_RS_MINR", 8) && (long)g.B * (g.Wv / 32) * (g.Hv / R) * ngr < want) return 1;
  const int n_units = g.B * (g.Wv / 32) * (g.Hv / R);
  const size_t lds = (size_t)NT * NCH * kRsSlab;
  const int wps = rs_fwd_knob("PIDM_CONV_RS_WPS", 1) == 2 ? 2 : 1;
  if (knob("PIDM_TRACE_CONV"))
    fprintf(stderr, "[pidm]   -> conv3x3_rs_kernel<%d, %d, %d, %d>, %d strips of %d rows, %d n-groups, %zu B LDS\n", NCH, NT, wps, R % 3, n_units, R, ngr, lds);
  const bool prof = prof_enabled();
  if (prof) prof_begin_launch(2, 2.0 * g.B * g.Hv * g.Wv * (double)g.Cout * g.Cin * 9, st);
  const dim3 grid(cdiv(n_units, 4), ngr), block(256);
  const float* s1 = src1 ? src1 : src0;
  const unsigned sb = (unsigned)bytes;
#define PIDM_RSF_GO(a, b, c, d)                                                                                                   \
  {                                                                                                                               \
    static bool attr__ = false;                                                                                                   \
    if (!attr__) {                                                                                                                \
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3_rs_kernel<a, b, c, d>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 256); \
      attr__ = true;                                                                                                              \
    }                                                                                                                             \
    hipLaunchKernelGGL(HIP_KERNEL_NAME(conv3x3_rs_kernel<a, b, c, d>), grid, block, lds, st, g, src0, s1, wsplit, bias, residual, out, R, n_units, sb); \
  }
#define PIDM_RSF_GO_RM(a, b, c) if (R % 3 == 1) PIDM_RSF_GO(a, b, c, 1) else PIDM_RSF_GO(a, b, c, 2)
  if (NCH == 2 && NT == 2) PIDM_RSF_GO_RM(2, 2, 1)
  else if (NCH == 2 && wps == 2) PIDM_RSF_GO_RM(2, 1, 2)
  else if (NCH == 2) PIDM_RSF_GO_RM(2, 1, 1)
  else if (wps == 2) PIDM_RSF_GO_RM(4, 1, 2)
  else PIDM_RSF_GO_RM(4, 1, 1)
#undef PIDM_RSF_GO_RM
#undef PIDM_RSF_GO
  if (prof) prof_end_launch(st);
  PIDM_CHECK_LAUNCH("conv3x3_rs_kernel");
  return 0;
}

}  // namespace pidm
