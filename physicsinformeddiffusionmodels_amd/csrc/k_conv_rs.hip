// Row-streaming 3x3 / stride-1 convolution (forward and input gradient) for the wide levels of the UNet (rows of 32 or 64 pixels,
// 32 or 64 input channels): the split-form contraction of conv3x3_split_kernel (k_conv.hip: fp32 operands as three bf16 pieces, six
// v_mfma_f32_32x32x16_bf16 per fp32 product, small terms first, even / odd taps in two accumulator chains) with the ACTIVATIONS
// NEVER PASSING THROUGH LDS.  Replaces the `convolution` ATen op of `Block.proj` at the 64x64 and 32x32 levels
// (/root/reference/src/unet_model.py:227 via :236) and its input gradient.
//
// Why: at these levels a tile of conv3x3_split_kernel has only 2-4 stages of 16 input channels, so its per-stage staging (global ->
// registers -> split -> LDS -> barrier -> fragment reads: ~1950 of a stage's ~5800 cycles, profiles/r02_split_conv_notes.txt) and
// its per-tile epilogue with both waves of a SIMD lined up by the barrier (~2200 cycles every 2-4 stages) keep the matrix pipe at
// 0.32-0.45 occupancy by the PMC measure (profiles/r04_pmc_split_kernels.txt).  The operand fragment of the 32x32x16 MFMA is "lane = pixel, 8
// consecutive channels" - in a channels-last image that is 32 contiguous bytes of the lane's own pixel - so a wave can load its
// fragments straight from global memory, split them in registers and keep going: no staging buffer, no barrier, no tile.
//
// A wave owns a strip of 32 pixels x R output rows of one image and walks the R + 2 input rows top to bottom.  For input row r it
// loads the row three times (shifted by -1 / 0 / +1 pixels: the kx taps; the re-reads hit L1), splits each 16-channel chunk into its
// pieces and issues, for ky = 0..2, the six MFMAs of tap (ky, kx) into the accumulators of output row r + 1 - ky: three output rows
// are live (x 2 chains x NT n-tiles x 16 registers), the one that received its last row (ky = 2) is finished.
// Zero padding comes from the buffer descriptor: out-of-image columns and rows use an out-of-range offset (reads as 0 without a memory
// access, pidm_common.h).  The pre-split weights of the workgroup's n-tiles (the packing of conv3x3_split_kernel: an image of 112-byte
// LDS rows) are copied to LDS once per workgroup by global_load_lds; a wave reads each tap's fragments once per input row.  The only
// barrier of the kernel publishes the weights.
// Epilogue of a finished row = the tile epilogue of conv3x3_split_kernel (bias, GroupNorm partial sums, GroupNorm-backward sums, 4x4
// register transposes, residual, 16-byte stores), but (i) cut into 8-13 pieces that ride between the matrix instructions of the NEXT
// input row - with one wave per SIMD nothing else can fill the gaps between a wave's MFMAs -, (ii) without a branch (absent operands =
// buffer descriptors of size 0), (iii) with the GroupNorm sums accumulated over the strip's rows and written once per strip
// (ConvGeom::part_chunks_out tells the caller how many chunks per image to total).
// Results: same pieces, same products, same two chains per output element as conv3x3_split_kernel, but the chains run tap-major
// instead of chunk-major - equal to it within fp32 rounding of the accumulation order, not bit for bit.
// Measured (PIDM_RS_TRACE, profiles/r04_m_conv_rs_clock.txt): 76-83 % of the matrix pipe busy in shader cycles; the shader clock under
// this kernel is 1.5-1.87 GHz.
#include <stdio.h>
#include <stdlib.h>
#include <type_traits>
#include <utility>

#include "pidm_launch.h"
#include "k_conv_epilogue.h"

namespace pidm {

#ifndef PIDM_RSF_ABLATE
#define PIDM_RSF_ABLATE 0   // measurement builds only (tools/rs_ablate.py; wrong results): 1 = no epilogue pieces, 2 = no split, 4 = no activation loads in the row loop, 8 = no fragment reads, 16 = no matrix instructions
#endif
// PIDM_RS_TRACE=1: shader-clock and 100 MHz real-time stamps of workgroup 0 / wave 0 around its row loop (launch_conv_rs prints the
// clock the kernel ran at and the cycles per row)
__device__ unsigned long long g_rs_trace[4];
static constexpr int kRsRow = 112;                  // bytes per LDS weight row (k_conv.hip: kSplitRow)
static constexpr int kRsSlab = 9 * 32 * kRsRow;     // pre-split weights of one (n-tile, 16-channel chunk)
static constexpr unsigned kRsOob = 0x80000000u;     // a byte offset no tensor reaches (the launcher checks): reads as 0

// order of the tap rows inside a (kx, chunk) step: ky = 1, 2, 0 - the slot of ky = 0 is the one the previous input row finished, and
// its sums are taken out during the first two groups; km = the tap rows present (bit ky)
__host__ __device__ constexpr int rs_nk(int km) { return (km & 1) + ((km >> 1) & 1) + ((km >> 2) & 1); }
__host__ __device__ constexpr int rs_ord(int km, int oi) {
  int n = 0;
  if (km & 2) { if (n == oi) return 1; ++n; }
  if (km & 4) { if (n == oi) return 2; ++n; }
  if (km & 1) { if (n == oi) return 0; ++n; }
  return 0;
}
// group slot (of nslot in the row, sps per (kx, chunk) step) in which epilogue piece k of np runs: the two halves of the take in the
// first step, before its ky = 0 group adds to the slot they read; the others evenly behind them
__host__ __device__ constexpr int rs_slot(int k, int np, int nslot, int sps) {
  if (k < 2) return k < sps ? k : sps - 1;
  const int first = sps < 2 ? sps : 2, s = first + (k - 2) * (nslot - first) / (np - 2);
  return s < nslot ? s : nslot - 1;
}
// compile-time loops: f(integral_constant<int, 0>) ... f(integral_constant<int, N - 1>) - every index a constant the front end folds
// (as unrolled loops the dead alternatives of every iteration reached the optimiser, the unroller gave up on the size and the
// accumulators ended up in scratch memory; as nested macros the translation unit ran out of source locations)
template <int N> using rs_ic = std::integral_constant<int, N>;
template <class F, int... I>
__device__ __forceinline__ void rs_for_impl(F& f, std::integer_sequence<int, I...>) { (f(rs_ic<I>{}), ...); }
template <int N, class F>
__device__ __forceinline__ void rs_for(F&& f) { rs_for_impl(f, std::make_integer_sequence<int, N>{}); }
// piece number of the k-th epilogue piece in running order: take, loads, GroupNorm sums, [sums of the backward], quarters
__host__ __device__ constexpr int rs_piece(int k, int bnp) {
  if (!bnp) return k;
  return k <= 2 ? k : (k == 3 ? 8 : (k == 4 ? 3 : (k <= 8 ? k + 4 : k - 5)));
}
#define PIDM_RSF_PIN(x_) asm volatile("" : "+v"(x_))   // the value exists here: computations on it neither start before nor end after

#define PIDM_RSF_MFMA6(acc_, a_, b_)                          \
  acc_ = pidm_mfma_bf16_32x32x16(a_[2], b_[0], acc_);         \
  acc_ = pidm_mfma_bf16_32x32x16(a_[0], b_[2], acc_);         \
  acc_ = pidm_mfma_bf16_32x32x16(a_[1], b_[1], acc_);         \
  acc_ = pidm_mfma_bf16_32x32x16(a_[1], b_[0], acc_);         \
  acc_ = pidm_mfma_bf16_32x32x16(a_[0], b_[1], acc_);         \
  acc_ = pidm_mfma_bf16_32x32x16(a_[0], b_[0], acc_);

// NCH = Cin / 16 (2 or 4), NT = n-tiles of 32 output channels per wave (1 or 2), RM = R % 3 (1 or 2: R is a power of two >= 4; it
// fixes which accumulator slot the last rows of a strip use), BNP = 1: the launch also leaves the GroupNorm-backward sums
// (ConvGeom::bn_part; ~200 vector instructions per output row and n-tile that the other launches do not carry).
// One wave per SIMD (512 registers): what overlaps the matrix instructions is this wave's own vector work, interleaved by the
// compiler inside each fenced group - so the row code has NO branch: optional operands (residual, GroupNorm partials) go through
// buffer descriptors of size 0 when absent (loads return 0, stores are dropped), lane predicates through out-of-range offsets.
template <int NCH, int NT, int RM, int BNP>
__global__ void __launch_bounds__(256) PIDM_WAVES_PER_SIMD(1)
conv3x3_rs_kernel(ConvGeom g, const float* __restrict__ src0, const float* __restrict__ src1, const unsigned short* __restrict__ ws,
                  const float* __restrict__ bias, const float* __restrict__ residual, float* __restrict__ out, int R, int n_units,
                  unsigned src_bytes, unsigned res_bytes, int pch, int trace) {
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
  // optional operands of the epilogue: size 0 when absent
  const pidm_rsrc rs_res = pidm_make_rsrc(residual, residual ? res_bytes : 0u);
  // partial sums: ONE chunk per strip (pch = strips per image; chunk = the unit's index inside its image), summed over the strip's
  // rows in double per lane - the consumers (k_norm.hip) total H*W/32 chunks per image behind the tile kernels, 8-32x fewer here
  const pidm_rsrc rs_gn = pidm_make_rsrc(g.gn_part, g.gn_part ? (unsigned)g.B * (unsigned)pch * (unsigned)g.gn_G * 16u : 0u);
  const pidm_rsrc rs_bn = pidm_make_rsrc(g.bn_part, (BNP && g.bn_part) ? (unsigned)g.B * (unsigned)pch * (unsigned)g.Cout * 16u : 0u);
  const int chunk = sx * nrb + rb;
  const pidm_rsrc rs_bnres = pidm_make_rsrc(residual, (BNP && g.bn_res && residual) ? res_bytes : 0u);
  const int HW = g.Ho * g.Wo;
  float bvs[NT];
  // per lane and n-tile: constants of the GroupNorm-backward sums (the wave's image and the lane's channel never change)
  float bn_mean[NT], bn_rstd[NT], bn_gm[NT], bn_bt[NT], bn_sc[NT], bn_sh[NT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) {
    const int c = (ng * NT + nt) * 32 + l31;
    bvs[nt] = bias ? bias[c] : 0.f;
    bn_mean[nt] = bn_rstd[nt] = bn_gm[nt] = bn_bt[nt] = 0.f;
    bn_sc[nt] = 1.f;
    bn_sh[nt] = 0.f;
    if (BNP && g.bn_part) {
      const int gI = c / g.bn_cpg;
      bn_mean[nt] = g.bn_stats[((size_t)b * g.bn_G + gI) * 2];
      bn_rstd[nt] = g.bn_stats[((size_t)b * g.bn_G + gI) * 2 + 1];
      bn_gm[nt] = g.bn_gamma[c];
      bn_bt[nt] = g.bn_beta[c];
      if (g.bn_ss) {
        bn_sc[nt] = 1.f + (g.bn_ss[(size_t)b * g.bn_ldss + c] + g.bn_ssb[c]);
        bn_sh[nt] = g.bn_ss[(size_t)b * g.bn_ldss + g.Cout + c] + g.bn_ssb[g.Cout + c];
      }
    }
  }
  // per-lane byte offset of the lane's 8 channels inside a row, per kx (out-of-image columns: out of range)
  unsigned voff[3];
#pragma unroll
  for (int kx = 0; kx < 3; ++kx) {
    const int x = x0 + l31 + kx - 1;
    voff[kx] = (live && x >= 0 && x < g.Wv) ? (unsigned)x * ldb + 32u * (unsigned)half : kRsOob;
  }
  const unsigned img_off = (unsigned)b * (unsigned)g.Hi * (unsigned)g.Wi * ldb, row_b = (unsigned)g.Wi * ldb;
  const char* bl = smem + l31 * kRsRow + 48 * half;   // this lane's B fragments: + ((nt * NCH + ch) * 9 + tap) * 32 * 112 + 16 * piece
  // epilogue addressing.  Accumulator register r of a lane is pixel (r & 3) + 8 (r >> 2) + 4 half of the wave's 32, channel l31;
  // after the 4x4 transposes a lane holds channels 4 (l31 >> 2) .. + 3 of pixel 8 q + 4 half + (l31 & 3), q = 0..3.
  const int tp = 4 * half + (l31 & 3);                                        // the lane's pixel inside a quarter after the transposes
  const unsigned lres = (unsigned)(tp * g.ldr + 4 * (l31 >> 2)) * 4u;          // its bytes inside the residual row block
  const size_t lout = (size_t)tp * g.sox + 4 * (l31 >> 2);
  const int gn_cpg = g.gn_part ? g.gn_cpg : 1;                                 // (no partials: any valid divisor)
  const unsigned gn_lane = (half == 0 && (l31 & (gn_cpg - 1)) == 0) ? 0u : kRsOob;   // the lane of a group that writes its sums
  const unsigned bn_lane = (half == 0) ? 0u : kRsOob;

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

  // ---- the epilogue of a finished output row, in pieces that ride inside the NEXT input row's groups -------------------------
  // pend = the row's sums (both chains + bias), taken out of the accumulators during the first groups of the next input row (the
  // slot is that row's ky = 0 slot, which the order 1, 2, 0 touches last).  Every piece pins its inputs and outputs
  // (PIDM_RSF_PIN) or ends in a store: unpinned arithmetic floats out of the group it is written in - into the fragment-read region,
  // where no matrix instruction covers it.
  float pend[NT][16];
  double gd[NT][2], bd[BNP ? NT : 1][2];     // GroupNorm / GroupNorm-backward sums of the strip so far
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) { gd[nt][0] = gd[nt][1] = 0.0; bd[BNP ? nt : 0][0] = bd[BNP ? nt : 0][1] = 0.0; }
  f32x4 rres[NT][4];                 // residual rows, loaded a few groups ahead of their use
  float bxv[BNP ? NT : 1][16], brv[BNP ? NT : 1][16], ba1[BNP ? NT : 1], ba2[BNP ? NT : 1];
  // pieces: 0 / 1 take the sums (halves), 2 residual loads, 3 GroupNorm sums of the row (added to the strip's), 4-7 the four
  // quarters (transposes, residual, stores); BNP: 8 loads of x (and of the residual as dy's second term), 9-12 the sums' quarters
  auto piece = [&](auto P_, auto S_, int o_) __attribute__((always_inline)) {
    constexpr int p = decltype(P_)::value, sl = decltype(S_)::value;
    const int pin = (y0 + o_) * g.Wv + x0;                          // first pixel of the row inside the image
    const unsigned pixb = (unsigned)b * (unsigned)HW + (unsigned)pin;
    if constexpr (p == 0 || p == 1) {
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
#pragma unroll
        for (int r = (p == 0 ? 0 : 8); r < (p == 0 ? 8 : 16); ++r) {
          pend[nt][r] = (acc[sl][nt][0][r] + acc[sl][nt][1][r]) + bvs[nt];
          PIDM_RSF_PIN(pend[nt][r]);
          acc[sl][nt][0][r] = 0.f;
          acc[sl][nt][1][r] = 0.f;
        }
      }
    } else if constexpr (p == 2) {
#pragma unroll
      for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int q4 = 0; q4 < 4; ++q4)
          rres[nt][q4] = pidm_buf_load_f32x4(rs_res, lres, ((pixb + 8u * q4) * (unsigned)g.ldr + (unsigned)((ng * NT + nt) * 32)) * 4u);
    } else if constexpr (p == 3) {
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        PIDM_RSF_PIN(pend[nt][0]);
        float a1 = pend[nt][0], a2 = pend[nt][0] * pend[nt][0];
#pragma unroll
        for (int r = 1; r < 16; ++r) { a1 += pend[nt][r]; a2 += pend[nt][r] * pend[nt][r]; }
        gd[nt][0] += (double)a1;
        gd[nt][1] += (double)a2;
        PIDM_RSF_PIN(gd[nt][0]);
        PIDM_RSF_PIN(gd[nt][1]);
      }
    } else if constexpr (p >= 4 && p <= 7) {
      constexpr int q4 = p - 4;
      const bool odd1 = (l31 & 1) != 0, odd2 = (l31 & 2) != 0;
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        float e0 = pend[nt][4 * q4], e1 = pend[nt][4 * q4 + 1], e2 = pend[nt][4 * q4 + 2], e3 = pend[nt][4 * q4 + 3];
        PIDM_RSF_PIN(e0); PIDM_RSF_PIN(e1); PIDM_RSF_PIN(e2); PIDM_RSF_PIN(e3);
        const float r01 = pidm_quad_xor1(odd1 ? e0 : e1), r23 = pidm_quad_xor1(odd1 ? e2 : e3);
        e0 = odd1 ? r01 : e0; e1 = odd1 ? e1 : r01;
        e2 = odd1 ? r23 : e2; e3 = odd1 ? e3 : r23;
        const float r02 = pidm_quad_xor2(odd2 ? e0 : e2), r13 = pidm_quad_xor2(odd2 ? e1 : e3);
        e0 = odd2 ? r02 : e0; e2 = odd2 ? e2 : r02;
        e1 = odd2 ? r13 : e1; e3 = odd2 ? e3 : r13;
        f32x4 o = {e0, e1, e2, e3};
        o += rres[nt][q4];
        *reinterpret_cast<f32x4*>(out + (size_t)b * g.sob + (size_t)(pin + 8 * q4) * g.sox + (ng * NT + nt) * 32 + lout) = o;
      }
    } else if constexpr (BNP && p == 8) {
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        const int c = (ng * NT + nt) * 32 + l31;
        const float* xrow = g.bn_x + ((size_t)pixb) * g.Cout + c;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int prow = (r & 3) + 8 * (r >> 2) + 4 * half;
          bxv[nt][r] = xrow[(size_t)prow * g.Cout];
          brv[nt][r] = pidm_buf_load_f32(rs_bnres, (unsigned)(prow * g.ldr + l31) * 4u, (pixb * (unsigned)g.ldr + (unsigned)((ng * NT + nt) * 32)) * 4u);
        }
        ba1[nt] = 0.f;
        ba2[nt] = 0.f;
      }
    } else if constexpr (BNP && p >= 9 && p <= 12) {
      constexpr int k4 = p - 9;
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        PIDM_RSF_PIN(ba1[nt]);
        PIDM_RSF_PIN(ba2[nt]);
#pragma unroll
        for (int r = 4 * k4; r < 4 * k4 + 4; ++r) {
          const float xh = (bxv[nt][r] - bn_mean[nt]) * bn_rstd[nt];
          const float v = (xh * bn_gm[nt] + bn_bt[nt]) * bn_sc[nt] + bn_sh[nt];
          const float sg = pidm_sigmoid(v);
          const float dv = (pend[nt][r] + brv[nt][r]) * (sg * (1.f + v * (1.f - sg)));
          ba1[nt] += dv;
          ba2[nt] += dv * xh;
        }
        PIDM_RSF_PIN(ba1[nt]);
        PIDM_RSF_PIN(ba2[nt]);
        if constexpr (k4 == 3) {
          bd[nt][0] += (double)ba1[nt];
          bd[nt][1] += (double)ba2[nt];
          PIDM_RSF_PIN(bd[nt][0]);
          PIDM_RSF_PIN(bd[nt][1]);
        }
      }
    }
  };
  constexpr int NP = BNP ? 13 : 8;

  // One input row (index i, i % 3 == J): groups of six MFMAs in the order (kx, chunk, ky in the order 1, 2, 0, n-tile); the
  // output row of tap row ky is o = i - ky in slot (J + 3 - ky) % 3.  KM = the tap rows whose output row exists (bit ky): 1 and 3
  // for the first two rows of a strip, 7 in between, 6 and 4 for the last two.  PEND = 1: output row i - 3 - slot J, finished by
  // the previous input row - is waiting: its epilogue pieces are spread over this row's groups (all but the last of every
  // (kx, chunk) step, which carries the split).
  // The pipeline runs across groups, steps and rows.  A group = [reads of the NEXT group's weight fragments (NKY = tap row of the
  // next row's first group)] fence [its six MFMAs + its share of vector work, which the compiler interleaves] fence; the last
  // group of a step splits the registers of the next step into its pieces and re-loads them with the next input row.
#define PIDM_RSF_FRAGS(dst_, bl_, kx_, ch_, ky_, nt_)                                                                 \
  {                                                                                                                   \
    const u32x4* bp__ = reinterpret_cast<const u32x4*>((bl_) + (((nt_) * NCH + (ch_)) * 9 + (ky_) * 3 + (kx_)) * (32 * kRsRow)); \
    dst_[0] = bp__[0]; dst_[1] = bp__[1]; dst_[2] = bp__[2];                                                          \
  }
  u32x4 pc[3], fb[2][3];
  auto row = [&](auto J_, auto KM_, auto NKY_, auto PEND_, int i) __attribute__((always_inline)) {
    constexpr int J = decltype(J_)::value, KM = decltype(KM_)::value, NKY = decltype(NKY_)::value, PEND = decltype(PEND_)::value;
    constexpr int NK = rs_nk(KM), GPS = NK * NT;                          // tap rows present, groups per step
    constexpr int SPS = GPS > 1 ? GPS - 1 : 1, NSLOT = 3 * NCH * SPS;     // epilogue slots per step / row
    rs_for<3 * NCH>([&](auto ST_) __attribute__((always_inline)) {
      constexpr int st = decltype(ST_)::value, kx = st / NCH, ch = st % NCH;
      constexpr int sn = (st + 1) % (3 * NCH), kxn = sn / NCH, chn = sn % NCH;   // next step (of the next row after the last)
      int z = 0;                         // the fragment reads are loop-invariant: keep them where they are written
      PIDM_OPAQUE_I32(z);
      const char* blz = bl + z;
      rs_for<GPS>([&](auto GS_) __attribute__((always_inline)) {
        constexpr int gs = decltype(GS_)::value, oi = gs / NT, nt = gs % NT, ky = rs_ord(KM, oi);
        constexpr int gi = st * GPS + gs;                                   // group inside the row
        if constexpr ((PIDM_RSF_ABLATE & 8) != 0) { }
        else if constexpr (nt + 1 < NT) PIDM_RSF_FRAGS(fb[(gi + 1) & 1], blz, kx, ch, ky, nt + 1)
        else if constexpr (oi + 1 < NK) PIDM_RSF_FRAGS(fb[(gi + 1) & 1], blz, kx, ch, rs_ord(KM, oi + 1), 0)
        else if constexpr (sn != 0) PIDM_RSF_FRAGS(fb[(gi + 1) & 1], blz, kxn, chn, rs_ord(KM, 0), 0)
        else PIDM_RSF_FRAGS(fb[(gi + 1) & 1], blz, 0, 0, NKY, 0)
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (!(PIDM_RSF_ABLATE & 16)) PIDM_RSF_MFMA6(acc[(J + 3 - ky) % 3][nt][(ky * 3 + kx) & 1], pc, fb[gi & 1])
        if constexpr (PEND && (GPS == 1 || gs < GPS - 1) && !(PIDM_RSF_ABLATE & 1)) {
          constexpr int slot = st * SPS + (GPS == 1 ? 0 : gs);
          rs_for<NP>([&](auto K_) __attribute__((always_inline)) {
            constexpr int k = decltype(K_)::value;
            if constexpr (rs_slot(k, NP, NSLOT, SPS) == slot) piece(rs_ic<rs_piece(k, BNP)>{}, rs_ic<J>{}, i - 3);
          });
        }
        if constexpr (gs == GPS - 1 && (PIDM_RSF_ABLATE & 2)) {
          if (!(PIDM_RSF_ABLATE & 4)) PIDM_RSF_LOAD(i + (sn == 0 ? 2 : 1), kxn, chn)
        }
        if constexpr (gs == GPS - 1 && !(PIDM_RSF_ABLATE & 2)) {
          float e[8] = {raw[kxn][chn][0][0], raw[kxn][chn][0][1], raw[kxn][chn][0][2], raw[kxn][chn][0][3],
                        raw[kxn][chn][1][0], raw[kxn][chn][1][1], raw[kxn][chn][1][2], raw[kxn][chn][1][3]};
#pragma unroll
          for (int k = 0; k < 8; ++k) PIDM_RSF_PIN(e[k]);
          unsigned q0[4], q1[4], q2[4];
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            pidm_split3_pk(e[2 * k], e[2 * k + 1], q0[k], q1[k], q2[k]);
            PIDM_RSF_PIN(q0[k]); PIDM_RSF_PIN(q1[k]); PIDM_RSF_PIN(q2[k]);
          }
          pc[0] = u32x4{q0[0], q0[1], q0[2], q0[3]};
          pc[1] = u32x4{q1[0], q1[1], q1[2], q1[3]};
          pc[2] = u32x4{q2[0], q2[1], q2[2], q2[3]};
          if (!(PIDM_RSF_ABLATE & 4)) PIDM_RSF_LOAD(i + (sn == 0 ? 2 : 1), kxn, chn)
        }
        __builtin_amdgcn_sched_barrier(0);
      });
    });
  };
#define PIDM_RSF_ROW(J_, i_, KM_, NKY_, PEND_) row(rs_ic<J_>{}, rs_ic<KM_>{}, rs_ic<NKY_>{}, rs_ic<PEND_>{}, (i_));
#define PIDM_RSF_FLUSH(s_) rs_for<NP>([&](auto K_) __attribute__((always_inline)) { piece(rs_ic<rs_piece(decltype(K_)::value, BNP)>{}, rs_ic<s_>{}, R - 1); });

  // pipeline prologue: pieces of the first step of the first row (its registers go on to the second row), fragments of the first group
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
    if (PIDM_RSF_ABLATE & 8) PIDM_RSF_FRAGS(fb[1], bl, 0, 0, 1, 0)
  }
  const bool tr = trace && blockIdx.x == 0 && blockIdx.y == 0 && tid == 0;
  if (tr) { g_rs_trace[0] = __builtin_readcyclecounter(); g_rs_trace[1] = __builtin_amdgcn_s_memrealtime(); }
  // rows 0 and 1 feed one and two output rows, rows 2 .. R - 1 three, rows R and R + 1 two and one (R >= 4, R % 3 == RM); the
  // first finished row appears after row 2, so rows 3 ... carry a pending epilogue and the last one is flushed behind the loop
  PIDM_RSF_ROW(0, 0, 1, 1, 0)
  PIDM_RSF_ROW(1, 1, 3, 1, 0)
  PIDM_RSF_ROW(2, 2, 7, 1, 0)
  int i0 = 3;
  for (; i0 + 2 < R; i0 += 3) {
    PIDM_RSF_ROW(0, i0, 7, 1, 1)
    PIDM_RSF_ROW(1, i0 + 1, 7, 1, 1)
    PIDM_RSF_ROW(2, i0 + 2, 7, 1, 1)
  }
  // (RM is a template parameter because a run-time choice between the two tails - both reading every accumulator - makes the
  // register allocator copy accumulators around the diamond and spill: 801 registers for <2, 2>)
  if (RM == 1) {                     // (R - 3) % 3 == 1 row left; the last output row R - 1 sits in slot 0
    PIDM_RSF_ROW(0, R - 1, 7, 1, 1)
    PIDM_RSF_ROW(1, R, 6, 2, 1)
    PIDM_RSF_ROW(2, R + 1, 4, 0, 1)
    PIDM_RSF_FLUSH(0)
  } else {                           // two; slot 1
    PIDM_RSF_ROW(0, R - 2, 7, 1, 1)
    PIDM_RSF_ROW(1, R - 1, 7, 1, 1)
    PIDM_RSF_ROW(2, R, 6, 2, 1)
    PIDM_RSF_ROW(0, R + 1, 4, 0, 1)
    PIDM_RSF_FLUSH(1)
  }
  // the strip's partial sums: lanes of a group / the two halves added in a fixed order (64-bit values as two DPP / swap moves), one
  // 16-byte store per (group leader | channel) - dropped through the size-0 descriptor when the launch has no such epilogue
  {
    auto shuf = [&](double v, auto f) __attribute__((always_inline)) {
      const unsigned long long w = __builtin_bit_cast(unsigned long long, v);
      const unsigned lo = __builtin_bit_cast(unsigned, f(__builtin_bit_cast(float, (unsigned)w)));
      const unsigned hi = __builtin_bit_cast(unsigned, f(__builtin_bit_cast(float, (unsigned)(w >> 32))));
      return __builtin_bit_cast(double, ((unsigned long long)hi << 32) | lo);
    };
    auto x1 = [](float v) { return pidm_quad_xor1(v); };
    auto x2 = [](float v) { return pidm_quad_xor2(v); };
    auto s4 = [](float v) { return pidm_row_shl4(v); };
    auto s8 = [](float v) { return pidm_row_shl8(v); };
    auto oh = [](float v) { return pidm_other_half(v); };
    auto x16 = [](float v) { return __shfl_xor(v, 16); };   // the other row of 16 of the lane's half (groups of 32 channels only)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      const int c = (ng * NT + nt) * 32 + l31;
      double a1 = gd[nt][0], a2 = gd[nt][1];
      a1 += shuf(a1, x1); a2 += shuf(a2, x1);
      a1 += shuf(a1, x2); a2 += shuf(a2, x2);
      { const double t1 = shuf(a1, s4), t2 = shuf(a2, s4); a1 += (gn_cpg > 4) ? t1 : 0.0; a2 += (gn_cpg > 4) ? t2 : 0.0; }
      { const double t1 = shuf(a1, s8), t2 = shuf(a2, s8); a1 += (gn_cpg > 8) ? t1 : 0.0; a2 += (gn_cpg > 8) ? t2 : 0.0; }
      // a DPP row is 16 lanes: a group of 32 channels spans two rows, whose leaders (lanes 0 and 16 of the half) are added here
      if (gn_cpg > 16) { const double t1 = shuf(a1, x16), t2 = shuf(a2, x16); a1 += t1; a2 += t2; }
      a1 += shuf(a1, oh);
      a2 += shuf(a2, oh);
      const unsigned long long w1 = __builtin_bit_cast(unsigned long long, a1), w2 = __builtin_bit_cast(unsigned long long, a2);
      const unsigned go = (((unsigned)b * (unsigned)pch + (unsigned)chunk) * (unsigned)g.gn_G + (unsigned)(c / gn_cpg)) * 16u;
      pidm_buf_store_u32x4(rs_gn, gn_lane + go, 0u, u32x4{(unsigned)w1, (unsigned)(w1 >> 32), (unsigned)w2, (unsigned)(w2 >> 32)});
      if constexpr (BNP != 0) {
        double e1 = bd[nt][0], e2 = bd[nt][1];
        e1 += shuf(e1, oh);
        e2 += shuf(e2, oh);
        const unsigned long long v1 = __builtin_bit_cast(unsigned long long, e1), v2 = __builtin_bit_cast(unsigned long long, e2);
        const unsigned bo = (((unsigned)b * (unsigned)pch + (unsigned)chunk) * (unsigned)g.Cout + (unsigned)c) * 16u;
        pidm_buf_store_u32x4(rs_bn, bn_lane + bo, 0u, u32x4{(unsigned)v1, (unsigned)(v1 >> 32), (unsigned)v2, (unsigned)(v2 >> 32)});
      }
    }
  }
  if (tr) { g_rs_trace[2] = __builtin_readcyclecounter(); g_rs_trace[3] = __builtin_amdgcn_s_memrealtime(); }
#undef PIDM_RSF_FRAGS
#undef PIDM_RSF_FLUSH
#undef PIDM_RSF_ROW
#undef PIDM_RSF_LOAD
}

// PIDM_CONV_RS=0: off (conv3x3_split_kernel takes the launch).  PIDM_CONV_RS_WAVES: waves a launch should have at least before rows
// per strip are doubled (default 1024 = one per SIMD of an MI355X).  PIDM_CONV_RS_MINR: fewest rows per strip (8).
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
  const int bnp = g.bn_part ? 1 : 0;
  // two n-tiles per wave where the weights fit in LDS and the registers hold (the GroupNorm-backward sums need 64 more per n-tile)
  const int NT = (NCH == 2 && (ntn % 2) == 0 && !bnp) ? 2 : 1;
  const int ngr = ntn / NT;
  // One wave per SIMD.  Two (the <2, 1, *, 0> kernel fits 243 registers) ran at the same speed, 64x64 32 -> 32 at batch 256: 100 vs 103
  // us - the chip lowers its clock to fit its power budget (PIDM_RS_TRACE: 1.47-1.52 GHz in this kernel at batch 256, 1.70-1.78 at
  // batch 64), and what the second wave wins in issue slots the clock takes back (profiles/r04_m_conv_rs_clock.txt)
  const int want = rs_fwd_knob("PIDM_CONV_RS_WAVES", 1024);
  int R = g.Hv;
  if (R < 4 || (R & (R - 1))) return 1;             // (a power of two: R % 3 is 1 or 2)
  while (R > 4 && (long)g.B * (g.Wv / 32) * (g.Hv / R) * ngr < want) R >>= 1;
  // too little work for strips of 8 rows: conv3x3_split_kernel's 256-pixel tiles fill the chip better (measured at batch 64: the
  // 32x32 level 20-30 us there, 28-33 us here with strips of 4 rows, where every row is an edge row)
  if (R < rs_fwd_knob("PIDM_CONV_RS_MINR", 8) && (long)g.B * (g.Wv / 32) * (g.Hv / R) * ngr < want) return 1;
  const int n_units = g.B * (g.Wv / 32) * (g.Hv / R);
  const int pch = (g.Wv / 32) * (g.Hv / R);           // partial chunks per image: one per strip
  if ((g.gn_part || g.bn_part) && !g.part_chunks_out) return 1;   // the caller counts on Ho*Wo/32 chunks: a tile kernel's layout
  const size_t lds = (size_t)NT * NCH * kRsSlab;
  const double rbytes = (double)g.B * g.Ho * g.Wo * g.ldr * 4.0;
  if (residual && rbytes >= 2147483648.0) return 1;
  if (g.gn_part && (g.gn_cpg < 1 || g.gn_cpg > 32 || (g.gn_cpg & (g.gn_cpg - 1)))) return 1;
  // (only now - behind every `return 1` - does the caller learn this kernel's chunk count: a tile kernel that takes the launch
  // after a fall-through writes Ho*Wo/32 chunks per image)
  if (g.gn_part || g.bn_part) *g.part_chunks_out = pch;
  if (knob("PIDM_TRACE_CONV"))
    fprintf(stderr, "[pidm]   -> conv3x3_rs_kernel<%d, %d, %d, %d>, %d strips of %d rows, %d n-groups, %zu B LDS\n", NCH, NT, R % 3, bnp, n_units, R, ngr, lds);
  const bool prof = prof_enabled();
  if (prof) prof_begin_launch(2, 2.0 * g.B * g.Hv * g.Wv * (double)g.Cout * g.Cin * 9, st);
  const dim3 grid(cdiv(n_units, 4), ngr), block(256);
  const float* s1 = src1 ? src1 : src0;
  const unsigned sb = (unsigned)bytes;
  const unsigned rb = residual ? (unsigned)rbytes : 0u;
  // (the read-back below synchronises the stream: never while it is being captured into a hipGraph)
  int trace = knob("PIDM_RS_TRACE") ? 1 : 0;
  if (trace) {
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(st, &cs) != hipSuccess || cs != hipStreamCaptureStatusNone) trace = 0;
  }
#define PIDM_RSF_GO(a, b, c, d)                                                                                                   \
  {                                                                                                                               \
    static bool attr__ = false;                                                                                                   \
    if (!attr__) {                                                                                                                \
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3_rs_kernel<a, b, c, d>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 256); \
      attr__ = true;                                                                                                              \
    }                                                                                                                             \
    hipLaunchKernelGGL(HIP_KERNEL_NAME(conv3x3_rs_kernel<a, b, c, d>), grid, block, lds, st, g, src0, s1, wsplit, bias, residual, out, R, n_units, sb, rb, pch, trace); \
  }
#define PIDM_RSF_GO_RM(a, b, d) if (R % 3 == 1) PIDM_RSF_GO(a, b, 1, d) else PIDM_RSF_GO(a, b, 2, d)
  if (NCH == 2 && NT == 2) PIDM_RSF_GO_RM(2, 2, 0)
  else if (NCH == 2 && bnp) PIDM_RSF_GO_RM(2, 1, 1)
  else if (NCH == 2) PIDM_RSF_GO_RM(2, 1, 0)
  else if (bnp) PIDM_RSF_GO_RM(4, 1, 1)
  else PIDM_RSF_GO_RM(4, 1, 0)
#undef PIDM_RSF_GO_RM
#undef PIDM_RSF_GO
  if (prof) prof_end_launch(st);
  PIDM_CHECK_LAUNCH("conv3x3_rs_kernel");
  if (trace) {
    unsigned long long t[4] = {0, 0, 0, 0};
    if (hipStreamSynchronize(st) == hipSuccess && hipMemcpyFromSymbol(t, HIP_SYMBOL(g_rs_trace), sizeof(t)) == hipSuccess && t[3] > t[1])
      fprintf(stderr, "[pidm] conv3x3_rs_kernel<%d, %d, %d, %d> R=%d: row loop %.2f us, shader clock %.3f GHz, %.0f cycles per input row\n", NCH, NT, R % 3, bnp,
              R, (double)(t[3] - t[1]) * 0.01, (double)(t[2] - t[0]) / ((double)(t[3] - t[1]) * 10.0), (double)(t[2] - t[0]) / (R + 2));
  }
  return 0;
}

}  // namespace pidm

// measurement aid: the four stamps the last traced launch of conv3x3_rs_kernel left (PIDM_RS_TRACE=1): shader-clock counter and
// 100 MHz real-time counter of workgroup 0 / wave 0 before and after its row loop
extern "C" int pidm_debug_conv_rs_trace(unsigned long long* out4) {
  if (!out4) return pidm::fail("debug_conv_rs_trace: null argument");
  return hipMemcpyFromSymbol(out4, HIP_SYMBOL(pidm::g_rs_trace), sizeof(unsigned long long) * 4) == hipSuccess ? 0 : -1;
}
