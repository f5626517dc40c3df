// Implicit-GEMM convolutions on the fp32 matrix cores (v_mfma_f32_32x32x2_f32): the kernels every convolution ran on in round 1 and
// the fall-back of the split-form kernels of k_conv.hip since (PIDM_CONV_SPLIT=0, shapes those do not take: ragged channel counts, the
// memory-bound 1x1 layers, the linears).  Replaces `convolution` / `conv_transpose` / `addmm` of the reference UNet
// (src/unet_model.py:163,197,227,253,275,279,453,517, :248,332,339,466,468).  Split out of k_conv.hip in round 4.
//   conv_igemm_kernel<KC,NT>                       generic forward / dgrad (ragged channels, 7x7, tap groups, scalar staging)
//   conv_igemm_pipe_kernel<KC,NT,..,PHASED,PERSIST,MT>   software-pipelined forward / dgrad and its variants
//   conv3x3_stream_kernel                          streaming persistent 3x3 kernel
//
// Data layout: activations channels-last (NHWC == the reference's [B, P*P, C] interchange layout), weights
// re-packed once per step to [Cout_p][tap][Cin_p] (K contiguous) by `pack_kernel`.
// GEMM view: M = output pixels, N = Cout, K = taps x Cin.   One workgroup = 4 waves = a 128-pixel x (32*NT)
// channel tile; each wave owns one 32-pixel m-tile x NT 32x32 accumulators.  The input tile WITH ITS HALO is
// staged once per Cin-chunk in LDS and re-used by all KHxKW taps (9x fewer global->LDS bytes for 3x3), the
// weight slab of the tap group sits next to it.  LDS rows are KC+4 floats so the per-lane ds_read_b128 of 4
// consecutive k (lanes = consecutive pixels / output channels) is bank-conflict free.
// The MFMA k-slots are permuted (lane-half h supplies channels 8g+4h+s for step s): A and B use the same
// permutation, so the sum is unchanged and each operand fetch is one 16-byte LDS read per 4 MFMAs.
//
#include <stdio.h>
#include <stdlib.h>

#include "pidm_launch.h"
#include "k_conv_epilogue.h"

namespace pidm {

#ifndef PIDM_ABLATE_FLAGS
#define PIDM_ABLATE_FLAGS 0
#endif
static constexpr int kAblate = PIDM_ABLATE_FLAGS;

// ---------------------------------------------------------------------------------------------------
// forward / dgrad kernel
// ---------------------------------------------------------------------------------------------------
template <int KC, int NT>
__global__ void __launch_bounds__(256) conv_igemm_kernel(ConvGeom g, int tgs, int sigmoid_last,
                                                         const float* __restrict__ src0, const float* __restrict__ src1,
                                                         const float* __restrict__ wp, const float* __restrict__ bias,
                                                         const float* __restrict__ residual, float* __restrict__ out) {
  constexpr int KCP = KC + 4;
  constexpr int BN = 32 * NT;
  constexpr int Q = KC / 4;  // float4 quads per chunk
  HIP_DYNAMIC_SHARED(float, smem)
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int half = lane >> 5, l31 = lane & 31;
  const int z = blockIdx.z;
  const int tiles_n = (g.Cout + BN - 1) / BN;
  const int tile_m = blockIdx.x / tiles_n, tile_n = blockIdx.x % tiles_n;
  const int n0 = tile_n * BN;
  const int T = g.KH * g.KW;
  const int CinP = (g.Cin + KC - 1) / KC * KC;
  const int npixA = g.NI * g.IHt * g.IWt;
  float* As = smem;
  float* Bs = smem + (size_t)npixA * KCP;

  const int tpi = g.Hv / g.TH;                 // tiles per image (1 when NI > 1)
  const int b0 = (tile_m / tpi) * g.NI;
  const int vy0 = (tile_m % tpi) * g.TH;
  const int iy0 = vy0 * g.stride - g.pad_y[z];
  const int ix0 = -g.pad_x[z];
  const float* wz = wp + g.w_off[z];
  const bool vec_ok = ((g.ld0 & 3) == 0) && ((g.ld1 & 3) == 0) && ((g.C0 & 3) == 0) && ((g.Cin & 3) == 0);

  // this lane's A row (pixel) inside the wave's m-tile
  const int pm = wave * 32 + l31;
  const int a_tx = pm & (g.Wv - 1), a_ty = (pm >> g.wsh) & (g.TH - 1), a_img = pm >> (g.wsh + g.tsh);
  // (g.NI may be smaller than 128/(Wv*TH) when the halo tile of tiny strided images would not fit in LDS:
  //  rows of the missing images read tile pixel 0 and are discarded in the epilogue)
  const int abase = (a_img < g.NI) ? (a_img * g.IHt + a_ty * g.stride) * g.IWt + a_tx * g.stride : 0;

  f32x16 acc[NT];
#pragma unroll
  for (int i = 0; i < NT; ++i)
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;

  // per-tap halo offsets (pixels), computed once: no integer divisions in the MFMA loop
  __shared__ int tap_off[64];
  if (tid < T) tap_off[tid] = (tid / g.KW) * g.IWt + (tid % g.KW);

  const int rows = g.NI * g.IHt;   // halo rows of the tile
  const int rowf4 = g.IWt * Q;     // 16-byte quads per halo row
  for (int c0 = 0; c0 < CinP; c0 += KC) {
    __syncthreads();
    // ---- stage the input tile (with halo) for channels [c0, c0+KC): one wave per halo row ----
    for (int rrow = wave; rrow < rows; rrow += 4) {
      const int img = rrow / g.IHt, hy = rrow - img * g.IHt;
      const int b = b0 + img, iy = iy0 + hy;
      const bool rowvalid = (b < g.B) && (iy >= 0) && (iy < g.Hi);
      const size_t rowpix = ((size_t)b * g.Hi + iy) * g.Wi;
      float* arow_s = As + (size_t)rrow * g.IWt * KCP;
      for (int e = lane; e < rowf4; e += 64) {
        const int hx = e / Q, q = e % Q;
        const int ix = ix0 + hx, c = c0 + 4 * q;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (rowvalid && ix >= 0 && ix < g.Wi && c < g.Cin) {
          const size_t pix = rowpix + ix;
          if (vec_ok) {
            v = (c < g.C0) ? *reinterpret_cast<const float4*>(src0 + pix * g.ld0 + c)
                           : *reinterpret_cast<const float4*>(src1 + pix * g.ld1 + (c - g.C0));
          } else {
            float t4[4];
            for (int k = 0; k < 4; ++k) {
              const int ck = c + k;
              t4[k] = (ck < g.Cin) ? ((ck < g.C0) ? src0[pix * g.ld0 + ck] : src1[pix * g.ld1 + (ck - g.C0)]) : 0.f;
            }
            v = make_float4(t4[0], t4[1], t4[2], t4[3]);
          }
        }
        *reinterpret_cast<float4*>(arow_s + (size_t)hx * KCP + 4 * q) = v;
      }
    }
    for (int t0 = 0; t0 < T; t0 += tgs) {
      const int nt = (T - t0 < tgs) ? (T - t0) : tgs;
      if (t0 > 0) __syncthreads();
      // ---- stage the weight slab of this tap group: Bs[tl][row][KCP] ----
      for (int e = tid; e < nt * BN * Q; e += 256) {
        const int q = e % Q, row = (e / Q) % BN, tl = e / (Q * BN);
        const float4 v = *reinterpret_cast<const float4*>(wz + ((size_t)(n0 + row) * T + (t0 + tl)) * CinP + c0 + 4 * q);
        *reinterpret_cast<float4*>(Bs + ((size_t)tl * BN + row) * KCP + 4 * q) = v;
      }
      __syncthreads();
      // ---- MFMA over the taps of the group ----
      for (int tl = 0; tl < nt; ++tl) {
        const float* arow = As + (size_t)(abase + tap_off[t0 + tl]) * KCP + 4 * half;
        const float* brow = Bs + ((size_t)tl * BN + l31) * KCP + 4 * half;
#pragma unroll
        for (int g8 = 0; g8 < KC / 8; ++g8) {
          const float4 a4 = *reinterpret_cast<const float4*>(arow + 8 * g8);
          float4 b4[NT];
#pragma unroll
          for (int ni = 0; ni < NT; ++ni) b4[ni] = *reinterpret_cast<const float4*>(brow + (size_t)ni * 32 * KCP + 8 * g8);
#pragma unroll
          for (int s = 0; s < 4; ++s) {
#pragma unroll
            for (int ni = 0; ni < NT; ++ni) acc[ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[s], b4[ni][s], acc[ni], 0, 0, 0);
          }
        }
      }
    }
  }

  // ---- epilogue: bias, residual, (sigmoid), store ----
#pragma unroll
  for (int ni = 0; ni < NT; ++ni) {
    const int c = n0 + ni * 32 + l31;
    if (c >= g.Cout) continue;
    const float bv = bias ? bias[c] : 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = (r & 3) + 8 * (r >> 2) + 4 * half;
      const int p = wave * 32 + row;
      const int tx = p & (g.Wv - 1), ty = (p >> g.wsh) & (g.TH - 1), img = p >> (g.wsh + g.tsh);
      const int b = b0 + img;
      if (b >= g.B || img >= g.NI) continue;
      const int oy = (vy0 + ty) * g.os + g.ooy[z], ox = tx * g.os + g.oox[z];
      float v = acc[ni][r] + bv;
      if (residual) v += residual[(((size_t)b * g.Ho + oy) * g.Wo + ox) * g.ldr + c];
      if (sigmoid_last && c == g.Cout - 1) v = 1.f / (1.f + expf(-v));
      out[(size_t)b * g.sob + (size_t)oy * g.soy + (size_t)ox * g.sox + (size_t)c * g.soc] = v;
    }
  }
}

// ---------------------------------------------------------------------------------------------------
// software-pipelined variant (the hot one): all taps of a Cin-chunk in one LDS slab (T <= 9), Cin % KC == 0,
// 16-byte aligned sources.  Per thread the (halo pixel, quad) -> (global offset, LDS offset) decode is done ONCE
// in the prologue; per chunk the thread issues its <= AMAX + BMAX global loads for chunk i+1 into registers right
// after the barrier that publishes chunk i, so HBM/L2 latency hides under the 9*KC/2*NT MFMAs of chunk i.
// ---------------------------------------------------------------------------------------------------
// PERSIST: a workgroup walks g.tpw consecutive m-tiles of one n-tile; (tile, Cin-chunk) stages form ONE software
// pipeline - the first chunk of the next tile is prefetched under the MFMAs of the current tile's last chunk and the
// epilogue stores drain under the next tile's MFMAs.  Short-K layers (64x64: Cin = 32, two chunks per tile) otherwise
// spend most of a workgroup's life in its prologue/epilogue with the matrix pipe idle (PMC: 45 % MFMA busy).
// MT = m-tiles (32 pixels each) per wave: MT == 2 is the 256-pixel workgroup tile (two independent accumulator chains per
// wave, half the barriers / weight staging / B-fragment reads per MFMA).
template <int KC, int NT, int AMAX, int BMAX, int KH, int KW, bool PHASED, bool PERSIST, int MT>
__global__ void __launch_bounds__(256) conv_igemm_pipe_kernel(ConvGeom g, int sigmoid_last, const float* __restrict__ src0,
                                                              const float* __restrict__ src1, const float* __restrict__ wp,
                                                              const float* __restrict__ bias,
                                                              const float* __restrict__ residual, float* __restrict__ out) {
  constexpr int KCP = KC + 4;
  constexpr int BN = 32 * NT;
  constexpr int Q = KC / 4;
  HIP_DYNAMIC_SHARED(float, smem)
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int half = lane >> 5, l31 = lane & 31;
  const int z = blockIdx.z;
  const int tiles_n = (g.Cout + BN - 1) / BN;
  const int tile_n = blockIdx.x % tiles_n;
  const int tm_first = PERSIST ? (int)(blockIdx.x / tiles_n) * g.tpw : (int)(blockIdx.x / tiles_n);
  const int tm_end = PERSIST ? ((tm_first + g.tpw < g.tiles_m) ? tm_first + g.tpw : g.tiles_m) : tm_first + 1;
  const int n0 = tile_n * BN;
  constexpr int T = KH * KW;
  const int npixA = g.NI * g.IHt * g.IWt;
  float* As = smem;
  float* Bs = smem + (size_t)npixA * KCP;
  const int tpi = g.Hv / g.TH;
  const int ix0 = -g.pad_x[z];
  const float* wz = wp + g.w_off[z];

  int abase[MT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    const int pm = mt * 128 + wave * 32 + l31;
    const int a_tx = pm & (g.Wv - 1), a_ty = (pm >> g.wsh) & (g.TH - 1), a_img = pm >> (g.wsh + g.tsh);
    abase[mt] = (a_img < g.NI) ? (a_img * g.IHt + a_ty * g.stride) * g.IWt + a_tx * g.stride : 0;
  }

  // ---- staging slots of this thread (q = tid % Q is the same for every slot).  The LDS side is tile independent;
  //      the global side (a_pix) is decoded per tile by PIDM_SET_TILE ----
  const int nA = npixA * Q, nB = T * BN * Q;
  const int aq = tid % Q;
  int a_pix[AMAX];    // global pixel index, -1: zero fill (PHASED: packed tile-relative (img<<20 | hy<<10 | hx))
  int a_lds[AMAX];    // float offset in As, -1: slot unused
#pragma unroll
  for (int k = 0; k < AMAX; ++k) {
    const int e = tid + k * 256;
    a_pix[k] = -1;
    a_lds[k] = (e < nA) ? (e / Q) * KCP + 4 * aq : -1;
  }
  int p_b0 = 0, p_vy0 = 0;   // geometry of the tile the NEXT prefetch belongs to
#define PIDM_SET_TILE(tm_)                                                                                         \
  {                                                                                                                \
    const int tm__ = (tm_);                                                                                        \
    p_b0 = (tm__ / tpi) * g.NI;                                                                                    \
    p_vy0 = (tm__ % tpi) * g.TH;                                                                                   \
    const int iy0__ = p_vy0 * g.stride - g.pad_y[z];                                                               \
    _Pragma("unroll") for (int k = 0; k < AMAX; ++k) {                                                             \
      const int e = tid + k * 256;                                                                                 \
      a_pix[k] = -1;                                                                                               \
      if (e < nA) {                                                                                                \
        const int hp = e / Q;                                                                                      \
        const int hrow = fast_div(hp, g.IWt, g.mIWt), hx = hp - hrow * g.IWt;                                      \
        const int img = fast_div(hrow, g.IHt, g.mIHt), hy = hrow - img * g.IHt;                                    \
        const int b = p_b0 + img, iy = iy0__ + hy, ix = ix0 + hx;                                                  \
        if (PHASED) {                                                                                              \
          a_pix[k] = (img << 20) | (hy << 10) | hx;   /* the source pixel depends on the K-phase */                \
        } else if (b < g.B && iy >= 0 && iy < g.Hi && ix >= 0 && ix < g.Wi) {                                      \
          a_pix[k] = (b * g.Hi + iy) * g.Wi + ix;                                                                  \
        }                                                                                                          \
      }                                                                                                            \
    }                                                                                                              \
  }
  const int CinP = g.Kw;  // packed weight row length (nph * Cin); Cin % KC == 0 for this kernel
  // B slot k of this thread: e = tid + 256k -> (q = e % Q, row = (e / Q) % BN, tl = e / (Q*BN)); all powers of two
  const int bq = tid % Q, brow0 = (tid / Q) % BN;
  int b_g[BMAX], b_l[BMAX];  // per-slot global (floats, relative to the n-tile's first row) and LDS offsets
#pragma unroll
  for (int k = 0; k < BMAX; ++k) {
    const int e = tid + k * 256;
    const int row = (brow0 + (k * 256 / Q)) % BN, tl = (e < nB) ? e / (Q * BN) : 0;
    // NT == 4 ("permuted" 128-channel tile): output channel 4j+s of the tile is computed by lane j of accumulator s, so
    // that a lane's 4 accumulators are 4 CONSECUTIVE channels (16-byte stores); its weight row sits at LDS row 32s+j
    const int lrow = (NT == 4) ? ((row & 3) * 32 + (row >> 2)) : row;
    b_g[k] = (row * T + tl) * CinP + 4 * bq;
    b_l[k] = (e < nB) ? (tl * BN + lrow) * KCP + 4 * bq : -1;
  }
  const float* wn = wz + (size_t)n0 * T * CinP;
  f32x4 ra[AMAX], rb[BMAX];  // native vectors: stay in VGPRs across the loop back-edge

#define PIDM_PREFETCH(c0_)                                                                                         \
  if (kAblate & 1) {                                                                                               \
    _Pragma("unroll") for (int k = 0; k < AMAX; ++k) ra[k] = f32x4{1.f, 0.f, 0.f, 0.f};                            \
    _Pragma("unroll") for (int k = 0; k < BMAX; ++k) rb[k] = f32x4{1.f, 0.f, 0.f, 0.f};                            \
  } else {                                                                                                         \
    const int kk__ = (c0_);                 /* position along the packed K axis */                                 \
    const int ph__ = PHASED ? kk__ / g.Cin : 0;                                                                    \
    const int c0__ = PHASED ? kk__ - ph__ * g.Cin : kk__;                                                          \
    const float* sp__ = (c0__ < g.C0) ? src0 + c0__ : src1 + (c0__ - g.C0);                                       \
    const int ld__ = (c0__ < g.C0) ? g.ld0 : g.ld1;                                                                \
    _Pragma("unroll") for (int k = 0; k < AMAX; ++k) {                                                             \
      ra[k] = f32x4{0.f, 0.f, 0.f, 0.f};                                                                           \
      if (PHASED) {                                                                                                \
        if (a_lds[k] >= 0) {                                                                                       \
          const int b = p_b0 + (a_pix[k] >> 20);                                                                   \
          const int iy = (p_vy0 - g.ph_pad_y[ph__] + ((a_pix[k] >> 10) & 1023)) * g.in_step + g.ph_oy[ph__];       \
          const int ix = ((a_pix[k] & 1023) - g.ph_pad_x[ph__]) * g.in_step + g.ph_ox[ph__];                       \
          if (b < g.B && iy >= 0 && iy < g.Hi && ix >= 0 && ix < g.Wi)                                             \
            ra[k] = *reinterpret_cast<const f32x4*>(sp__ + (((size_t)b * g.Hi + iy) * g.Wi + ix) * ld__ + 4 * aq); \
        }                                                                                                          \
      } else if (a_pix[k] >= 0) {                                                                                  \
        ra[k] = *reinterpret_cast<const f32x4*>(sp__ + (size_t)a_pix[k] * ld__ + 4 * aq);                          \
      }                                                                                                            \
    }                                                                                                              \
    _Pragma("unroll") for (int k = 0; k < BMAX; ++k) rb[k] = *reinterpret_cast<const f32x4*>(wn + b_g[k] + kk__);   \
  }

  f32x16 acc[MT * NT];   // [mt][ni]
#pragma unroll
  for (int i = 0; i < MT * NT; ++i)
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;

  PIDM_SET_TILE(tm_first)
  PIDM_PREFETCH(0)
  for (int tm = tm_first; tm < tm_end; ++tm) {
  const int b0 = p_b0, vy0 = p_vy0;   // geometry of the tile being computed (its epilogue runs after the next prefetch)
  for (int c0 = 0; c0 < CinP; c0 += KC) {
    __syncthreads();          // previous chunk's LDS reads are done
#pragma unroll
    for (int k = 0; k < AMAX; ++k)
      if (a_lds[k] >= 0) *reinterpret_cast<f32x4*>(As + a_lds[k]) = ra[k];
#pragma unroll
    for (int k = 0; k < BMAX; ++k)
      if (b_l[k] >= 0) *reinterpret_cast<f32x4*>(Bs + b_l[k]) = rb[k];
    __syncthreads();          // chunk c0 visible
    if (c0 + KC < CinP) {
      PIDM_PREFETCH(c0 + KC)
    } else if (PERSIST && tm + 1 < tm_end) {
      PIDM_SET_TILE(tm + 1)
      PIDM_PREFETCH(0)
    }
    // taps fully unrolled (compile-time KHxKW).  The LDS fragments of tap t+1 are fetched into a second register set
    // BEFORE the MFMAs of tap t are issued, so the ~128-cycle ds_read latency hides under 8*NT*KC/8 MFMAs instead of
    // stalling the matrix pipe once per tap.
    const float* bbase_p = Bs + (size_t)l31 * KCP + 4 * half;
    f32x4 fa[2][MT][KC / 8], fb[2][KC / 8][NT];
#define PIDM_LOAD_FRAGS(set_, t_)                                                                                  \
  {                                                                                                                \
    const float* brow = bbase_p + (size_t)((t_)*BN) * KCP;                                                         \
    _Pragma("unroll") for (int mt = 0; mt < MT; ++mt) {                                                            \
      const float* arow = As + (size_t)(abase[mt] + ((t_) / KW) * g.IWt + ((t_) % KW)) * KCP + 4 * half;           \
      _Pragma("unroll") for (int g8 = 0; g8 < KC / 8; ++g8)                                                        \
          fa[set_][mt][g8] = *reinterpret_cast<const f32x4*>(arow + 8 * g8);                                       \
    }                                                                                                              \
    _Pragma("unroll") for (int g8 = 0; g8 < KC / 8; ++g8) {                                                        \
      _Pragma("unroll") for (int ni = 0; ni < NT; ++ni)                                                            \
          fb[set_][g8][ni] = *reinterpret_cast<const f32x4*>(brow + (size_t)ni * 32 * KCP + 8 * g8);               \
    }                                                                                                              \
  }
    PIDM_LOAD_FRAGS(0, 0)
    __builtin_amdgcn_sched_group_barrier(0x100, (MT + NT) * (KC / 8), 0);   // tap 0's reads form their own group
#pragma unroll
    for (int t = 0; t < T; ++t) {
      const int cur = t & 1;
      if (t + 1 < T) PIDM_LOAD_FRAGS(cur ^ 1, t + 1)
#pragma unroll
      for (int g8 = 0; g8 < KC / 8; ++g8) {
#pragma unroll
        for (int s = 0; s < 4; ++s) {
#pragma unroll
          for (int ni = 0; ni < NT; ++ni)
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
              if (kAblate & 4) acc[mt * NT + ni][(g8 * 4 + s) & 15] += fa[cur][mt][g8][s] * fb[cur][g8][ni][s];
              else acc[mt * NT + ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[cur][mt][g8][s], fb[cur][g8][ni][s], acc[mt * NT + ni], 0, 0, 0);
            }
        }
      }
      // pin the order hipcc otherwise undoes (it sinks the next tap's ds_reads below this tap's MFMAs):
      // first the (MT + NT) * KC/8 LDS reads of tap t+1, then the 4 * MT * NT * KC/8 MFMAs of tap t
      if (t + 1 < T) __builtin_amdgcn_sched_group_barrier(0x100, (MT + NT) * (KC / 8), 0);
      __builtin_amdgcn_sched_group_barrier(0x008, 4 * MT * NT * (KC / 8), 0);
    }
#undef PIDM_LOAD_FRAGS
  }

  // ---- epilogue of tile tm: bias, residual, (sigmoid), store; PERSIST: the accumulators restart at zero ----
  static_assert(NT != 4 || MT == 1, "the permuted 128-channel tile is a 1x1-conv configuration (MT == 1)");
  if constexpr (NT == 4) {
    // permuted tile: lane l31 owns channels n0 + 4*l31 .. +3 (one per accumulator) of 16 pixel rows -> one 16-byte
    // store per row: a wave instruction writes 2 x 512 contiguous bytes instead of 2 x 128
    const int c = n0 + 4 * l31;
    const bool vec = g.soc == 1 && ((g.sox | g.soy | g.sob) & 3) == 0 && (reinterpret_cast<size_t>(out) & 15) == 0 &&
                     (!residual || ((g.ldr & 3) == 0 && (reinterpret_cast<size_t>(residual) & 15) == 0)) && c + 4 <= g.Cout;
    float bv[4];
#pragma unroll
    for (int s4 = 0; s4 < 4; ++s4) bv[s4] = (bias && c + s4 < g.Cout) ? bias[c + s4] : 0.f;
    // the residual quads of four rows at a time as one batch of loads (rows outside the batch / image re-read the first quad of
    // the tensor and are skipped below): per row hipcc emitted branch - load - s_waitcnt vmcnt(0) - store, a memory round trip each
#pragma unroll
    for (int r0 = 0; r0 < 16; r0 += 4) {
    f32x4 rq[4];
    if (residual && vec) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int r = r0 + i;
        const int row = (r & 3) + 8 * (r >> 2) + 4 * half;
        const int p = wave * 32 + row;
        const int tx = p & (g.Wv - 1), ty = (p >> g.wsh) & (g.TH - 1), img = p >> (g.wsh + g.tsh);
        const int b = b0 + img;
        const int oy = (vy0 + ty) * g.os + g.ooy[z], ox = tx * g.os + g.oox[z];
        const float* q = (b < g.B && img < g.NI) ? residual + (((size_t)b * g.Ho + oy) * g.Wo + ox) * g.ldr + c : residual;
        rq[i] = *reinterpret_cast<const f32x4*>(q);
      }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int r = r0 + i;
      const int row = (r & 3) + 8 * (r >> 2) + 4 * half;
      const int p = wave * 32 + row;
      const int tx = p & (g.Wv - 1), ty = (p >> g.wsh) & (g.TH - 1), img = p >> (g.wsh + g.tsh);
      const int b = b0 + img;
      if (b >= g.B || img >= g.NI) continue;
      const int oy = (vy0 + ty) * g.os + g.ooy[z], ox = tx * g.os + g.oox[z];
      float* op = out + (size_t)b * g.sob + (size_t)oy * g.soy + (size_t)ox * g.sox;
      const float* rp = residual ? residual + (((size_t)b * g.Ho + oy) * g.Wo + ox) * g.ldr : nullptr;
      if (vec) {
        f32x4 o = {acc[0][r] + bv[0], acc[1][r] + bv[1], acc[2][r] + bv[2], acc[3][r] + bv[3]};
        if (rp) o += rq[i];
        if (sigmoid_last && c + 4 == g.Cout) o[3] = 1.f / (1.f + expf(-o[3]));
        *reinterpret_cast<f32x4*>(op + c) = o;
      } else {
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4) {
          if (c + s4 >= g.Cout) continue;
          float o = acc[s4 % NT][r] + bv[s4];
          if (rp) o += rp[c + s4];
          if (sigmoid_last && c + s4 == g.Cout - 1) o = 1.f / (1.f + expf(-o));
          op[(size_t)(c + s4) * g.soc] = o;
        }
      }
    }
    }
  } else if (g.Wv >= 32) {
    // the wave's 32 pixels are consecutive in x inside one image row: one 64-bit base per (wave, n-tile), then
    // row * (os*sox) steps with compile-time row constants (no per-row index math)
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
    const int p0 = mt * 128 + wave * 32;
    const int tx0 = p0 & (g.Wv - 1), ty = (p0 >> g.wsh) & (g.TH - 1), img = p0 >> (g.wsh + g.tsh);
    const int b = b0 + img;
    if (b < g.B && img < g.NI) {
      const int oy = (vy0 + ty) * g.os + g.ooy[z];
      const long rstep = (long)g.os * g.sox, rrstep = (long)g.os * g.ldr;
      const long opix = (long)b * g.sob + (long)oy * g.soy + (long)(tx0 * g.os + g.oox[z]) * g.sox + 4 * half * rstep;
      const long rpix = (((long)b * g.Ho + oy) * g.Wo + (tx0 * g.os + g.oox[z])) * g.ldr + 4 * half * rrstep;
#pragma unroll
      for (int ni = 0; ni < NT; ++ni) {
        const int c = n0 + ni * 32 + l31;
        if (c >= g.Cout) continue;
        const float bv = bias ? bias[c] : 0.f;
        float* op = out + opix + (long)c * g.soc;
        const float* rp = residual ? residual + rpix + c : nullptr;
        const bool sig = sigmoid_last && c == g.Cout - 1;
        float gs1 = 0.f, gs2 = 0.f;
        // the 16 residual values as ONE batch of loads (one branch): inside the row loop hipcc gave every row its own branch, load
        // and s_waitcnt vmcnt(0) - which also waits for the previous row's store: 16 memory round trips per tile in a 18 us kernel
        float rv[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) rv[r] = 0.f;
        if (rp) {
#pragma unroll
          for (int r = 0; r < 16; ++r) rv[r] = rp[((r & 3) + 8 * (r >> 2)) * rrstep];
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int rowc = (r & 3) + 8 * (r >> 2);   // compile-time part of the row index
          float v = acc[mt * NT + ni][r] + bv;
          if (rp) v += rv[r];
          if (sig) v = 1.f / (1.f + expf(-v));
          if (!(kAblate & 2) || v == 1.2345e30f) op[rowc * rstep] = v;
          gs1 += v;
          gs2 += v * v;
        }
        if (g.gn_part) PIDM_GN_PARTIAL(gs1, gs2, b, (vy0 + ty) * g.Wv + tx0, c)
        if (g.bn_part) {   // plain stride-1 geometry (os == 1, output grid == x grid): the wave's pixels are consecutive in x
          const float* xrow = g.bn_x + (((size_t)b * g.Ho + oy) * g.Wo + tx0) * g.Cout + c;
          PIDM_BN_PARTIAL(acc[mt * NT + ni], bv, b, (vy0 + ty) * g.Wv + tx0, c, xrow, g.Cout, false, (const float*)nullptr, 0)
        }
      }
    }
    }
  } else {
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int ni = 0; ni < NT; ++ni) {
      const int c = n0 + ni * 32 + l31;
      if (c >= g.Cout) continue;
      const float bv = bias ? bias[c] : 0.f;
      float gs1 = 0.f, gs2 = 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = (r & 3) + 8 * (r >> 2) + 4 * half;
        const int p = mt * 128 + wave * 32 + row;
        const int tx = p & (g.Wv - 1), ty = (p >> g.wsh) & (g.TH - 1), img = p >> (g.wsh + g.tsh);
        const int b = b0 + img;
        if (b >= g.B || img >= g.NI) continue;
        const int oy = (vy0 + ty) * g.os + g.ooy[z], ox = tx * g.os + g.oox[z];
        float v = acc[mt * NT + ni][r] + bv;
        if (residual) v += residual[(((size_t)b * g.Ho + oy) * g.Wo + ox) * g.ldr + c];
        if (sigmoid_last && c == g.Cout - 1) v = 1.f / (1.f + expf(-v));
        out[(size_t)b * g.sob + (size_t)oy * g.soy + (size_t)ox * g.sox + (size_t)c * g.soc] = v;
        gs1 += v;
        gs2 += v * v;
      }
      if (g.gn_part) {   // the wave's 32 pixels are whole rows of ONE image (32 % Wv == 0, H*W % 32 == 0): wave-uniform b and chunk
        const int p0w = mt * 128 + wave * 32;
        const int ty0 = (p0w >> g.wsh) & (g.TH - 1), img0 = p0w >> (g.wsh + g.tsh);
        if (b0 + img0 < g.B && img0 < g.NI) PIDM_GN_PARTIAL(gs1, gs2, b0 + img0, (vy0 + ty0) * g.Wv, c)
      }
      if (g.bn_part) {   // 32 % Wv == 0: the wave's 32 pixels are whole, consecutive rows of one image = 32 consecutive pixels of x
        const int p0w = mt * 128 + wave * 32;
        const int ty0 = (p0w >> g.wsh) & (g.TH - 1), img0 = p0w >> (g.wsh + g.tsh);
        if (b0 + img0 < g.B && img0 < g.NI) {
          const float* xrow = g.bn_x + (((size_t)(b0 + img0) * g.Ho + (vy0 + ty0)) * g.Wo) * g.Cout + c;
          PIDM_BN_PARTIAL(acc[mt * NT + ni], bv, b0 + img0, (vy0 + ty0) * g.Wv, c, xrow, g.Cout, false, (const float*)nullptr, 0)
        }
      }
    }
  }
  if constexpr (PERSIST) {
#pragma unroll
    for (int i = 0; i < MT * NT; ++i)
      for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  }
  }
#undef PIDM_PREFETCH
#undef PIDM_SET_TILE
}

// ---------------------------------------------------------------------------------------------------
// Streaming 3x3 / stride-1 convolution (forward and dgrad of every resblock conv with Cin % 32 == 0, Cout % 32 == 0).
//
// Why a second kernel: conv_igemm_pipe_kernel runs a workgroup as load -> barrier -> MFMAs -> store, and the two workgroups
// that fit a CU start together and stay in lock-step, so the chip alternates between a load burst and a compute phase
// (tools/conv_probe.py, 64x64 32->32 at batch 64: 62 us per launch; 43 us with neither loads nor stores, 42 us with no MFMAs at
// all, 31 us MFMA-only floor).  Here ONE persistent workgroup per CU walks its (m-tile, 32-channel chunk) stages as a single
// software pipeline over two LDS buffers:
//   stage s:  MFMAs read buffer s&1 | tap t's slot of stage s+1 (already in registers) is written to buffer (s+1)&1 right
//             after tap t's MFMAs | the same registers are re-loaded with stage s+2's slot | ONE barrier per stage
// so operand fetch (a whole stage = 9216 matrix-pipe cycles ahead), LDS staging and the MFMAs of a wave overlap instead of
// taking turns, across chunks AND across tiles.  The stage body is branch-free (out-of-image pixels: clamped address +
// select; slots beyond the halo tile: a dump location in the row padding), which lets the scheduler interleave it.
// Tile: 128 pixels x 32 output channels, K chunk 32 channels x 9 taps (144 MFMAs per wave and stage), LDS rows of 36 floats.
// ---------------------------------------------------------------------------------------------------
// cycle stamps of workgroup 0 / wave 0 (tools/conv_trace.py; PIDM_STREAM_TRACE=1): [stage][0..3] = stage top, first MFMA issued,
// tap loop done, after the barrier
static __device__ unsigned long long g_stream_trace_fp32[4 * 64];   // (this kernel's stamps are no longer exported: pidm_debug_stream_trace reads the split-form kernels')

__global__ void __launch_bounds__(256) conv3x3_stream_kernel(ConvGeom g, int sigmoid_last, const float* __restrict__ src0,
                                                             const float* __restrict__ src1, const float* __restrict__ wp,
                                                             const float* __restrict__ bias, const float* __restrict__ residual,
                                                             float* __restrict__ out, int n_items, int items_per_wg, int trace) {
  constexpr int KCP = 36, T = 9;
  HIP_DYNAMIC_SHARED(float, smem)
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int half = lane >> 5, l31 = lane & 31;
  const int z = 0;
  const int npixA = g.NI * g.IHt * g.IWt;
  const int bufsz = (npixA + T * 32) * KCP;          // floats per buffer: halo tile | 9 x 32 weight rows
  const int tpi = g.Hv / g.TH;
  const int NCH = g.Cin >> 5;
  const int CinP = g.Kw;
  const int item0 = blockIdx.x * items_per_wg;
  const int my_items = (item0 + items_per_wg <= n_items) ? items_per_wg : n_items - item0;
  const int nst = my_items * NCH;

  // fragment base of this lane's A row (pixel) inside the halo tile
  const int pm = wave * 32 + l31;
  const int a_tx = pm & (g.Wv - 1), a_ty = (pm >> g.wsh) & (g.TH - 1), a_img = pm >> (g.wsh + g.tsh);
  const int abase = (a_img < g.NI) ? (a_img * g.IHt + a_ty) * g.IWt + a_tx : 0;

  // Staging.  The fp32 MFMA runs on the SIMD's vector ALUs (tools/mfma_overlap.hip: every VALU instruction of a wave ADDS its
  // issue time to the wave's MFMA chain, with one or two waves per SIMD alike), so the staging path carries no per-element
  // vector arithmetic: the halo COLUMNS (x = -1, x = W) are zero for every tile - written once, never staged again - and
  // the staged part of the halo tile is whole image rows, 8 pixels (one row segment) per wave and slot, so that row validity
  // (top / bottom padding, images past the batch) and the row's global base are wave-uniform scalars.  A slot's global
  // address is scalar base + a per-thread constant byte offset; LDS addresses are per-thread constants as well.
  const int aq = tid & 7;
  const int SEG = g.NI * g.IHt * g.Wv;               // staged pixels per tile (multiple of 32), slot k = pixels 32k .. 32k+31
  const int AS = SEG >> 5;                           // <= 8 slots
  const int wv8 = __builtin_amdgcn_readfirstlane(wave) * 8;
  int a_lds[8];
  unsigned a_vo[2];                                  // byte offset of (x, channel quad) inside a row, by slot parity
  int s_img[8], s_hy[8];                             // wave-uniform: image and halo row of this wave's segment in slot k
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    const int sp = (tid >> 3) + 32 * k;
    const int sr = sp >> g.wsh, x = sp & (g.Wv - 1);
    const int img = fast_div(sr, g.IHt, g.mIHt), hy = sr - img * g.IHt;
    // slots past the tile (k >= AS) are loaded and written like the others - no branch in the stage body, exact vmcnt
    // bookkeeping - but land in the 16 bytes of row padding, which nothing reads
    a_lds[k] = (k < AS) ? ((img * g.IHt + hy) * g.IWt + x + 1) * KCP + 4 * aq : ((tid >> 3) % npixA) * KCP + 32;
    if (k < 2) a_vo[k] = (unsigned)(x * g.ld0 + 4 * aq) * 4u;
    const int srw = (wv8 + 32 * k) >> g.wsh;
    s_img[k] = fast_div(srw, g.IHt, g.mIHt);
    s_hy[k] = srw - s_img[k] * g.IHt;
  }
  const int b_lds0 = npixA * KCP + (tid >> 3) * KCP + 4 * aq;     // + k * 32 * KCP
  const unsigned b_vo = (unsigned)((tid >> 3) * T * CinP + 4 * aq) * 4u;   // + (k * CinP + n0 * T * CinP + c0) * 4 as a scalar

  // zero halo columns of both buffers (and the whole pad region of unused rows stays untouched: never read)
  for (int e = tid; e < 2 * g.NI * g.IHt * 2 * 8; e += 256) {
    const int q = e & 7, side = (e >> 3) & 1, row = (e >> 4) % (g.NI * g.IHt), bufi = (e >> 4) / (g.NI * g.IHt);
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
    *reinterpret_cast<f32x4*>(smem + (size_t)bufi * bufsz + (size_t)(row * g.IWt + (side ? g.IWt - 1 : 0)) * KCP + 4 * q) = zero4;
  }

  f32x4 ra[8], rb[T];
  unsigned amask = 0;          // wave-uniform: bit k = slot k of the loads in flight is a real image row
  // stage geometry of the loads in flight (all scalar)
  const char* l_sp = reinterpret_cast<const char*>(src0);
  const char* l_wn = reinterpret_cast<const char*>(wp);
  int l_b0 = 0, l_iy0 = 0;
#define PIDM_ST_STAGE(s_)                                                                                          \
  {                                                                                                                \
    int ss__ = (s_);                                                                                               \
    if (ss__ >= nst) ss__ = nst - 1;                                                                               \
    const int it__ = item0 + ss__ / NCH, ch__ = ss__ - (ss__ / NCH) * NCH;                                         \
    const int tn__ = it__ / g.tiles_m, tm__ = it__ - tn__ * g.tiles_m;                                             \
    const int c0__ = ch__ * 32;                                                                                    \
    l_b0 = (tm__ / tpi) * g.NI;                                                                                    \
    l_iy0 = (tm__ % tpi) * g.TH - g.pad_y[z];                                                                      \
    l_sp = reinterpret_cast<const char*>((c0__ < g.C0) ? src0 + c0__ : src1 + (c0__ - g.C0));                      \
    l_wn = reinterpret_cast<const char*>(wp + (size_t)tn__ * 32 * T * CinP + c0__);                                \
  }
#define PIDM_ST_LOAD_A(k_)                                                                                         \
  {                                                                                                                \
    /* unconditional load (a branch around it makes the compiler's vmcnt bookkeeping wait for this stage's loads): */ \
    /* padding rows read row 0 of the source and are zeroed with a scalar-conditioned select when they go to LDS */  \
    const int b__ = l_b0 + s_img[k_], iy__ = l_iy0 + s_hy[k_];                                                     \
    const bool ok__ = (b__ < g.B) & (iy__ >= 0) & (iy__ < g.Hi);        /* wave-uniform */                         \
    const size_t row__ = ok__ ? (size_t)(b__ * g.Hi + iy__) * g.Wi : 0;                                            \
    ra[k_] = *reinterpret_cast<const f32x4*>(l_sp + row__ * (size_t)g.ld0 * 4 + a_vo[(k_) & 1]);                   \
    amask = (amask & ~(1u << (k_))) | ((ok__ ? 1u : 0u) << (k_));                                                  \
  }
#define PIDM_ST_LOAD_B(k_) rb[k_] = *reinterpret_cast<const f32x4*>(l_wn + (size_t)(k_)*CinP * 4 + b_vo);
#define PIDM_ST_WRITE_A(k_, buf_)                                                                                  \
  {                                                                                                                \
    const float keep__ = ((amask >> (k_)) & 1u) ? 1.f : 0.f;        /* scalar */                                   \
    *reinterpret_cast<f32x4*>((buf_) + a_lds[k_]) = ra[k_] * keep__;                                               \
  }
#define PIDM_ST_WRITE_B(k_, buf_) *reinterpret_cast<f32x4*>((buf_) + b_lds0 + (k_)*32 * KCP) = rb[k_];

  float* bufc = smem;              // buffer the MFMAs read
  float* bufn = smem + bufsz;      // buffer being filled
  PIDM_ST_STAGE(0)
#pragma unroll
  for (int k = 0; k < 8; ++k) PIDM_ST_LOAD_A(k)
#pragma unroll
  for (int k = 0; k < T; ++k) PIDM_ST_LOAD_B(k)
#pragma unroll
  for (int k = 0; k < 8; ++k) PIDM_ST_WRITE_A(k, bufc)
#pragma unroll
  for (int k = 0; k < T; ++k) PIDM_ST_WRITE_B(k, bufc)
  PIDM_ST_STAGE(1)
#pragma unroll
  for (int k = 0; k < 8; ++k) PIDM_ST_LOAD_A(k)
#pragma unroll
  for (int k = 0; k < T; ++k) PIDM_ST_LOAD_B(k)
  __syncthreads();

  // two accumulator chains (even / odd MFMA of a tap): a single dependent chain issues every 69.5 cycles, two alternate at the
  // pipe's 64 (tools/mfma_overlap.hip); they are added once per tile in the epilogue
  f32x16 acc, acc1;
  for (int r = 0; r < 16; ++r) { acc[r] = 0.f; acc1[r] = 0.f; }
  const bool tr_on = trace && blockIdx.x == 0 && tid == 0;
  for (int s = 0; s < nst; ++s) {
    if (tr_on && s < 64) g_stream_trace_fp32[4 * s + 0] = clock64();
    PIDM_ST_STAGE(s + 2)          // geometry of the loads issued during this stage
    const float* As = bufc;
    const float* abase_p = As + (size_t)abase * KCP + 4 * half;
    const float* bbase_p = As + (size_t)npixA * KCP + (size_t)l31 * KCP + 4 * half;
    f32x4 fa[2][4], fb[2][4];
#define PIDM_ST_FRAGS(set_, t_)                                                                                    \
  {                                                                                                                \
    const float* arow = abase_p + (size_t)(((t_) / 3) * g.IWt + ((t_) % 3)) * KCP;                                 \
    const float* brow = bbase_p + (size_t)((t_)*32) * KCP;                                                         \
    _Pragma("unroll") for (int g8 = 0; g8 < 4; ++g8) {                                                             \
      fa[set_][g8] = *reinterpret_cast<const f32x4*>(arow + 8 * g8);                                               \
      fb[set_][g8] = *reinterpret_cast<const f32x4*>(brow + 8 * g8);                                               \
    }                                                                                                              \
  }
    PIDM_ST_FRAGS(0, 0)
    if (tr_on && s < 64) g_stream_trace_fp32[4 * s + 1] = clock64();
    // One wave per SIMD: nothing but this wave's own instruction stream feeds the matrix pipe, and an in-order wave parks at the
    // next MFMA until the pipe frees (64 cycles) - whatever else it has to do is free only if it sits BETWEEN two MFMAs in
    // small pieces.  So each tap is 16 pinned steps "MFMA j; piece j" (sched_barrier(0): nothing crosses): pieces 0..7 = the 8
    // fragment reads of the next tap, 8/9 = this tap's slots of the next stage to LDS, 10/11 = reload of those registers.
#pragma unroll
    for (int t = 0; t < T; ++t) {
      const int cur = t & 1;
      const float* arow_n = abase_p + (size_t)(((t + 1) / 3) * g.IWt + ((t + 1) % 3)) * KCP;
      const float* brow_n = bbase_p + (size_t)((t + 1) * 32) * KCP;
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        const int g8 = j >> 2, q4 = j & 3;
        if (kAblate & 4) acc[j] += fa[cur][g8][q4] * fb[cur][g8][q4];
        else if (j & 1) acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[cur][g8][q4], fb[cur][g8][q4], acc1, 0, 0, 0);
        else acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[cur][g8][q4], fb[cur][g8][q4], acc, 0, 0, 0);
        if (t + 1 < T && j < 4) fa[cur ^ 1][j] = *reinterpret_cast<const f32x4*>(arow_n + 8 * j);
        if (t + 1 < T && j >= 4 && j < 8) fb[cur ^ 1][j - 4] = *reinterpret_cast<const f32x4*>(brow_n + 8 * (j - 4));
        if (j == 8 && t < 8) PIDM_ST_WRITE_A(t, bufn)
        if (j == 9) PIDM_ST_WRITE_B(t, bufn)
        if (j == 10 && t < 8) PIDM_ST_LOAD_A(t)
        if (j == 11) PIDM_ST_LOAD_B(t)
        __builtin_amdgcn_sched_barrier(0);
      }
    }
#undef PIDM_ST_FRAGS
    if (tr_on && s < 64) g_stream_trace_fp32[4 * s + 2] = clock64();
    // ---- last chunk of a tile: epilogue (bias, residual, store, GroupNorm partial sums), accumulators restart ----
    // A lane holds ONE channel of 16 pixels; stored like that, every store instruction moves 4 bytes per lane and the wave spends
    // ~2700 cycles issuing its 16 stores (measured with PIDM_STREAM_TRACE; nothing overlaps them here).  Each 4x4 block (4 lanes
    // x 4 registers) is transposed in registers - two DPP exchange stages - so that a lane holds 4 consecutive channels of one
    // pixel and a store instruction writes 8 whole 128-byte pixel rows.
    const int it = item0 + s / NCH, ch = s - (s / NCH) * NCH;
    if (ch == NCH - 1) {
      const int tn = it / g.tiles_m, tm = it - tn * g.tiles_m;
      const int b0 = (tm / tpi) * g.NI, vy0 = (tm % tpi) * g.TH, n0 = tn * 32;
      const int c = n0 + l31;
      const float bv = bias ? bias[c] : 0.f;
      // the wave's 32 pixels are consecutive pixels of ONE image (tiles are whole image rows)
      const int p0 = wave * 32;
      const int tx0 = p0 & (g.Wv - 1), ty0 = (p0 >> g.wsh) & (g.TH - 1), img0 = p0 >> (g.wsh + g.tsh);
      const int b = b0 + img0;
      if (b < g.B && img0 < g.NI) {        // wave-uniform
        const int pin = (vy0 + ty0) * g.Wv + tx0;                      // first pixel of the wave inside its image
        float v[16];
        float gs1 = 0.f, gs2 = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          v[r] = (acc[r] + acc1[r]) + bv;
          gs1 += v[r];
          gs2 += v[r] * v[r];
        }
        if (g.gn_part) PIDM_GN_PARTIAL(gs1, gs2, b, pin, c)
        if (g.bn_part) {
          const float* xrow = g.bn_x + ((size_t)b * g.Ho * g.Wo + pin) * g.Cout + c;
          f32x16 accs;
#pragma unroll
          for (int r = 0; r < 16; ++r) accs[r] = acc[r] + acc1[r];
          PIDM_BN_PARTIAL(accs, bv, b, pin, c, xrow, g.Cout, false, (const float*)nullptr, 0)
        }
        const bool odd1 = (l31 & 1) != 0, odd2 = (l31 & 2) != 0;
        const size_t opix = (size_t)b * g.sob + (size_t)pin * g.sox + n0 + 4 * (l31 >> 2);
        const size_t rpix = ((size_t)b * g.Ho * g.Wo + pin) * g.ldr + n0 + 4 * (l31 >> 2);
#pragma unroll
        for (int q4 = 0; q4 < 4; ++q4) {
          float x0 = v[4 * q4], x1 = v[4 * q4 + 1], x2 = v[4 * q4 + 2], x3 = v[4 * q4 + 3];
          // stage 1: 2x2 blocks across lane bit 0
          const float r01 = pidm_quad_xor1(odd1 ? x0 : x1), r23 = pidm_quad_xor1(odd1 ? x2 : x3);
          x0 = odd1 ? r01 : x0; x1 = odd1 ? x1 : r01;
          x2 = odd1 ? r23 : x2; x3 = odd1 ? x3 : r23;
          // stage 2: across lane bit 1, registers (0,2) and (1,3)
          const float r02 = pidm_quad_xor2(odd2 ? x0 : x2), r13 = pidm_quad_xor2(odd2 ? x1 : x3);
          x0 = odd2 ? r02 : x0; x2 = odd2 ? x2 : r02;
          x1 = odd2 ? r13 : x1; x3 = odd2 ? x3 : r13;
          // lane (l31 & 3) = pixel 8 q4 + 4 half + (l31 & 3) of the wave, channels n0 + 4 (l31 >> 2) .. + 3
          const int prow = 8 * q4 + 4 * half + (l31 & 3);
          f32x4 o = {x0, x1, x2, x3};
          if (residual) o += *reinterpret_cast<const f32x4*>(residual + rpix + (size_t)prow * g.ldr);
          if (!(kAblate & 2) || x0 == 1.2345e30f) *reinterpret_cast<f32x4*>(out + opix + (size_t)prow * g.sox) = o;
        }
      }
      for (int r = 0; r < 16; ++r) { acc[r] = 0.f; acc1[r] = 0.f; }
    }
    __syncthreads();               // buffer (s+1)&1 complete, buffer s&1 free
    if (tr_on && s < 64) g_stream_trace_fp32[4 * s + 3] = clock64();
    float* tswap = bufc; bufc = bufn; bufn = tswap;
  }
#undef PIDM_ST_STAGE
#undef PIDM_ST_LOAD_A
#undef PIDM_ST_LOAD_B
#undef PIDM_ST_WRITE_A
#undef PIDM_ST_WRITE_B
}

template <int KC, int NT>
static int launch_conv_t(ConvGeom g, const float* src0, const float* src1, const float* wp, const float* bias,
                         const float* residual, float* out, int sigmoid_last, hipStream_t st) {
  constexpr int KCP = KC + 4, BN = 32 * NT, Q = KC / 4;
  constexpr int AMAX = 5;
  const int T = g.KH * g.KW;
  const size_t a_bytes = (size_t)g.NI * g.IHt * g.IWt * KCP * sizeof(float);
  const size_t b_tap = (size_t)BN * KCP * sizeof(float);
  size_t off = 0;
  const size_t Np = (size_t)packed_np(g.Cout), Kp = (size_t)cdiv(g.Kw, KC) * KC;
  for (int z = 0; z < g.nz; ++z) { g.w_off[z] = (long)off; off += Np * T * Kp; }
  const int tiles_n = cdiv(g.Cout, BN);
  const bool prof = prof_enabled();
  const double flops = 2.0 * g.B * g.Hv * g.Wv * g.nz * (double)g.Cout * g.Kw * T;
  // ---- pipelined kernel when the whole tap set fits one slab and the sources are chunk-aligned ----
  const bool aligned = ((g.ld0 & 3) == 0) && ((g.ld1 & 3) == 0) && (g.Cin % KC == 0) && (g.C0 % KC == 0);
  const int nA = g.NI * g.IHt * g.IWt * Q;
  const size_t lds_pipe = a_bytes + (size_t)T * b_tap;
  if (g.nph > 1 && !aligned) return fail("conv: phased 4x4/s2 path needs 16-byte aligned channel counts");
  const bool khw_ok = (g.KH == 3 && g.KW == 3) || (g.KH == 1 && g.KW == 1) || (g.KH == 2 && g.KW == 2);
  const int amax_eff = (KC == 32 && g.KH == 3) ? 9 : AMAX;
  if (aligned && khw_ok && nA <= amax_eff * 256 && lds_pipe <= 80 * 1024) {
    if (prof) prof_begin_launch(0, flops, st);
    const dim3 grid(g.tiles_m * tiles_n, 1, g.nz);
#define PIDM_LAUNCH_PIPE(KH_, KW_, PH_, PS_, MT_, AMAX_, grid_)                                                                                \
  {                                                                                                                        \
    constexpr int BMAXk = (KH_ * KW_ * BN * Q + 255) / 256;                                                                \
    static bool attr_pipe = false;                                                                                         \
    if (!attr_pipe) {                                                                                                      \
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_igemm_pipe_kernel<KC, NT, AMAX_, BMAXk, KH_, KW_, PH_, PS_, MT_>), \
                                hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);                                    \
      attr_pipe = true;                                                                                                    \
    }                                                                                                                      \
    hipLaunchKernelGGL(HIP_KERNEL_NAME(conv_igemm_pipe_kernel<KC, NT, AMAX_, BMAXk, KH_, KW_, PH_, PS_, MT_>), grid_, dim3(256), \
                       lds_pipe, st, g, sigmoid_last, src0, src1 ? src1 : src0, wp, bias, residual, out);                 \
  }
    if constexpr (NT == 4 || KC == 32) {
      // the permuted 128-channel tile exists for 1x1 convolutions only; the 32-channel chunk for 1x1 and (NT == 1) 3x3
      // convolutions
      if constexpr (KC == 32 && NT == 1) {
        if (g.KH == 3 && g.KW == 3 && g.nph == 1) {
          PIDM_LAUNCH_PIPE(3, 3, false, false, 1, 9, grid)
          if (prof) prof_end_launch(st);
          PIDM_CHECK_LAUNCH("conv_igemm_pipe_kernel");
          return 0;
        }
      }
      if (g.KH != 1 || g.KW != 1 || g.nph != 1) return fail("conv: internal error - 1x1-only tile configuration on a %dx%d conv", g.KH, g.KW);
      PIDM_LAUNCH_PIPE(1, 1, false, false, 1, AMAX, grid)
    } else {
      if (g.KH == 3) {
        // persistent walk when the launch has more tiles than resident workgroup slots: 2 workgroups per CU, each
        // owning tpw consecutive m-tiles of one n-tile
        const char* pe = knob("PIDM_PERSIST_SLOTS");   // experiments / tests: 0 = off, else resident workgroup slots
        const long slots = pe ? atol(pe) : 512;
        const long nwork = (long)g.tiles_m * tiles_n * g.nz;
        bool persist = false;
        if constexpr (KC == 16 && NT == 1) {
          // 256-pixel workgroup tile (MT = 2) when the launch still fills the chip: two accumulator chains per wave,
          // half the barriers and weight staging per MFMA (measured on MI355X: +0..5 % over the persistent 128-pixel
          // walk for 32-channel tiles; with 64-channel tiles it needs 272 registers = one workgroup per CU, so NT == 1 only)
          const char* me = knob("PIDM_MT2_MIN_WGS");   // experiments / tests: 0 = off, else the occupancy gate
          const long mt2_min = me ? atol(me) : 512;
          ConvGeom g2 = g;
          if (mt2_min > 0 && retile_bm(&g2, 256)) {
            const size_t lds2 = (size_t)g2.NI * g2.IHt * g2.IWt * KCP * sizeof(float) + (size_t)T * b_tap;
            if (g2.NI * g2.IHt * g2.IWt * Q <= 8 * 256 && lds2 <= 80 * 1024 && (long)g2.tiles_m * tiles_n >= mt2_min) {
              const ConvGeom g_outer = g;
              {
                const ConvGeom g = g2;
                const size_t lds_pipe = lds2;
                const dim3 grid2(g.tiles_m * tiles_n, 1, 1);
                PIDM_LAUNCH_PIPE(3, 3, false, false, 2, 8, grid2)
              }
              (void)g_outer;
              persist = true;   // launched
            }
          }
        }
        if constexpr (NT == 1) if (!persist) {   // the 64-channel tile needs > 256 registers in the persistent form (1 workgroup per CU)
          if (slots > 0 && nwork > slots) {
            persist = true;
            g.tpw = (int)((nwork + slots - 1) / slots);
            const dim3 gridp(cdiv(g.tiles_m, g.tpw) * tiles_n, 1, g.nz);
            PIDM_LAUNCH_PIPE(3, 3, false, true, 1, AMAX, gridp)
          }
        }
        if (!persist) PIDM_LAUNCH_PIPE(3, 3, false, false, 1, AMAX, grid)
      }
      else if (g.KH == 2 && g.nph > 1) PIDM_LAUNCH_PIPE(2, 2, true, false, 1, AMAX, grid)
      else if (g.KH == 2) PIDM_LAUNCH_PIPE(2, 2, false, false, 1, AMAX, grid)
      else PIDM_LAUNCH_PIPE(1, 1, false, false, 1, AMAX, grid)
    }
#undef PIDM_LAUNCH_PIPE
    if (prof) prof_end_launch(st);
    PIDM_CHECK_LAUNCH("conv_igemm_pipe_kernel");
    return ((g.gn_part || g.bn_part) && NT == 4) ? 1 : 0;   // 1: convolution done, the requested GroupNorm partials were NOT produced
  }
  if (g.nph > 1) return fail("conv: phased 4x4/s2 geometry is not eligible for the pipelined kernel (tile too large)");
  if constexpr (NT == 4 || KC == 32) {
    return fail("conv: internal error - 1x1-only tile configuration selected for an ineligible geometry");
  } else {
  // ---- generic kernel (tap groups, ragged channels, scalar staging) ----
  const size_t budget = 72 * 1024;  // keep two workgroups per CU where the halo tile allows
  int tgs = T;
  if (a_bytes + (size_t)tgs * b_tap > budget) {
    tgs = a_bytes < budget ? (int)((budget - a_bytes) / b_tap) : 1;
    if (tgs < 1) tgs = 1;
  }
  if (tgs > T) tgs = T;
  const size_t lds = a_bytes + (size_t)tgs * b_tap;
  if (lds > 160 * 1024 - 512) return fail("conv: tile needs %zu B of LDS", lds);
  static bool attr_done = false;
  if (!attr_done) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_igemm_kernel<KC, NT>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 512);
    attr_done = true;
  }
  if (prof) prof_begin_launch(0, flops, st);
  hipLaunchKernelGGL(HIP_KERNEL_NAME(conv_igemm_kernel<KC, NT>), dim3(g.tiles_m * tiles_n, 1, g.nz), dim3(256), lds, st, g,
                     tgs, sigmoid_last, src0, src1 ? src1 : src0, wp, bias, residual, out);
  if (prof) prof_end_launch(st);
  PIDM_CHECK_LAUNCH("conv_igemm_kernel");
  return (g.gn_part || g.bn_part) ? 1 : 0;   // the generic kernel has no statistics epilogue
  }
}

// 1x1 convolutions with Cout % 128 == 0 and enough pixel tiles: 128 output channels per workgroup in the permuted
// layout (16-byte stores).  These launches are store-bound (qkv projections: 12.6 MB/sample at 64x64).
bool conv_nt4_ok(const ConvGeom& g, int KC) {
  static int off = -1;
  if (off < 0) { const char* e = knob("PIDM_NO_NT4"); off = (e && atoi(e)) ? 1 : 0; }
  if (off) return false;
  const char* mw = knob("PIDM_NT4_MIN_WGS");   // tests lower the occupancy threshold to reach this path with small shapes
  const long min_wgs = mw ? atol(mw) : 512;
  const bool aligned = ((g.ld0 & 3) == 0) && ((g.ld1 & 3) == 0) && (g.Cin % KC == 0) && (g.C0 % KC == 0);
  return KC == 16 && g.KH == 1 && g.KW == 1 && g.nph == 1 && g.nz == 1 && aligned && (g.Cout % 128 == 0) &&
         g.NI * g.IHt * g.IWt * (KC / 4) <= 5 * 256 && (long)g.tiles_m * (g.Cout / 128) >= min_wgs;
}

// everything launch_conv (k_conv.hip) did not hand to a split-form kernel
int launch_conv_fp32(const ConvGeom& g, const float* src0, const float* src1, const float* wp, const float* bias, const float* residual,
                     float* out, int sigmoid_last, hipStream_t st, int KC, int NT, bool nt4) {
  {
    // streaming persistent 3x3 kernel (PIDM_CONV_STREAM=0: off, for A/B measurements and to reach the older tilings in tests)
    const char* se = knob("PIDM_CONV_STREAM");
    const bool on = !(se && !atoi(se));
    const int npixA = g.NI * g.IHt * g.IWt;
    const size_t lds = (size_t)2 * (npixA + 9 * 32) * 36 * sizeof(float);
    if (on && g.KH == 3 && g.KW == 3 && g.stride == 1 && g.nph == 1 && g.nz == 1 && g.os == 1 && g.soc == 1 && (g.Cin % 32 == 0) &&
        (g.C0 % 32 == 0) && ((g.ld0 | g.ld1) & 3) == 0 && (g.C1 == 0 || g.ld1 == g.ld0) && (g.Cout % 32 == 0) && g.Wv >= 8 &&
        g.Wv == g.Wi && (g.NI * g.IHt * g.Wv) % 32 == 0 && g.NI * g.IHt * g.Wv <= 256 && lds <= 160 * 1024 - 256 &&
        g.pad_y[0] == 1 && g.pad_x[0] == 1 && !sigmoid_last && (g.sox & 3) == 0 && (reinterpret_cast<size_t>(out) & 15) == 0 &&
        (!residual || ((g.ldr & 3) == 0 && (reinterpret_cast<size_t>(residual) & 15) == 0))) {
      const bool prof = prof_enabled();
      static bool attr_s = false;
      if (!attr_s) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3_stream_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 256);
        attr_s = true;
      }
      ConvGeom gs = g;
      gs.w_off[0] = 0;
      const int n_items = g.tiles_m * (g.Cout / 32);
      const char* ce = knob("PIDM_STREAM_WGS");        // persistent workgroups (default: one per CU of an MI355X); read per launch
      int n_cu = ce ? atoi(ce) : 256;                      // (the unit tests lower it to get several work items per workgroup)
      if (n_cu < 1) n_cu = 256;
      const int ipw = cdiv(n_items, n_cu), wgs = cdiv(n_items, ipw);
      if (prof) prof_begin_launch(0, 2.0 * g.B * g.Hv * g.Wv * (double)g.Cout * g.Kw * 9, st);
      hipLaunchKernelGGL(conv3x3_stream_kernel, dim3(wgs), dim3(256), lds, st, gs, sigmoid_last, src0, src1 ? src1 : src0, wp, bias, residual, out,
                         n_items, ipw, knob("PIDM_STREAM_TRACE") ? 1 : 0);
      if (prof) prof_end_launch(st);
      PIDM_CHECK_LAUNCH("conv3x3_stream_kernel");
      return 0;
    }
  }
  if (nt4) {
    // K == 32 (the qkv projections of the 64x64 level, to_out dgrad): one 32-channel chunk = ONE dependent load round per
    // workgroup instead of two (PIDM_KC32=0 disables, for A/B measurements)
    static int kc32 = -1;
    if (kc32 < 0) { const char* e = knob("PIDM_KC32"); kc32 = (e && !atoi(e)) ? 0 : 1; }
    if (kc32 && g.Cin == 32 && g.C0 == 32) return launch_conv_t<32, 4>(g, src0, src1, wp, bias, residual, out, sigmoid_last, st);
    return launch_conv_t<16, 4>(g, src0, src1, wp, bias, residual, out, sigmoid_last, st);
  }
  {
    // 1x1 convolutions with Cin % 32 == 0: 32-channel chunks (half the barriers / dependent load rounds per tile)
    static int kc32 = -1;
    if (kc32 < 0) { const char* e = knob("PIDM_KC32"); kc32 = (e && !atoi(e)) ? 0 : 1; }
    const bool al32 = ((g.ld0 & 3) == 0) && ((g.ld1 & 3) == 0) && (g.Cin % 32 == 0) && (g.C0 % 32 == 0);
    if (kc32 && KC == 16 && g.KH == 1 && g.KW == 1 && g.nph == 1 && al32 && g.NI * g.IHt * g.IWt * 8 <= 5 * 256) {
      if (NT == 2) return launch_conv_t<32, 2>(g, src0, src1, wp, bias, residual, out, sigmoid_last, st);
      return launch_conv_t<32, 1>(g, src0, src1, wp, bias, residual, out, sigmoid_last, st);
    }
  }
  if (KC == 16 && NT == 1 && g.KH == 3 && g.KW == 3 && g.nph == 1 && g.nz == 1 && (g.Cin % 32 == 0) && (g.C0 % 32 == 0) &&
      ((g.ld0 | g.ld1) & 3) == 0 && g.NI * g.IHt * g.IWt * 8 <= 9 * 256 &&
      ((size_t)g.NI * g.IHt * g.IWt + 9 * 32) * 36 * sizeof(float) <= 80 * 1024) {
    // 32-channel chunks for 32-channel-tile 3x3 convolutions: 144 MFMAs per wave between barriers instead of 72 (two
    // workgroups per CU instead of three); +3..14 % on the 8x8 / 64x64 levels, neutral elsewhere (PIDM_KC32_3X3=0: off)
    const char* e = knob("PIDM_KC32_3X3");
    if (!(e && !atoi(e))) return launch_conv_t<32, 1>(g, src0, src1, wp, bias, residual, out, sigmoid_last, st);
  }
  if (KC == 16 && NT == 2) return launch_conv_t<16, 2>(g, src0, src1, wp, bias, residual, out, sigmoid_last, st);
  if (KC == 16 && NT == 1) return launch_conv_t<16, 1>(g, src0, src1, wp, bias, residual, out, sigmoid_last, st);
  if (KC == 8 && NT == 2) return launch_conv_t<8, 2>(g, src0, src1, wp, bias, residual, out, sigmoid_last, st);
  return launch_conv_t<8, 1>(g, src0, src1, wp, bias, residual, out, sigmoid_last, st);
}

}  // namespace pidm
