// Weight gradients of the convolutions and the fixed-order reductions behind them (split out of k_conv.hip in round 4).
//
// Replaces the weight / bias gradient half of every `convolution_backward` / `addmm` backward of the reference UNet
// (src/unet_model.py:163,197,227,253,275,279,453,517 and the nn.Linear layers :248,332,339,466,468, reached through loss.backward(),
// main.py:164):  dW[m][t][n] = sum_p dY[p][m] * X[p (+tap)][n]  (M = dY's channels, N = X's channels), as a deterministic split-K over
// pixel ranges: every kernel writes per-split partial slabs [split][MP][T][NP], one fixed-order reduction sums them (no float
// atomics anywhere: results are bit-identical run to run).
//
// Kernels in this file:
//   conv_wgrad_kernel<MAXT>, conv_wgrad_pipe_kernel<KH,KW,PHASED,WIDE,MINW,ROWST>   fp32-MFMA weight gradients (generic / pipelined)
//   conv_wgrad_split_kernel<P>                       3x3 on the bf16 pipe, LDS-staged (the row-streaming successor: k_wgrad_rs.hip)
//   conv_wgrad_smallc_kernel<MAXN>                   few input channels, many taps (7x7 init conv)
//   conv_wgrad_1x1_stream_kernel / _stream4_kernel   1x1 weight gradients as LDS-free pixel streams (HBM-bound)
//   smallm_splitk_kernel                             few rows x very long K (input gradient of the concatenated FiLM linears)
//   wgrad_reduce_kernel / reduce_multi_kernel        fixed-order split-K sums (immediate / all deferred sums of a backward in ONE launch)
//   colsum_partial_kernel / colsum_final_kernel      bias gradients of transposed convs, LayerNorm gamma
#include <stdio.h>
#include <stdlib.h>

#include "pidm_launch.h"

namespace pidm {

static const int kBM = 128;  // pixels per workgroup tile (as k_conv.hip)
// ---------------------------------------------------------------------------------------------------
// wgrad: dW[m][t][n] = sum_p dY[p][m] * X[p (+tap)][n]   (M = rows of dY's channels, N = X's channels)
// one workgroup = 32 x 32 output block for up to `tgs` taps, over `tiles_per_split` pixel tiles;
// the 4 waves split each 128-pixel tile 4-ways along K and are reduced through LDS at the end.
// ---------------------------------------------------------------------------------------------------
// (struct WgradGeom: pidm_common.h - shared with k_wgrad_rs.hip)

template <int MAXT>
__global__ void __launch_bounds__(256) conv_wgrad_kernel(WgradGeom wg, const float* __restrict__ src0,
                                                         const float* __restrict__ src1, const float* __restrict__ dy,
                                                         float* __restrict__ partial, float* __restrict__ bias_partial) {
  const ConvGeom& g = wg.g;
  HIP_DYNAMIC_SHARED(float, smem)
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, half = lane >> 5, l31 = lane & 31;
  const int T = g.KH * g.KW;
  const int ntn = wg.NP / 32;
  int rest = blockIdx.y;
  const int tg = rest % wg.ntg;
  rest /= wg.ntg;
  const int tn = rest % ntn, tm = rest / ntn;
  const int m0 = tm * 32, n0 = tn * 32;
  const int t0 = tg * wg.tgs;
  const int nt = (T - t0 < wg.tgs) ? (T - t0) : wg.tgs;
  const int split = blockIdx.x;
  const int npixA = g.NI * g.IHt * g.IWt;
  float* Xs = smem;                         // [npixA][32]
  float* Ys = smem + (size_t)npixA * 32;    // [128][32]
  const int tpi = g.Hv / g.TH;
  const bool vec_ok = ((g.ld0 & 3) == 0) && ((g.ld1 & 3) == 0) && ((g.C0 & 3) == 0) && ((g.Cin & 3) == 0);
  const bool vec_dy = ((wg.ld_dy & 3) == 0) && ((g.Cout & 3) == 0);

  f32x16 acc[MAXT];
#pragma unroll
  for (int i = 0; i < MAXT; ++i)
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  // bias gradient = column sums of dY: done by the (tn == 0, tg == 0) blocks from the tile already in LDS
  const bool do_bias = (bias_partial != nullptr) && (tn == 0) && (tg == 0);
  float bacc = 0.f;

  __shared__ int tap_off[64];
  if (tid < T) tap_off[tid] = (tid / g.KW) * g.IWt + (tid % g.KW);
  const int rows = g.NI * g.IHt, rowf4 = g.IWt * 8;
  const int tile_lo = split * wg.tiles_per_split;
  const int tile_hi = (tile_lo + wg.tiles_per_split < g.tiles_m) ? tile_lo + wg.tiles_per_split : g.tiles_m;
  for (int tile = tile_lo; tile < tile_hi; ++tile) {
    const int b0 = (tile / tpi) * g.NI;
    const int vy0 = (tile % tpi) * g.TH;
    const int iy0 = vy0 * g.stride - g.pad_y[0], ix0 = -g.pad_x[0];
    __syncthreads();
    for (int rrow = wave; rrow < rows; rrow += 4) {
      const int img = rrow / g.IHt, hy = rrow - img * g.IHt;
      const int b = b0 + img, iy = iy0 + hy;
      const bool rowvalid = (b < g.B) && (iy >= 0) && (iy < g.Hi);
      const size_t rowpix = ((size_t)b * g.Hi + iy) * g.Wi;
      float* xrow_s = Xs + (size_t)rrow * g.IWt * 32;
      for (int e = lane; e < rowf4; e += 64) {
        const int hx = e >> 3, q = e & 7;
        const int ix = ix0 + hx, c = n0 + 4 * q;
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
        *reinterpret_cast<float4*>(xrow_s + (size_t)hx * 32 + 4 * q) = v;
      }
    }
    for (int e = tid; e < kBM * 8; e += 256) {
      const int q = e & 7, p = e >> 3;
      const int tx = p & (g.Wv - 1), ty = (p >> g.wsh) & (g.TH - 1), img = p >> (g.wsh + g.tsh);
      const int b = b0 + img, c = m0 + 4 * q;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (b < g.B && img < g.NI && c < g.Cout) {
        const size_t pix = ((size_t)b * g.Hv + (vy0 + ty)) * g.Wv + tx;
        if (vec_dy) {
          v = *reinterpret_cast<const float4*>(dy + pix * wg.ld_dy + c);
        } else {
          float t4[4];
          for (int k = 0; k < 4; ++k) t4[k] = (c + k < g.Cout) ? dy[pix * wg.ld_dy + c + k] : 0.f;
          v = make_float4(t4[0], t4[1], t4[2], t4[3]);
        }
      }
      *reinterpret_cast<float4*>(Ys + (size_t)p * 32 + 4 * q) = v;
    }
    __syncthreads();
    if (do_bias) {
      const int o = tid & 31, part = tid >> 5;
#pragma unroll
      for (int k = 0; k < 16; ++k) bacc += Ys[(part * 16 + k) * 32 + o];
    }
    for (int ks = 0; ks < 16; ++ks) {
      const int p = wave * 32 + 2 * ks + half;
      const int tx = p & (g.Wv - 1), ty = (p >> g.wsh) & (g.TH - 1), img = p >> (g.wsh + g.tsh);
      const int xb = (img < g.NI) ? (img * g.IHt + ty * g.stride) * g.IWt + tx * g.stride : 0;
      const float a = Ys[p * 32 + l31];
#pragma unroll
      for (int tl = 0; tl < MAXT; ++tl) {
        if (tl < nt) {
          const float bv = Xs[(size_t)(xb + tap_off[t0 + tl]) * 32 + l31];
          acc[tl] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, bv, acc[tl], 0, 0, 0);
        }
      }
    }
  }
  // ---- cross-wave reduction, one tap at a time through LDS (re-using the staging area) ----
  float* red = smem;  // [4][1024]
#pragma unroll
  for (int tl = 0; tl < MAXT; ++tl) {
    if (tl < nt) {
      __syncthreads();
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = (r & 3) + 8 * (r >> 2) + 4 * half;
        red[wave * 1024 + row * 32 + l31] = acc[tl][r];
      }
      __syncthreads();
      for (int e = tid; e < 1024; e += 256) {
        const float s = (red[e] + red[1024 + e]) + (red[2048 + e] + red[3072 + e]);
        const int row = e >> 5, col = e & 31;
        partial[(((size_t)split * wg.MP + (m0 + row)) * T + (t0 + tl)) * wg.NP + n0 + col] = s;
      }
    }
  }
  if (do_bias) {
    __syncthreads();
    red[tid] = bacc;
    __syncthreads();
    if (tid < 32) {
      float sb = 0.f;
      for (int k = 0; k < 8; ++k) sb += red[k * 32 + tid];
      bias_partial[(size_t)split * wg.MP + m0 + tid] = sb;
    }
  }
}

// software-pipelined wgrad (3x3 / 1x1, 16-byte aligned operands): the (halo pixel, quad) decode is done once, the
// next pixel tile is prefetched into registers while the 16 x nt MFMAs per wave of the current one run.
// second launch-bound argument = minimum waves per SIMD: it caps the register allocation (512 / n) so that 2 (3x3: 144
// accumulators) resp. 4 (1x1) workgroups share a CU and hide each other's barrier / staging phases
// WIDE: the wave's 32 pixels are consecutive in x inside one image row (Wv >= 32): pointer bumps instead of a per-step
// decode.  A compile-time switch: as a runtime branch the two loops made the register allocator copy all accumulators.
// ROWST (3x3, stride 1, full-width tiles): the staging path without vector arithmetic of conv3x3_stream_kernel - halo columns
// zeroed once, whole image rows staged with wave-uniform validity and scalar row bases, per-thread constant offsets.  The fp32
// MFMA shares the SIMD's vector ALUs, so the ~350 VALU instructions of the per-tile slot decode cost a quarter of a tile's 144
// MFMAs (measured: 75 TFLOP/s on the 3x3 weight gradients before).
template <int KH, int KW, bool PHASED, bool WIDE, int MINW, bool ROWST = false>
__global__ void __launch_bounds__(256, MINW) conv_wgrad_pipe_kernel(WgradGeom wg, const float* __restrict__ src0,
                                                              const float* __restrict__ src1, const float* __restrict__ dy,
                                                              float* __restrict__ partial, float* __restrict__ bias_partial) {
  constexpr int XMAX = (KH * KW == 1 && !PHASED) ? 4 : 9, YMAX = 4, T_ = KH * KW;   // 1x1: the tile has no halo (128 pixels)
  constexpr int MAXT = T_;   // all taps of a 32x32 (dY-channel x X-channel) tile live in this wave's accumulators; the 4 waves
                             // split the 128-pixel tile (K-split) and are reduced through LDS at the end
  const ConvGeom& g = wg.g;
  HIP_DYNAMIC_SHARED(float, smem)
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, half = lane >> 5, l31 = lane & 31;
  constexpr int T = PHASED ? 16 : KH * KW;   // taps of the partial-buffer layout (4x4 source taps when phased)
  const int ph = PHASED ? blockIdx.z : 0;
  const int ntn = wg.NP / 32;
  const int tn = blockIdx.y % ntn, tm = blockIdx.y / ntn;   // one tap group (all KHxKW taps)
  const int m0 = tm * 32, n0 = tn * 32;
  const int split = blockIdx.x;
  const int npixA = g.NI * g.IHt * g.IWt;
  float* Xs = smem;
  float* Ys = smem + (size_t)npixA * 32;
  const int tpi = g.Hv / g.TH;

  // ---- prologue: per-thread staging slots.  quad q = tid & 7 is the same for every slot ----
  const int q = tid & 7;
  const int cx = n0 + 4 * q;                       // X channel of this thread
  const bool cx_ok = cx < g.Cin;
  const float* xsrc = (cx < g.C0) ? src0 + cx : src1 + (cx - g.C0);
  const int xld = (cx < g.C0) ? g.ld0 : g.ld1;
  const int cy = m0 + 4 * q;
  const bool cy_ok = cy < g.Cout;
  // the (halo pixel) -> (image, row, column) decode of a staging slot is recomputed per tile (a few VALU ops against
  // 144 MFMAs) instead of being kept in registers: the accumulators leave no room for it at 2 waves per SIMD
  f32x4 rx[XMAX], ry[YMAX];
  // ---- ROWST state: per-thread constants and wave-uniform row descriptors ----
  int rs_lds[8];
  unsigned rs_xvo[2], rs_yvo = 0, rs_xmask = 0, rs_ymask = 0;
  int rs_img[8], rs_hy[8], rs_yimg[4];
  int rs_AS = 0;
  if constexpr (ROWST) {
    rs_AS = (g.NI * g.IHt * g.Wv) >> 5;
    const int wv8 = __builtin_amdgcn_readfirstlane(wave) * 8;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const int sp = (tid >> 3) + 32 * k;
      const int sr = sp >> g.wsh, x = sp & (g.Wv - 1);
      const int img = fast_div(sr, g.IHt, g.mIHt), hy = sr - img * g.IHt;
      rs_lds[k] = (k < rs_AS) ? ((img * g.IHt + hy) * g.IWt + x + 1) * 32 + 4 * q : -1;
      if (k < 2) rs_xvo[k] = (unsigned)(x * xld + 4 * q) * 4u;
      const int srw = (wv8 + 32 * k) >> g.wsh;
      rs_img[k] = fast_div(srw, g.IHt, g.mIHt);
      rs_hy[k] = srw - rs_img[k] * g.IHt;
    }
    rs_yvo = (unsigned)((tid >> 3) * wg.ld_dy + cy) * 4u;
#pragma unroll
    for (int k = 0; k < 4; ++k) rs_yimg[k] = (wv8 + 32 * k) >> (g.wsh + g.tsh);
    // halo columns of the X tile: zero for every tile, written once
    for (int e = tid; e < g.NI * g.IHt * 2 * 8; e += 256) {
      const int qq = e & 7, side = (e >> 3) & 1, row = e >> 4;
      *reinterpret_cast<f32x4*>(Xs + (size_t)(row * g.IWt + (side ? g.IWt - 1 : 0)) * 32 + 4 * qq) = f32x4{0.f, 0.f, 0.f, 0.f};
    }
  }
  const char* rs_xsrc = reinterpret_cast<const char*>((n0 < g.C0) ? src0 + n0 : src1 + (n0 - g.C0));   // n-tile = one source (C0 % 32 == 0)
#define PIDM_WG_PREFETCH_ROWS(tile_)                                                                              \
  {                                                                                                               \
    const int tile__ = (tile_);                                                                                   \
    const int b0__ = (tile__ / tpi) * g.NI, vy0__ = (tile__ % tpi) * g.TH;                                        \
    _Pragma("unroll") for (int k = 0; k < 8; ++k) {                                                               \
      const int b__ = b0__ + rs_img[k], iy__ = vy0__ - g.pad_y[0] + rs_hy[k];                                     \
      const bool ok__ = (b__ < g.B) & (iy__ >= 0) & (iy__ < g.Hi) & cx_ok;                                         \
      const size_t row__ = ok__ ? (size_t)(b__ * g.Hi + iy__) * g.Wi : 0;                                         \
      rx[k] = *reinterpret_cast<const f32x4*>(rs_xsrc + row__ * (size_t)xld * 4 + rs_xvo[k & 1]);                 \
      rs_xmask = (rs_xmask & ~(1u << k)) | ((ok__ ? 1u : 0u) << k);                                               \
    }                                                                                                             \
    const char* yb__ = reinterpret_cast<const char*>(dy) + (((size_t)b0__ * g.Hv + vy0__) * g.Wv) * (size_t)wg.ld_dy * 4; \
    _Pragma("unroll") for (int k = 0; k < YMAX; ++k) {                                                            \
      const bool ok__ = (b0__ + rs_yimg[k] < g.B) & (rs_yimg[k] < g.NI) & cy_ok;                                   \
      ry[k] = *reinterpret_cast<const f32x4*>(yb__ + (ok__ ? (size_t)k * 32 * wg.ld_dy * 4 + rs_yvo : (size_t)0)); \
      rs_ymask = (rs_ymask & ~(1u << k)) | ((ok__ ? 1u : 0u) << k);                                               \
    }                                                                                                             \
  }

#define PIDM_WG_PREFETCH(tile_)                                                                                   \
  {                                                                                                               \
    const int tile__ = (tile_);                                                                                   \
    const int b0__ = (tile__ / tpi) * g.NI, vy0__ = (tile__ % tpi) * g.TH;                                        \
    const int iy0__ = PHASED ? vy0__ - g.ph_pad_y[ph] : vy0__ * g.stride - g.pad_y[0];                            \
    const int ix0__ = PHASED ? -g.ph_pad_x[ph] : -g.pad_x[0];                                                      \
    _Pragma("unroll") for (int k = 0; k < XMAX; ++k) {                                                            \
      rx[k] = f32x4{0.f, 0.f, 0.f, 0.f};                                                                          \
      const int hp = (tid + k * 256) >> 3;                                                                        \
      if (hp < npixA && cx_ok) {                                                                                  \
        const int hrow = fast_div(hp, g.IWt, g.mIWt), hx = hp - hrow * g.IWt;                                     \
        const int img = fast_div(hrow, g.IHt, g.mIHt), hy = hrow - img * g.IHt;                                   \
        const int b = b0__ + img;                                                                                 \
        int iy = iy0__ + hy, ix = ix0__ + hx;                                                                     \
        if (PHASED) { iy = iy * g.in_step + g.ph_oy[ph]; ix = ix * g.in_step + g.ph_ox[ph]; }                      \
        if (b < g.B && iy >= 0 && iy < g.Hi && ix >= 0 && ix < g.Wi)                                               \
          rx[k] = *reinterpret_cast<const f32x4*>(xsrc + (((size_t)b * g.Hi + iy) * g.Wi + ix) * xld);            \
      }                                                                                                           \
    }                                                                                                             \
    _Pragma("unroll") for (int k = 0; k < YMAX; ++k) {                                                            \
      const int p = (tid + k * 256) >> 3;                                                                         \
      const int tx = p & (g.Wv - 1), ty = (p >> g.wsh) & (g.TH - 1), img = p >> (g.wsh + g.tsh);                  \
      const int b = b0__ + img;                                                                                   \
      ry[k] = f32x4{0.f, 0.f, 0.f, 0.f};                                                                          \
      if (cy_ok && b < g.B && img < g.NI)                                                                         \
        ry[k] = *reinterpret_cast<const f32x4*>(dy + (((size_t)b * g.Hv + (vy0__ + ty)) * g.Wv + tx) * wg.ld_dy + cy); \
    }                                                                                                             \
  }

  f32x16 acc[MAXT];
#pragma unroll
  for (int i = 0; i < MAXT; ++i)
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  const bool do_bias = (bias_partial != nullptr) && (tn == 0) && (ph == 0);
  float bacc = 0.f;

  const int tile_lo = split * wg.tiles_per_split;
  const int tile_hi = (tile_lo + wg.tiles_per_split < g.tiles_m) ? tile_lo + wg.tiles_per_split : g.tiles_m;
  if (tile_lo < tile_hi) {
    if constexpr (ROWST) PIDM_WG_PREFETCH_ROWS(tile_lo)
    else PIDM_WG_PREFETCH(tile_lo)
  }
  for (int tile = tile_lo; tile < tile_hi; ++tile) {
    __syncthreads();
    if constexpr (ROWST) {
#pragma unroll
      for (int k = 0; k < 8; ++k)
        if (k < rs_AS) *reinterpret_cast<f32x4*>(Xs + rs_lds[k]) = rx[k] * (((rs_xmask >> k) & 1u) ? 1.f : 0.f);
#pragma unroll
      for (int k = 0; k < YMAX; ++k)
        *reinterpret_cast<f32x4*>(Ys + (size_t)((tid >> 3) + 32 * k) * 32 + 4 * q) = ry[k] * (((rs_ymask >> k) & 1u) ? 1.f : 0.f);
    } else {
#pragma unroll
      for (int k = 0; k < XMAX; ++k) {
        const int hp = (tid + k * 256) >> 3;
        if (hp < npixA) *reinterpret_cast<f32x4*>(Xs + (size_t)hp * 32 + 4 * q) = rx[k];
      }
#pragma unroll
      for (int k = 0; k < YMAX; ++k) {
        const int p = (tid + k * 256) >> 3;
        *reinterpret_cast<f32x4*>(Ys + (size_t)p * 32 + 4 * q) = ry[k];
      }
    }
    __syncthreads();
    if (tile + 1 < tile_hi) {
      if constexpr (ROWST) PIDM_WG_PREFETCH_ROWS(tile + 1)
      else PIDM_WG_PREFETCH(tile + 1)
    }
    if (do_bias) {
      const int o = tid & 31, part = tid >> 5;
#pragma unroll
      for (int k = 0; k < 16; ++k) bacc += Ys[(part * 16 + k) * 32 + o];
    }
    // k-loop over the wave's 32 pixels (16 MFMA k-steps of 2 pixels), explicitly double buffered: the T+1 LDS fragments of
    // step ks+1 are fetched into the second register set BEFORE the T MFMAs of step ks issue, pinned with
    // sched_group_barrier - otherwise hipcc re-uses one register triple and every 3 MFMAs wait for a full LDS round trip
    // (1 wave per SIMD at 144 accumulators: nothing else hides it; measured 44 % MFMA busy).
    {
      float fa_[2], fb_[2][T_];
      int xb_w = 0;
      const float* ap = nullptr;
      const float* xp = nullptr;
      if constexpr (WIDE) {
        const int p0 = wave * 32;
        const int tx0 = p0 & (g.Wv - 1), ty0 = (p0 >> g.wsh) & (g.TH - 1), img0 = p0 >> (g.wsh + g.tsh);
        xb_w = (img0 < g.NI) ? (img0 * g.IHt + ty0 * g.stride) * g.IWt + tx0 * g.stride : 0;
        ap = Ys + (size_t)(p0 + half) * 32 + l31;
        xp = Xs + (size_t)(xb_w + half * g.stride) * 32 + l31;
      }
      const int xstep = 2 * g.stride * 32;
#define PIDM_WG_FRAGS(set_, ks_)                                                                                   \
  {                                                                                                                \
    const float* xrow__;                                                                                           \
    if constexpr (WIDE) {                                                                                          \
      fa_[set_] = ap[(ks_)*64];                                                                                    \
      xrow__ = xp + (size_t)(ks_)*xstep;                                                                           \
    } else {                                                                                                       \
      const int p__ = wave * 32 + 2 * (ks_) + half;                                                                \
      const int tx__ = p__ & (g.Wv - 1), ty__ = (p__ >> g.wsh) & (g.TH - 1), img__ = p__ >> (g.wsh + g.tsh);       \
      const int xb__ = (img__ < g.NI) ? (img__ * g.IHt + ty__ * g.stride) * g.IWt + tx__ * g.stride : 0;           \
      fa_[set_] = Ys[p__ * 32 + l31];                                                                              \
      xrow__ = Xs + (size_t)xb__ * 32 + l31;                                                                       \
    }                                                                                                              \
    _Pragma("unroll") for (int t__ = 0; t__ < T_; ++t__)                                                           \
        fb_[set_][t__] = xrow__[((t__ / KW) * g.IWt + (t__ % KW)) * 32];                                           \
  }
      PIDM_WG_FRAGS(0, 0)
#pragma unroll
      for (int ks = 0; ks < 16; ++ks) {
        const int cur = ks & 1;
        if (ks + 1 < 16) PIDM_WG_FRAGS(cur ^ 1, ks + 1)
#pragma unroll
        for (int t = 0; t < T_; ++t)
          acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa_[cur], fb_[cur][t], acc[t], 0, 0, 0);
        // order: the DS reads of step ks+1 (grouped first), then the T MFMAs of step ks
        if (ks + 1 < 16) __builtin_amdgcn_sched_group_barrier(0x100, T_ + 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, T_, 0);
      }
#undef PIDM_WG_FRAGS
    }
  }
#undef PIDM_WG_PREFETCH
#undef PIDM_WG_PREFETCH_ROWS
  float* red = smem;  // [4][1024]
#pragma unroll
  for (int tl = 0; tl < MAXT; ++tl) {
    {
      __syncthreads();
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = (r & 3) + 8 * (r >> 2) + 4 * half;
        red[wave * 1024 + row * 32 + l31] = acc[tl][r];
      }
      __syncthreads();
      for (int e = tid; e < 1024; e += 256) {
        const float sv = (red[e] + red[1024 + e]) + (red[2048 + e] + red[3072 + e]);
        const int row = e >> 5, col = e & 31;
        int tdst = tl;
        if (PHASED) {   // local tap (jy, jx) of phase (py, px) -> source tap (ky, kx) of the 4x4 kernel
          const int py = ph >> 1, px = ph & 1, jy = tl >> 1, jx = tl & 1;
          tdst = (py == 0 ? 1 + 2 * jy : 2 * jy) * 4 + (px == 0 ? 1 + 2 * jx : 2 * jx);
        }
        partial[(((size_t)split * wg.MP + (m0 + row)) * T + tdst) * wg.NP + n0 + col] = sv;
      }
    }
  }
  if (do_bias) {
    __syncthreads();
    red[tid] = bacc;
    __syncthreads();
    if (tid < 32) {
      float sb = 0.f;
      for (int k = 0; k < 8; ++k) sb += red[k * 32 + tid];
      bias_partial[(size_t)split * wg.MP + m0 + tid] = sb;
    }
  }
}

// ---------------------------------------------------------------------------------------------------
// 3x3 / stride-1 weight gradient in the split form (bf16 matrix pipe, fp32-faithful 3-piece operands; pidm_common.h).
//   dW[m][tap][n] = sum_p dY[p][m] X[p + tap][n]: the contraction runs over PIXELS, so the 32x32x16 MFMA wants, per lane, 8
//   consecutive pixels of one channel in 4 registers - the transpose of the channels-last tensors.  Staging does it in registers:
//   a thread loads 4 channels of one pixel, v_permlane32_swap / v_permlane16_swap (2 + 2 instructions) turn 4 lanes x 4
//   registers into 4 pixels of one channel, the values are split into their 3 bf16 pieces and land in LDS as [piece][channel][pixel].
//   Rows of X are stored with one halo element on either side of an 8-element-aligned body, so the centre tap (kx = 1) is an
//   aligned 16-byte read and the taps kx = 0 / 2 are the same registers moved by one element (v_alignbit with the dword before /
//   after: 5 VALU per piece and k-step for both).
// Workgroup = 12 waves = 3 kernel rows (ky) x 4 pixel quarters of a P-pixel tile: a wave owns the 3 accumulators (kx) of its ky
// for a 32 (dY channels) x 32 (X channels) block and walks its quarter of every tile of the split; the quarters are summed
// through LDS at the end and the result goes to the split's partial slab like the other wgrad kernels (fixed-order reduction).
// ---------------------------------------------------------------------------------------------------
__device__ __forceinline__ int wgs_xrow_bytes(const ConvGeom& g) {
  int xr = ((g.NI * g.IHt * (g.Wv + 8) + 8) * 2 + 15) & ~15;
  if (((xr >> 4) & 1) == 0) xr += 16;       // odd number of 16-byte slots: conflict-free ds_read_b128 across channels
  return xr;
}
template <int P>
__global__ void __launch_bounds__(768) conv_wgrad_split_kernel(WgradGeom wg, const float* __restrict__ src0, const float* __restrict__ src1,
                                                               const float* __restrict__ dy, float* __restrict__ partial,
                                                               float* __restrict__ bias_partial) {
  constexpr int NQ = 4, KS = P / (16 * NQ), NSLOT = (P == 256) ? 7 : 4, NWV = 12;
  const ConvGeom& g = wg.g;
  HIP_DYNAMIC_SHARED(float, smemf)
  char* smem = reinterpret_cast<char*>(smemf);
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int half = lane >> 5, l31 = lane & 31;
  const int ntn = wg.NP / 32;
  const int tn = blockIdx.y % ntn, tm = blockIdx.y / ntn;
  const int m0 = tm * 32, n0 = tn * 32;
  const int split = blockIdx.x;
  const int RW = g.Wv + 8;
  const int XROW = wgs_xrow_bytes(g), YROW = P * 2 + 16;
  char* Xs = smem;
  const int tpi = g.Hv / g.TH;
  const int SEG8 = (g.NI * g.IHt * g.Wv) >> 3;       // 8-pixel groups of the X halo tile (whole rows), then P / 8 groups of dY

  // ---- staging slots: wave-slot j = wave + 12 k is an 8-pixel x 32-channel group of X (j < SEG8) or of dY ----
  // Channel c = 4 q + la of a tile lives in LDS row q + 8 la: the 16 lanes a ds_write_b64 serves per cycle have one la and all 8 q,
  // i.e. 8 consecutive rows (rows of 16 x odd bytes: every bank once), where rows 4 q + la would put them on two bank groups
  // (4-way conflict).  The MFMA operands are read by LDS row, so the accumulator's row / column r stands for channel
  // 4 (r & 7) + (r >> 3); the final write to the partial slab undoes the permutation.
  const int q = lane & 7, pb = (lane >> 3) & 1, la = lane >> 4;      // channel quad, pixel bit 2, pixel bits 0-1
  const char* xbase = reinterpret_cast<const char*>((n0 < g.C0) ? src0 + n0 : src1 + (n0 - g.C0));   // n-tile = one source (C0 % 32 == 0)
  const char* ybase = reinterpret_cast<const char*>(dy + m0);
  int s_kind[NSLOT], s_img[NSLOT], s_row[NSLOT];     // wave-uniform: 0 none, 1 X, 2 dY; image and (halo / tile) row of the group
  int lds_off[NSLOT];
  unsigned g_vo[NSLOT];
#pragma unroll
  for (int k = 0; k < NSLOT; ++k) {
    const int j = wave + NWV * k;
    if (j < SEG8) {
      const int sp = 8 * j, sr = sp >> g.wsh, x0 = sp & (g.Wv - 1);
      s_kind[k] = 1;
      s_img[k] = fast_div(sr, g.IHt, g.mIHt);
      s_row[k] = sr - s_img[k] * g.IHt;
      lds_off[k] = (q + 8 * la) * XROW + (sr * RW + 8 + x0 + 4 * pb) * 2;
      g_vo[k] = (unsigned)((x0 + 4 * pb + la) * g.ld0 + 4 * q) * 4u;
    } else if (j < SEG8 + P / 8) {
      const int p0 = 8 * (j - SEG8);
      s_kind[k] = 2;
      s_img[k] = p0 >> (g.wsh + g.tsh);
      s_row[k] = (p0 >> g.wsh) & (g.TH - 1);
      lds_off[k] = 96 * XROW + (q + 8 * la) * YROW + (p0 + 4 * pb) * 2;
      g_vo[k] = (unsigned)(((p0 & (g.Wv - 1)) + 4 * pb + la) * wg.ld_dy + 4 * q) * 4u;
    } else {
      s_kind[k] = 0; s_img[k] = 0; s_row[k] = 0; lds_off[k] = 0; g_vo[k] = 0;
    }
  }
  // X region zeroed once: the halo elements of every row stay zero, the bodies are rewritten per tile
  for (int e = tid; e < (96 * XROW) >> 4; e += 768) reinterpret_cast<u32x4*>(Xs)[e] = u32x4{0u, 0u, 0u, 0u};

  // ---- fragment offsets of this wave: kernel row ky, pixel quarter pq; per k-step the lane's 8 pixels (one row segment) ----
  const int ky = wave >> 2, pq = wave & 3;
  int xo[KS], yo[KS];
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) {
    const int p = pq * (P / NQ) + 16 * ks + 8 * half;
    const int tx = p & (g.Wv - 1), ty = (p >> g.wsh) & (g.TH - 1), img = p >> (g.wsh + g.tsh);
    xo[ks] = l31 * XROW + (((img * g.IHt + ty + ky) * RW) + 8 + tx) * 2;
    yo[ks] = 96 * XROW + l31 * YROW + p * 2;
  }

  f32x4 rr[NSLOT];
  unsigned keep = 0;           // wave-uniform: bit k = slot k of the loads in flight is inside the batch / image
  // (image group, row tile) of the tile being fetched, advanced incrementally: no integer division in the tile loop
  int pf_bi = 0, pf_rt = 0;
#define PIDM_WS_PREFETCH()                                                                                         \
  {                                                                                                                \
    const int b0__ = pf_bi * g.NI, vy0__ = pf_rt * g.TH;                                                           \
    _Pragma("unroll") for (int k = 0; k < NSLOT; ++k) {                                                            \
      const int b__ = b0__ + s_img[k];                                                                             \
      const int iy__ = vy0__ + s_row[k] - (s_kind[k] == 1 ? 1 : 0);                                                \
      const bool ok__ = (s_kind[k] != 0) & (b__ < g.B) & (iy__ >= 0) & (iy__ < g.Hi);                              \
      const size_t row__ = ok__ ? (size_t)(b__ * g.Hi + iy__) * g.Wi : 0;                                          \
      const char* base__ = (s_kind[k] == 2) ? ybase + row__ * (size_t)wg.ld_dy * 4 : xbase + row__ * (size_t)g.ld0 * 4; \
      rr[k] = *reinterpret_cast<const f32x4*>(base__ + g_vo[k]);                                                   \
      keep = (keep & ~(1u << k)) | ((ok__ ? 1u : 0u) << k);                                                        \
    }                                                                                                              \
  }

  f32x16 acc[3];
#pragma unroll
  for (int i = 0; i < 3; ++i)
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  const bool do_bias = (bias_partial != nullptr) && (tn == 0);
  float bacc = 0.f;

  const int tile_lo = split * wg.tiles_per_split;
  const int tile_hi = (tile_lo + wg.tiles_per_split < g.tiles_m) ? tile_lo + wg.tiles_per_split : g.tiles_m;
  pf_bi = tile_lo / tpi;
  pf_rt = tile_lo - pf_bi * tpi;
  if (tile_lo < tile_hi) PIDM_WS_PREFETCH()
  for (int tile = tile_lo; tile < tile_hi; ++tile) {
    __syncthreads();
    // ---- registers -> LDS: 4x4 transpose across lane bits 4-5, split, three 8-byte stores ----
#pragma unroll
    for (int k = 0; k < NSLOT; ++k) {
      if (s_kind[k] != 0) {       // wave-uniform
        const f32x4 v = rr[k] * (((keep >> k) & 1u) ? 1.f : 0.f);
        unsigned t0 = __float_as_uint(v[0]), t1 = __float_as_uint(v[1]), t2 = __float_as_uint(v[2]), t3 = __float_as_uint(v[3]);
        {
          const auto s02 = __builtin_amdgcn_permlane32_swap(t0, t2, false, false);
          const auto s13 = __builtin_amdgcn_permlane32_swap(t1, t3, false, false);
          const auto s01 = __builtin_amdgcn_permlane16_swap(s02[0], s13[0], false, false);
          const auto s23 = __builtin_amdgcn_permlane16_swap(s02[1], s13[1], false, false);
          t0 = s01[0]; t1 = s01[1]; t2 = s23[0]; t3 = s23[1];
        }
        // now: channel 4 q + la, pixels 4 pb + 0..3 of the group
        const float f0 = __uint_as_float(t0), f1 = __uint_as_float(t1), f2 = __uint_as_float(t2), f3 = __uint_as_float(t3);
        if (do_bias && s_kind[k] == 2) bacc += (f0 + f1) + (f2 + f3);
        unsigned a0, a1, a2, b0, b1, b2;
        pidm_split3_pk(f0, f1, a0, a1, a2);
        pidm_split3_pk(f2, f3, b0, b1, b2);
        const int ps = 32 * (s_kind[k] == 1 ? XROW : YROW);
        typedef unsigned u32x2_t __attribute__((ext_vector_type(2)));
        char* d = smem + lds_off[k];
        *reinterpret_cast<u32x2_t*>(d) = u32x2_t{a0, b0};
        *reinterpret_cast<u32x2_t*>(d + ps) = u32x2_t{a1, b1};
        *reinterpret_cast<u32x2_t*>(d + 2 * ps) = u32x2_t{a2, b2};
      }
    }
    __syncthreads();
    if (tile + 1 < tile_hi) {
      if (++pf_rt == tpi) { pf_rt = 0; ++pf_bi; }
      PIDM_WS_PREFETCH()
    }
    // ---- the wave's k-steps: 3 dY fragments, 3 x (aligned X chunk + dword before + dword after), 18 MFMAs ----
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      u32x4 ya[3], xc[3], xl[3], xr[3];
#pragma unroll
      for (int pc = 0; pc < 3; ++pc) {
        ya[pc] = *reinterpret_cast<const u32x4*>(smem + yo[ks] + pc * 32 * YROW);
        const char* xp = smem + xo[ks] + pc * 32 * XROW;
        xc[pc] = *reinterpret_cast<const u32x4*>(xp);
        const unsigned prev = *reinterpret_cast<const unsigned*>(xp - 4), next = *reinterpret_cast<const unsigned*>(xp + 16);
        const unsigned m01 = __builtin_amdgcn_alignbit(xc[pc][1], xc[pc][0], 16), m12 = __builtin_amdgcn_alignbit(xc[pc][2], xc[pc][1], 16),
                       m23 = __builtin_amdgcn_alignbit(xc[pc][3], xc[pc][2], 16);
        xl[pc] = u32x4{__builtin_amdgcn_alignbit(xc[pc][0], prev, 16), m01, m12, m23};
        xr[pc] = u32x4{m01, m12, m23, __builtin_amdgcn_alignbit(next, xc[pc][3], 16)};
      }
#define PIDM_WS_SIX(acc_, xb_)                                                                                     \
  acc_ = pidm_mfma_bf16_32x32x16(ya[2], xb_[0], acc_);                                                             \
  acc_ = pidm_mfma_bf16_32x32x16(ya[0], xb_[2], acc_);                                                             \
  acc_ = pidm_mfma_bf16_32x32x16(ya[1], xb_[1], acc_);                                                             \
  acc_ = pidm_mfma_bf16_32x32x16(ya[1], xb_[0], acc_);                                                             \
  acc_ = pidm_mfma_bf16_32x32x16(ya[0], xb_[1], acc_);                                                             \
  acc_ = pidm_mfma_bf16_32x32x16(ya[0], xb_[0], acc_);
      PIDM_WS_SIX(acc[0], xl)
      PIDM_WS_SIX(acc[1], xc)
      PIDM_WS_SIX(acc[2], xr)
#undef PIDM_WS_SIX
    }
  }
#undef PIDM_WS_PREFETCH
  // ---- sum of the 4 pixel quarters through LDS, then the split's partial slab ----
  __syncthreads();
  // The channel permutation of the staging (row / column r = channel 4 (r & 7) + (r >> 3)) is undone on the way INTO LDS, so the
  // sums are read as 16-byte vectors of four consecutive channels and leave as 16-byte stores (3 instead of 12 rounds per thread)
  float* red = smemf;      // [12 waves][3 kx][32 dY channels][32 X channels]
  const int colp = 4 * (l31 & 7) + (l31 >> 3);
#pragma unroll
  for (int kx = 0; kx < 3; ++kx)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = (r & 3) + 8 * (r >> 2) + 4 * half;
      red[(wave * 3 + kx) * 1024 + (4 * (row & 7) + (row >> 3)) * 32 + colp] = acc[kx][r];
    }
  __syncthreads();
  for (int e = tid; e < 9 * 256; e += 768) {
    const int tap = e >> 8, q4 = e & 255, kyo = tap / 3, kxo = tap - 3 * kyo;
    const int mrow = q4 >> 3, c4 = (q4 & 7) * 4;
    const float* rp = red + ((kyo * 4) * 3 + kxo) * 1024 + mrow * 32 + c4;
    const f32x4 a0 = *reinterpret_cast<const f32x4*>(rp), a1 = *reinterpret_cast<const f32x4*>(rp + 3 * 1024);
    const f32x4 a2 = *reinterpret_cast<const f32x4*>(rp + 6 * 1024), a3 = *reinterpret_cast<const f32x4*>(rp + 9 * 1024);
    const f32x4 sv = (a0 + a1) + (a2 + a3);
    *reinterpret_cast<f32x4*>(partial + (((size_t)split * wg.MP + (m0 + mrow)) * 9 + tap) * wg.NP + n0 + c4) = sv;
  }
  if (do_bias) {
    __syncthreads();
    red[tid] = bacc;
    __syncthreads();
    if (tid < 32) {        // channel c = 4 q + la: lanes (q, pb, la) of every wave
      float sb = 0.f;
      const int cq = tid >> 2, ca = tid & 3;
      for (int w = 0; w < 12; ++w)
        for (int b = 0; b < 2; ++b) sb += red[w * 64 + ca * 16 + b * 8 + cq];
      bias_partial[(size_t)split * wg.MP + m0 + tid] = sb;
    }
  }
}

// wgrad for convolutions with very few input channels (the 7x7 init conv: Cin = 2 or 10): the GEMM N dimension is
// the flattened (tap, channel) index - 98 columns for 7x7x2 instead of 49 taps x a 32-channel tile that is 94 % padding.
// Wave w owns n-tiles {w, w+4, ...} (<= MAXN) and walks the whole 128-pixel tile; lane j of an n-tile reads
// X[halo(p) + tap_j][c_j] straight from the LDS halo tile.
template <int MAXN>
__global__ void __launch_bounds__(256) conv_wgrad_smallc_kernel(WgradGeom wg, const float* __restrict__ src0,
                                                                const float* __restrict__ dy, float* __restrict__ partial,
                                                                float* __restrict__ bias_partial) {
  const ConvGeom& g = wg.g;
  HIP_DYNAMIC_SHARED(float, smem)
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, half = lane >> 5, l31 = lane & 31;
  const int T = g.KH * g.KW, Cin = g.Cin, NJ = T * Cin;
  const int m0 = blockIdx.y * 32;
  const int split = blockIdx.x;
  const int npixA = g.NI * g.IHt * g.IWt;
  float* Xs = smem;                                   // [npixA][Cin]
  float* Ys = smem + (((size_t)npixA * Cin + 3) & ~(size_t)3);   // [128][32]
  const int tpi = g.Hv / g.TH;
  // per-lane column -> offset inside the halo tile (floats), -1 = padding column
  int joff[MAXN], jt[MAXN], jc[MAXN];
#pragma unroll
  for (int i = 0; i < MAXN; ++i) {
    const int j = (wave + 4 * i) * 32 + l31;
    joff[i] = -1; jt[i] = 0; jc[i] = 0;
    if (j < NJ) {
      const int t = j / Cin, c = j - t * Cin;
      jt[i] = t; jc[i] = c;
      joff[i] = ((t / g.KW) * g.IWt + (t % g.KW)) * Cin + c;
    }
  }
  f32x16 acc[MAXN];
#pragma unroll
  for (int i = 0; i < MAXN; ++i)
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  const bool do_bias = bias_partial != nullptr;
  float bacc = 0.f;
  const bool vec_dy = ((wg.ld_dy & 3) == 0) && ((g.Cout & 3) == 0);
  const int tile_lo = split * wg.tiles_per_split;
  const int tile_hi = (tile_lo + wg.tiles_per_split < g.tiles_m) ? tile_lo + wg.tiles_per_split : g.tiles_m;
  for (int tile = tile_lo; tile < tile_hi; ++tile) {
    const int b0 = (tile / tpi) * g.NI, vy0 = (tile % tpi) * g.TH;
    const int iy0 = vy0 * g.stride - g.pad_y[0], ix0 = -g.pad_x[0];
    __syncthreads();
    for (int e = tid; e < npixA * Cin; e += 256) {
      const int c = e % Cin, hp = e / Cin;
      const int hx = hp % g.IWt, hy = (hp / g.IWt) % g.IHt, img = hp / (g.IWt * g.IHt);
      const int b = b0 + img, iy = iy0 + hy, ix = ix0 + hx;
      float v = 0.f;
      if (b < g.B && iy >= 0 && iy < g.Hi && ix >= 0 && ix < g.Wi) v = src0[(((size_t)b * g.Hi + iy) * g.Wi + ix) * g.ld0 + c];
      Xs[e] = v;
    }
    for (int e = tid; e < kBM * 8; e += 256) {
      const int q = e & 7, p = e >> 3;
      const int tx = p & (g.Wv - 1), ty = (p >> g.wsh) & (g.TH - 1), img = p >> (g.wsh + g.tsh);
      const int b = b0 + img, c = m0 + 4 * q;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (b < g.B && img < g.NI && c < g.Cout) {
        const size_t pix = ((size_t)b * g.Hv + (vy0 + ty)) * g.Wv + tx;
        if (vec_dy) {
          v = *reinterpret_cast<const float4*>(dy + pix * wg.ld_dy + c);
        } else {
          float t4[4];
          for (int k = 0; k < 4; ++k) t4[k] = (c + k < g.Cout) ? dy[pix * wg.ld_dy + c + k] : 0.f;
          v = make_float4(t4[0], t4[1], t4[2], t4[3]);
        }
      }
      *reinterpret_cast<float4*>(Ys + (size_t)p * 32 + 4 * q) = v;
    }
    __syncthreads();
    if (do_bias) {
      const int o = tid & 31, part = tid >> 5;
#pragma unroll
      for (int k = 0; k < 16; ++k) bacc += Ys[(part * 16 + k) * 32 + o];
    }
#pragma unroll 8
    for (int ks = 0; ks < 64; ++ks) {
      const int p = 2 * ks + half;
      const int tx = p & (g.Wv - 1), ty = (p >> g.wsh) & (g.TH - 1), img = p >> (g.wsh + g.tsh);
      const int xb = (img < g.NI) ? ((img * g.IHt + ty * g.stride) * g.IWt + tx * g.stride) * Cin : 0;
      const float a = Ys[p * 32 + l31];
#pragma unroll
      for (int i = 0; i < MAXN; ++i) {
        if ((wave + 4 * i) * 32 < NJ) {
          const float bv = (joff[i] >= 0) ? Xs[xb + joff[i]] : 0.f;
          acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, bv, acc[i], 0, 0, 0);
        }
      }
    }
  }
  // results: column j = (t, c) of this lane, rows = 32 output channels
#pragma unroll
  for (int i = 0; i < MAXN; ++i) {
    if (joff[i] >= 0) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = (r & 3) + 8 * (r >> 2) + 4 * half;
        partial[(((size_t)split * wg.MP + (m0 + row)) * T + jt[i]) * wg.NP + jc[i]] = acc[i][r];
      }
    }
  }
  if (do_bias) {
    __syncthreads();
    float* red = Ys;
    red[tid] = bacc;
    __syncthreads();
    if (tid < 32) {
      float sb = 0.f;
      for (int k = 0; k < 8; ++k) sb += red[k * 32 + tid];
      bias_partial[(size_t)split * wg.MP + m0 + tid] = sb;
    }
  }
}

// dst[(m*N + n)*T + t] = sum_s partial[s][m][t][n]  and  dbias[m] = sum_s bias_partial[s][m].
// One block = 32 consecutive outputs (n fastest: coalesced reads of the partials) x 8 split lanes.
__global__ void __launch_bounds__(256) wgrad_reduce_kernel(const float* __restrict__ partial, float* __restrict__ dst,
                                                           const float* __restrict__ bias_partial, float* __restrict__ dbias,
                                                           int nsplit, int M, int N, int T, int MP, int NP) {
  __shared__ float red[8][32];
  const int tid = threadIdx.x, ol = tid & 31, sl = tid >> 5;
  const long nw = (long)M * T * N;
  const long o = (long)blockIdx.x * 32 + ol;
  float s = 0.f;
  long dsti = -1;
  float* dptr = nullptr;
  if (o < nw) {
    const int n = (int)(o % N), t = (int)((o / N) % T), m = (int)(o / ((long)N * T));
    const float* p = partial + ((size_t)m * T + t) * NP + n;
    const size_t sstride = (size_t)MP * T * NP;
    for (int sp = sl; sp < nsplit; sp += 8) s += p[(size_t)sp * sstride];
    dptr = dst;
    dsti = ((long)m * N + n) * T + t;
  } else if (dbias && o < nw + M) {
    const int m = (int)(o - nw);
    for (int sp = sl; sp < nsplit; sp += 8) s += bias_partial[(size_t)sp * MP + m];
    dptr = dbias;
    dsti = m;
  }
  red[sl][ol] = s;
  __syncthreads();
  if (sl == 0 && dptr) {
    float a = 0.f;
#pragma unroll
    for (int k = 0; k < 8; ++k) a += red[k][ol];
    dptr[dsti] = a;
  }
}

// out[m][n] = sum_k A[m][k] W[n][k] for FEW rows m and a very long k (the input gradient of the concatenated FiLM linears:
// M = batch, N = time dimension, K = sum of 2 Cout over all resblocks = 3968 / 15872): as an implicit GEMM that is 4 workgroups
// walking 124 chunks each (134 us for 65 MFLOP).  Here K is split over workgroups of 128 columns each; partial[split][M][N] is
// summed in fixed order by wgrad_reduce_kernel.  grid = (k splits, N / 32, ceil(M / 32)); the 4 waves take a quarter of the
// 128 columns each (k-slots permuted: one 16-byte LDS read feeds four MFMAs) and are summed through LDS.
__global__ void __launch_bounds__(256) smallm_splitk_kernel(const float* __restrict__ A, int lda, const float* __restrict__ W, int ldw,
                                                            float* __restrict__ partial, int M, int N, int K, int MP, int NP) {
  constexpr int KC = 128, KP = KC + 4;
  __shared__ float As[32 * KP], Ws[32 * KP];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, half = lane >> 5, l31 = lane & 31;
  const int k0 = blockIdx.x * KC, n0 = blockIdx.y * 32, m0 = blockIdx.z * 32;
  for (int e = tid; e < 32 * (KC / 4); e += 256) {
    const int row = e / (KC / 4), q = e - row * (KC / 4), k = k0 + 4 * q;
    f32x4 a = {0.f, 0.f, 0.f, 0.f}, w = {0.f, 0.f, 0.f, 0.f};
    if (k < K) {       // K % 4 == 0
      if (m0 + row < M) a = *reinterpret_cast<const f32x4*>(A + (size_t)(m0 + row) * lda + k);
      if (n0 + row < N) w = *reinterpret_cast<const f32x4*>(W + (size_t)(n0 + row) * ldw + k);
    }
    *reinterpret_cast<f32x4*>(As + row * KP + 4 * q) = a;
    *reinterpret_cast<f32x4*>(Ws + row * KP + 4 * q) = w;
  }
  __syncthreads();
  f32x16 acc;
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
  for (int g8 = 0; g8 < 4; ++g8) {
    const f32x4 a4 = *reinterpret_cast<const f32x4*>(As + l31 * KP + 32 * wave + 8 * g8 + 4 * half);
    const f32x4 w4 = *reinterpret_cast<const f32x4*>(Ws + l31 * KP + 32 * wave + 8 * g8 + 4 * half);
#pragma unroll
    for (int i = 0; i < 4; ++i) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[i], w4[i], acc, 0, 0, 0);
  }
  __syncthreads();
  float* red = As;       // [4][1024] needs 4096 floats: As (4224) suffices
#pragma unroll
  for (int r = 0; r < 16; ++r) red[wave * 1024 + ((r & 3) + 8 * (r >> 2) + 4 * half) * 32 + l31] = acc[r];
  __syncthreads();
  for (int e = tid; e < 1024; e += 256) {
    const float sv = (red[e] + red[1024 + e]) + (red[2048 + e] + red[3072 + e]);
    partial[((size_t)blockIdx.x * MP + m0 + (e >> 5)) * NP + n0 + (e & 31)] = sv;
  }
}
size_t smallm_splitk_ws_floats(int M, int N, int K) { return (size_t)cdiv(K, 128) * (cdiv(M, 32) * 32) * (cdiv(N, 32) * 32) + 64; }
bool smallm_splitk_ok(int M, int N, int K, int lda, int ldw) {
  return M <= 512 && K >= 1024 && (K % 4) == 0 && (lda % 4) == 0 && (ldw % 4) == 0 && (N % 32) == 0;
}
int launch_smallm_splitk(const float* A, int lda, const float* W, int ldw, float* out, int M, int N, int K, float* scratch, hipStream_t st) {
  const int ks = cdiv(K, 128), MP = cdiv(M, 32) * 32, NP = cdiv(N, 32) * 32;
  hipLaunchKernelGGL(smallm_splitk_kernel, dim3(ks, NP / 32, MP / 32), dim3(256), 0, st, A, lda, W, ldw, scratch, M, N, K, MP, NP);
  PIDM_CHECK_LAUNCH("smallm_splitk_kernel");
  return launch_split_reduce(scratch, out, nullptr, nullptr, ks, M, N, 1, MP, NP, st);
}

// 1x1 (stride 1) weight gradient without LDS and without barriers: dW[m][n] = sum_p dY[p][m] X[p][n] is a pure stream over
// the pixels - a wave reads its own MFMA fragments straight from global memory (32 consecutive channels = one 128-byte
// line per pixel and operand; every dY element is used exactly once per n-tile) and keeps UNR independent loads per
// operand in flight.  The LDS-staged kernel did 16 MFMAs per wave between two barriers and ran at half the byte rate.
// Pixel range of a workgroup: tiles [split*tps, ...) of 128 pixels, one contiguous quarter per wave; cross-wave sum
// through LDS at the end, split-K partials as everywhere else.
// (problem `wg` by value in scalar registers - kernel arguments or a row of the grouped launch's table -, workgroup (split, by) of its
// splits x tiles grid, 4 x 1024 floats of LDS)
__device__ __forceinline__ void conv_wgrad_1x1_stream_body(const WgradItem& wg, const int split, const int by, float (*red)[1024]) {
  const WgradItem& g = wg;
  const float* __restrict__ src0 = wg.src0;
  const float* __restrict__ src1 = wg.src1;
  const float* __restrict__ dy = wg.dy;
  float* __restrict__ partial = wg.partial;
  float* __restrict__ bias_partial = wg.bias_partial;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, half = lane >> 5, l31 = lane & 31;
  const int ntn = wg.NP / 32;
  const int tn = by % ntn, tm = by / ntn;
  const int m0 = tm * 32, n0 = tn * 32;
  const size_t ptot = (size_t)g.B * g.Hv * g.Wv;
  size_t p_lo = (size_t)split * wg.tiles_per_split * 128, p_hi = p_lo + (size_t)wg.tiles_per_split * 128;
  if (p_hi > ptot) p_hi = ptot;
  if (p_lo > p_hi) p_lo = p_hi;
  const size_t per = ((p_hi - p_lo + 7) / 8) * 2;                     // even number of pixels per wave
  size_t w_lo = p_lo + (size_t)wave * per, w_hi = w_lo + per;
  if (w_lo > p_hi) w_lo = p_hi;
  if (w_hi > p_hi) w_hi = p_hi;
  const int cx = n0 + l31, cy = m0 + l31;
  const bool x_ok = cx < g.Cin, y_ok = cy < g.Cout;
  const float* xsrc = (cx < g.C0) ? src0 + cx : src1 + (cx - g.C0);
  const size_t xld = (cx < g.C0) ? g.ld0 : g.ld1;
  const float* ysrc = dy + cy;
  f32x16 acc;
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  const bool do_bias = (bias_partial != nullptr) && (tn == 0);
  float bacc = 0.f;
  constexpr int UNR = 8;
  size_t q = w_lo;                                   // wave-uniform loop bounds: the MFMA ignores the exec mask
  for (; q + 2 * UNR <= w_hi; q += 2 * UNR) {
    const size_t p = q + half;
    float a[UNR], b[UNR];
#pragma unroll
    for (int u = 0; u < UNR; ++u) {
      a[u] = y_ok ? ysrc[(p + 2 * u) * wg.ld_dy] : 0.f;
      b[u] = x_ok ? xsrc[(p + 2 * u) * xld] : 0.f;
    }
#pragma unroll
    for (int u = 0; u < UNR; ++u) {
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u], b[u], acc, 0, 0, 0);
      bacc += a[u];
    }
  }
  for (; q < w_hi; q += 2) {
    const size_t pp = q + half;
    const float a = (y_ok && pp < w_hi) ? ysrc[pp * wg.ld_dy] : 0.f;
    const float b = (x_ok && pp < w_hi) ? xsrc[pp * xld] : 0.f;
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
    bacc += a;
  }
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int row = (r & 3) + 8 * (r >> 2) + 4 * half;
    red[wave][row * 32 + l31] = acc[r];
  }
  __syncthreads();
  for (int e = tid; e < 1024; e += 256) {
    const float sv = (red[0][e] + red[1][e]) + (red[2][e] + red[3][e]);
    partial[((size_t)split * wg.MP + (m0 + (e >> 5))) * wg.NP + n0 + (e & 31)] = sv;
  }
  if (do_bias) {
    __syncthreads();
    red[0][tid] = bacc;           // lane (wave, half, l31): partial column sum of channel m0 + l31
    __syncthreads();
    if (tid < 32) {
      float sb = 0.f;
      for (int k = 0; k < 8; ++k) sb += red[0][k * 32 + tid];
      bias_partial[(size_t)split * wg.MP + m0 + tid] = sb;
    }
  }
}

// Same stream with a 128-channel "wide" operand read as one float4 per lane (lane l holds channels 4l..4l+3 of pixel p + half:
// 1 KiB per wave-wide load instruction) against 32 channels of the other operand: four accumulators, MFMA tile t pairs
// component t of the wide fragment (channel 4*lane + t - the permuted-tile trick of the NT=4 forward kernel) with the narrow
// fragment, so the narrow operand is re-read four times less often than with 32x32 tiles.  WIDE_DY: the wide side is dY
// (GEMM M), otherwise X (GEMM N).
__global__ void __launch_bounds__(256) conv_wgrad_1x1_stream_kernel(WgradItem wg) {
  __shared__ float red[4][1024];
  conv_wgrad_1x1_stream_body(wg, (int)blockIdx.x, (int)blockIdx.y, red);
}

template <bool WIDE_DY>
__device__ __forceinline__ void conv_wgrad_1x1_stream4_body(const WgradItem& wg, const int bx, const int split, float (*red)[1024]) {
  const WgradItem& g = wg;
  const float* __restrict__ src0 = wg.src0;
  const float* __restrict__ src1 = wg.src1;
  const float* __restrict__ dy = wg.dy;
  float* __restrict__ partial = wg.partial;
  float* __restrict__ bias_partial = wg.bias_partial;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, half = lane >> 5, l31 = lane & 31;
  const int n_nar = WIDE_DY ? wg.NP / 32 : wg.MP / 32;          // 32-channel tiles of the narrow operand
  // tile index fastest: workgroups launched together read neighbouring channel groups of the same pixel rows
  const int t_nar = bx % n_nar, t_wid = bx / n_nar;
  const int w0 = t_wid * 128, r0 = t_nar * 32;
  const size_t ptot = (size_t)g.B * g.Hv * g.Wv;
  size_t p_lo = (size_t)split * wg.tiles_per_split * 128, p_hi = p_lo + (size_t)wg.tiles_per_split * 128;
  if (p_hi > ptot) p_hi = ptot;
  if (p_lo > p_hi) p_lo = p_hi;
  const size_t per = ((p_hi - p_lo + 7) / 8) * 2;
  size_t w_lo = p_lo + (size_t)wave * per, w_hi = w_lo + per;
  if (w_lo > p_hi) w_lo = p_hi;
  if (w_hi > p_hi) w_hi = p_hi;
  // wide operand: channels cw..cw+3, narrow operand: channel cn
  const int cw = w0 + 4 * l31, cn = r0 + l31;
  const float* wsrc;
  const float* nsrc;
  size_t wld, nld;
  bool w_ok, n_ok;
  if (WIDE_DY) {
    wsrc = dy + cw; wld = wg.ld_dy; w_ok = cw < g.Cout;
    nsrc = (cn < g.C0) ? src0 + cn : src1 + (cn - g.C0); nld = (cn < g.C0) ? g.ld0 : g.ld1; n_ok = cn < g.Cin;
  } else {
    wsrc = (cw < g.C0) ? src0 + cw : src1 + (cw - g.C0); wld = (cw < g.C0) ? g.ld0 : g.ld1; w_ok = cw < g.Cin;
    nsrc = dy + cn; nld = wg.ld_dy; n_ok = cn < g.Cout;
  }
  f32x16 acc[4];
  for (int t = 0; t < 4; ++t)
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
  const bool do_bias = (bias_partial != nullptr) && ((WIDE_DY ? t_nar : t_wid) == 0);
  f32x4 bw = {0.f, 0.f, 0.f, 0.f};
  float bn = 0.f;
  const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
  constexpr int UNR = 4;
  size_t q = w_lo;                                   // wave-uniform loop bounds: the MFMA ignores the exec mask
  for (; q + 2 * UNR <= w_hi; q += 2 * UNR) {
    const size_t p = q + half;
    f32x4 w[UNR];
    float n[UNR];
#pragma unroll
    for (int u = 0; u < UNR; ++u) {
      w[u] = w_ok ? *reinterpret_cast<const f32x4*>(wsrc + (p + 2 * u) * wld) : zero4;
      n[u] = n_ok ? nsrc[(p + 2 * u) * nld] : 0.f;
    }
#pragma unroll
    for (int u = 0; u < UNR; ++u) {
#pragma unroll
      for (int t = 0; t < 4; ++t)
        acc[t] = WIDE_DY ? __builtin_amdgcn_mfma_f32_32x32x2f32(w[u][t], n[u], acc[t], 0, 0, 0)
                         : __builtin_amdgcn_mfma_f32_32x32x2f32(n[u], w[u][t], acc[t], 0, 0, 0);
      if (WIDE_DY) bw += w[u]; else bn += n[u];
    }
  }
  for (; q < w_hi; q += 2) {
    const size_t pp = q + half;
    const f32x4 w = (w_ok && pp < w_hi) ? *reinterpret_cast<const f32x4*>(wsrc + pp * wld) : zero4;
    const float n = (n_ok && pp < w_hi) ? nsrc[pp * nld] : 0.f;
#pragma unroll
    for (int t = 0; t < 4; ++t)
      acc[t] = WIDE_DY ? __builtin_amdgcn_mfma_f32_32x32x2f32(w[t], n, acc[t], 0, 0, 0)
                       : __builtin_amdgcn_mfma_f32_32x32x2f32(n, w[t], acc[t], 0, 0, 0);
    if (WIDE_DY) bw += w; else bn += n;
  }
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    if (t) __syncthreads();
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = (r & 3) + 8 * (r >> 2) + 4 * half;
      red[wave][row * 32 + l31] = acc[t][r];
    }
    __syncthreads();
    for (int e = tid; e < 1024; e += 256) {
      const float sv = (red[0][e] + red[1][e]) + (red[2][e] + red[3][e]);
      const int m = WIDE_DY ? w0 + 4 * (e >> 5) + t : r0 + (e >> 5);
      const int n = WIDE_DY ? r0 + (e & 31) : w0 + 4 * (e & 31) + t;
      if (m < wg.MP && n < wg.NP) partial[((size_t)split * wg.MP + m) * wg.NP + n] = sv;
    }
  }
  if (do_bias) {
    __syncthreads();
    if (WIDE_DY) {
      for (int t = 0; t < 4; ++t) red[0][(wave * 2 + half) * 128 + 4 * l31 + t] = bw[t];
      __syncthreads();
      if (tid < 128 && w0 + tid < wg.MP) {
        float sb = 0.f;
        for (int k = 0; k < 8; ++k) sb += red[0][k * 128 + tid];
        bias_partial[(size_t)split * wg.MP + w0 + tid] = sb;
      }
    } else {
      red[0][tid] = bn;
      __syncthreads();
      if (tid < 32) {
        float sb = 0.f;
        for (int k = 0; k < 8; ++k) sb += red[0][k * 32 + tid];
        bias_partial[(size_t)split * wg.MP + r0 + tid] = sb;
      }
    }
  }
}

template <bool WIDE_DY>
__global__ void __launch_bounds__(256, 4) conv_wgrad_1x1_stream4_kernel(WgradItem wg) {
  __shared__ float red[4][1024];
  conv_wgrad_1x1_stream4_body<WIDE_DY>(wg, (int)blockIdx.x, (int)blockIdx.y, red);
}

// The same 128 x 32 wave tile on the bf16 pipe (round 6): the stream above is bound by its fp32 MFMAs - 32 of them (2048 cycles of
// the SIMD's vector ALUs) per 16 pixels, 45-70 TFLOP/s over the to_qkv / res_conv problems of a step.  Here a k-step is 16 PIXELS:
// lane (l31, half) loads pixels q + 8 half + 0..7 of its channel(s) - the strided loads ARE the transpose into the MFMA's
// k-contiguous operand fragment -, splits them into three bf16 pieces (5 fragments x 4 pairs x 11 vector instructions) and issues
// 4 tiles x 6 v_mfma_f32_32x32x16_bf16 (768 cycles of the matrix pipe).  The next k-step's 16 loads per lane are in flight while
// this one is split and multiplied (~190 registers: two waves per SIMD).  Same tile mapping, partial layout, cross-wave sum and
// bias columns as conv_wgrad_1x1_stream4_body; pixels past the wave's range are loaded as zeros.
template <bool WIDE_DY>
__device__ __forceinline__ void conv_wgrad_1x1_split4_body(const WgradItem& wg, const int bx, const int split, float (*red)[1024]) {
  const WgradItem& g = wg;
  const float* __restrict__ src0 = wg.src0;
  const float* __restrict__ src1 = wg.src1;
  const float* __restrict__ dy = wg.dy;
  float* __restrict__ partial = wg.partial;
  float* __restrict__ bias_partial = wg.bias_partial;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, half = lane >> 5, l31 = lane & 31;
  const int n_nar = WIDE_DY ? wg.NP / 32 : wg.MP / 32;
  const int t_nar = bx % n_nar, t_wid = bx / n_nar;
  const int w0 = t_wid * 128, r0 = t_nar * 32;
  const size_t ptot = (size_t)g.B * g.Hv * g.Wv;
  size_t p_lo = (size_t)split * wg.tiles_per_split * 128, p_hi = p_lo + (size_t)wg.tiles_per_split * 128;
  if (p_hi > ptot) p_hi = ptot;
  if (p_lo > p_hi) p_lo = p_hi;
  const size_t per = ((p_hi - p_lo + 7) / 8) * 2;
  size_t w_lo = p_lo + (size_t)wave * per, w_hi = w_lo + per;
  if (w_lo > p_hi) w_lo = p_hi;
  if (w_hi > p_hi) w_hi = p_hi;
  const int cw = w0 + 4 * l31, cn = r0 + l31;
  const float* wsrc;
  const float* nsrc;
  size_t wld, nld;
  bool w_ok, n_ok;
  if (WIDE_DY) {
    wsrc = dy + cw; wld = wg.ld_dy; w_ok = cw < g.Cout;
    nsrc = (cn < g.C0) ? src0 + cn : src1 + (cn - g.C0); nld = (cn < g.C0) ? g.ld0 : g.ld1; n_ok = cn < g.Cin;
  } else {
    wsrc = (cw < g.C0) ? src0 + cw : src1 + (cw - g.C0); wld = (cw < g.C0) ? g.ld0 : g.ld1; w_ok = cw < g.Cin;
    nsrc = dy + cn; nld = wg.ld_dy; n_ok = cn < g.Cout;
  }
  f32x16 acc[4];
  for (int t = 0; t < 4; ++t)
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
  const bool do_bias = (bias_partial != nullptr) && ((WIDE_DY ? t_nar : t_wid) == 0);
  f32x4 bw = {0.f, 0.f, 0.f, 0.f};
  float bn = 0.f;
  const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
  // Branch-free loads: a lane whose channel does not exist reads channel 0 of its operand (its products land in rows / columns of
  // the padded slab that nothing reads), a pixel past the wave's range is clamped to the range's last pixel and ZEROED before use
  // (tail k-step only).  With conditional loads hipcc wrapped every one of the 16 loads in a branch and waited for all of them -
  // the prefetch included - in front of the arithmetic (s_waitcnt vmcnt(0)): a full memory latency per k-step, 141 us for the
  // to_qkv problem of the 16 x 16 level at batch 256 where this form takes 122 (fp32 stream: 169; profiles/r06_wgrad1x1_split.txt).
  const float* const wp = w_ok ? wsrc : (WIDE_DY ? dy : src0);
  const float* const np = n_ok ? nsrc : (WIDE_DY ? src0 : dy);
  if (!w_ok) wld = WIDE_DY ? (size_t)wg.ld_dy : (size_t)g.ld0;
  if (!n_ok) nld = WIDE_DY ? (size_t)g.ld0 : (size_t)wg.ld_dy;
  f32x4 wa[8], wb[8];
  float na[8], nb[8];
#define PIDM_WG1S_LOAD(w_, n_, q_)                                                                                  \
  {                                                                                                                 \
    const size_t pb_ = (q_) + 8 * half;                                                                             \
    _Pragma("unroll") for (int j = 0; j < 8; ++j) {                                                                 \
      const size_t pp_ = pb_ + j < w_hi ? pb_ + j : w_hi - 1;                                                       \
      w_[j] = *reinterpret_cast<const f32x4*>(wp + pp_ * wld);                                                      \
      n_[j] = np[pp_ * nld];                                                                                        \
    }                                                                                                               \
  }
#define PIDM_WG1S_COMPUTE(w_, n_, q_)                                                                               \
  {                                                                                                                 \
    if ((q_) + 16 > w_hi) {                          /* wave-uniform: the range's last, partial k-step */           \
      const size_t pb_ = (q_) + 8 * half;                                                                           \
      _Pragma("unroll") for (int j = 0; j < 8; ++j) {                                                               \
        const bool in_ = pb_ + j < w_hi;                                                                            \
        w_[j] = in_ ? w_[j] : zero4;                                                                                \
        n_[j] = in_ ? n_[j] : 0.f;                                                                                  \
      }                                                                                                             \
    }                                                                                                               \
    if (do_bias) {     /* the k-step's 8 pixels as a tree, then one add to the running sum: a chain of range / 16 adds */ \
      if (WIDE_DY) bw += ((w_[0] + w_[1]) + (w_[2] + w_[3])) + ((w_[4] + w_[5]) + (w_[6] + w_[7]));                 \
      else bn += ((n_[0] + n_[1]) + (n_[2] + n_[3])) + ((n_[4] + n_[5]) + (n_[6] + n_[7]));                         \
    }                                                                                                               \
    u32x4 nf[3];                                                                                                    \
    _Pragma("unroll") for (int k = 0; k < 4; ++k) {                                                                 \
      unsigned p0, p1, p2;                                                                                          \
      pidm_split3_pk(n_[2 * k], n_[2 * k + 1], p0, p1, p2);                                                         \
      nf[0][k] = p0; nf[1][k] = p1; nf[2][k] = p2;                                                                  \
    }                                                                                                               \
    /* two tiles' pieces, then their 12 MFMAs term by term across the two: consecutive MFMAs write different accumulators (a   \
       tile's six terms back to back are one dependency chain; all four tiles' pieces at once spill) */                      \
    _Pragma("unroll") for (int tp = 0; tp < 4; tp += 2) {                                                           \
      u32x4 wf[2][3];                                                                                               \
      _Pragma("unroll") for (int t = 0; t < 2; ++t) {                                                               \
        _Pragma("unroll") for (int k = 0; k < 4; ++k) {                                                             \
          unsigned p0, p1, p2;                                                                                      \
          pidm_split3_pk(w_[2 * k][tp + t], w_[2 * k + 1][tp + t], p0, p1, p2);                                     \
          wf[t][0][k] = p0; wf[t][1][k] = p1; wf[t][2][k] = p2;                                                     \
        }                                                                                                           \
      }                                                                                                             \
      /* smallest terms first (the order of every split-form kernel of this library): (2,0) (0,2) (1,1) (1,0) (0,1) (0,0) */ \
      _Pragma("unroll") for (int term = 0; term < 6; ++term) {                                                      \
        const int iw = term == 0 ? 2 : (term == 2 || term == 3) ? 1 : 0;                                            \
        const int in = term == 1 ? 2 : (term == 2 || term == 4) ? 1 : 0;                                            \
        _Pragma("unroll") for (int t = 0; t < 2; ++t)                                                               \
          acc[tp + t] = WIDE_DY ? pidm_mfma_bf16_32x32x16(wf[t][iw], nf[in], acc[tp + t])                           \
                                : pidm_mfma_bf16_32x32x16(nf[iw], wf[t][in], acc[tp + t]);                          \
      }                                                                                                             \
    }                                                                                                               \
  }
  // two register sets take turns: the loads of k-step i + 1 are issued in front of the arithmetic of k-step i
  if (w_lo < w_hi) {
    size_t q = w_lo;
    PIDM_WG1S_LOAD(wa, na, q)
    for (;;) {
      // (unconditional: past the range the clamp makes these 16 loads of the last pixel's lines.  Behind `if (more)` hipcc's
      // wait-count pass joins the two paths with the count of the path WITHOUT the prefetch - and the arithmetic waits for it)
      PIDM_WG1S_LOAD(wb, nb, q + 16)
      PIDM_WG1S_COMPUTE(wa, na, q)
      q += 16;
      if (q >= w_hi) break;                            // wave-uniform
      PIDM_WG1S_LOAD(wa, na, q + 16)
      PIDM_WG1S_COMPUTE(wb, nb, q)
      q += 16;
      if (q >= w_hi) break;
    }
  }
#undef PIDM_WG1S_LOAD
#undef PIDM_WG1S_COMPUTE
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    if (t) __syncthreads();
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = (r & 3) + 8 * (r >> 2) + 4 * half;
      red[wave][row * 32 + l31] = acc[t][r];
    }
    __syncthreads();
    for (int e = tid; e < 1024; e += 256) {
      const float sv = (red[0][e] + red[1][e]) + (red[2][e] + red[3][e]);
      const int m = WIDE_DY ? w0 + 4 * (e >> 5) + t : r0 + (e >> 5);
      const int nc = WIDE_DY ? r0 + (e & 31) : w0 + 4 * (e & 31) + t;
      if (m < wg.MP && nc < wg.NP) partial[((size_t)split * wg.MP + m) * wg.NP + nc] = sv;
    }
  }
  if (do_bias) {
    __syncthreads();
    if (WIDE_DY) {
      for (int t = 0; t < 4; ++t) red[0][(wave * 2 + half) * 128 + 4 * l31 + t] = bw[t];
      __syncthreads();
      if (tid < 128 && w0 + tid < wg.MP) {
        float sb = 0.f;
        for (int k = 0; k < 8; ++k) sb += red[0][k * 128 + tid];
        bias_partial[(size_t)split * wg.MP + w0 + tid] = sb;
      }
    } else {
      red[0][tid] = bn;
      __syncthreads();
      if (tid < 32) {
        float sb = 0.f;
        for (int k = 0; k < 8; ++k) sb += red[0][k * 32 + tid];
        bias_partial[(size_t)split * wg.MP + r0 + tid] = sb;
      }
    }
  }
}

// XCD-aware order of a problem's workgroups (observed, for speed only: hardware block h runs on XCD h % 8, each XCD has its own L2).
// The n workgroups [first, first + n) of a problem are renumbered so that the ones of XCD k take one CONTIGUOUS range of logical
// ids: with logical id = split * tiles + tile, the tiles of a split - which read the same pixels - then share an L2 instead of
// having every XCD fetch every pixel range (number of x < A with x % 8 == i: (A + 7 - i) / 8).
__device__ __forceinline__ unsigned wgrad_xcd_order(unsigned h, unsigned first, unsigned n) {
  const unsigned k = h & 7u;
  unsigned before = 0;
#pragma unroll
  for (unsigned i = 0; i < 8; ++i)
    if (i < k) before += (first + n + 7u - i) / 8u - (first + 7u - i) / 8u;
  return before + (h + 7u - k) / 8u - (first + 7u - k) / 8u;
}

template <bool WIDE_DY>
__global__ void __launch_bounds__(256, 2) conv_wgrad_1x1_split4_kernel(WgradItem wg, int xcd) {
  __shared__ float red[4][1024];
  unsigned local = blockIdx.y * gridDim.x + blockIdx.x;
  if (xcd) local = wgrad_xcd_order(local, 0u, gridDim.x * gridDim.y);
  conv_wgrad_1x1_split4_body<WIDE_DY>(wg, (int)(local % wg.gx), (int)(local / wg.gx), red);
}

// the split-form problems of a pass in one launch (family kWgFam1x1Split: its own table - these bodies need two waves per SIMD's
// registers, the fp32 streams four)
__global__ void __launch_bounds__(256, 2) conv_wgrad_1x1_split_multi_kernel(const WgradItem* __restrict__ table, int n, unsigned blk_base,
                                                                            int xcd) {
  __shared__ float red[4][1024];
  const unsigned bid = blockIdx.x + blk_base;
  const int lane_ = threadIdx.x & 63;
  const unsigned first_ = lane_ < n ? table[lane_].blk0 : 0xffffffffu;
  const int p = __builtin_amdgcn_readfirstlane(__popcll(__ballot(bid >= first_)) - 1);
  const WgradItem wg = table[p];
  // (the XCD of a workgroup follows its index in THIS launch: blockIdx.x)
  const unsigned local = xcd ? wgrad_xcd_order(blockIdx.x, wg.blk0 - blk_base, wg.gx * wg.gy) : bid - wg.blk0;
  const int bx = (int)(local % wg.gx), by = (int)(local / wg.gx);
  if (wg.kind == kWgKindStream4Dy) conv_wgrad_1x1_split4_body<true>(wg, bx, by, red);
  else conv_wgrad_1x1_split4_body<false>(wg, bx, by, red);
}

// The three 1x1 stream kernels for a TABLE of problems in one launch (round 5; WgradQueue, pidm_launch.h; the lookup of
// conv_wgrad_rs_multi_kernel): a problem's `kind` - wave-uniform - picks the body, its own grid is gx x gy.
__global__ void __launch_bounds__(256, 4) conv_wgrad_1x1_multi_kernel(const WgradItem* __restrict__ table, int n, unsigned blk_base) {
  __shared__ float red[4][1024];
  const unsigned bid = blockIdx.x + blk_base;
  const int lane_ = threadIdx.x & 63;
  const unsigned first_ = lane_ < n ? table[lane_].blk0 : 0xffffffffu;
  const int p = __builtin_amdgcn_readfirstlane(__popcll(__ballot(bid >= first_)) - 1);
  const WgradItem wg = table[p];
  const unsigned local = bid - wg.blk0;
  const int bx = (int)(local % wg.gx), by = (int)(local / wg.gx);
  if (wg.kind == kWgKindStream4Dy) conv_wgrad_1x1_stream4_body<true>(wg, bx, by, red);
  else if (wg.kind == kWgKindStream4X) conv_wgrad_1x1_stream4_body<false>(wg, bx, by, red);
  else conv_wgrad_1x1_stream_body(wg, bx, by, red);
}

// all deferred reductions of one backward pass in ONE launch (descriptor table on the device, binary search per block
// like pack_multi_kernel); same arithmetic as wgrad_reduce_kernel with 4 independent chains per split lane
__global__ void __launch_bounds__(256) reduce_multi_kernel(const ReduceDesc* __restrict__ table, int ndesc, unsigned blk_base) {
  __shared__ float red[8][32];
  int lo = 0, hi = ndesc - 1;
  const unsigned bid = blockIdx.x + blk_base;   // descriptors carry absolute block offsets; a launch may cover a sub-range
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (table[mid].blk0 <= bid) lo = mid; else hi = mid - 1;
  }
  const ReduceDesc d = table[lo];
  if (d.mode == 1) {
    // block = (row m, 64 columns): T partial rows of 64 floats per split are read as whole lines, summed in split order, turned
    // through LDS and written as ONE contiguous run of 64*T floats of the [m][n][t] result
    __shared__ float tile[64 * 16];
    const int nb = (d.N + 63) / 64;
    const int m = (int)((bid - d.blk0) / nb), n0 = (int)((bid - d.blk0) % nb) * 64;
    const int nn = (d.N - n0 < 64) ? d.N - n0 : 64;
    for (int e = threadIdx.x; e < 64 * d.T; e += 256) {
      const int t = e >> 6, nl = e & 63;
      float a = 0.f;
      if (nl < nn) {
        const float* p = d.src + ((size_t)m * d.T + t) * d.NP + n0 + nl;
        for (int sp = 0; sp < d.nsplit; ++sp) a += p[(size_t)sp * d.sstride];
      }
      tile[nl * d.T + t] = a;
    }
    __syncthreads();
    float* o = d.dst + ((size_t)m * d.N + n0) * d.T;
    for (int e = threadIdx.x; e < nn * d.T; e += 256) o[e] = tile[e];
    if (d.bdst && n0 == 0 && threadIdx.x == 0) {       // the row's bias gradient rides with its first column block
      float a = 0.f;
      for (int sp = 0; sp < d.nsplit; ++sp) a += d.bsrc[(size_t)sp * d.MP + m];
      d.bdst[m] = a;
    }
    return;
  }
  if (d.mode == 2) {
    __shared__ f32x4 red4[8][32];
    const int tid = threadIdx.x, ql = tid & 31, sl = tid >> 5;
    const size_t flat = (size_t)d.MP * d.T * d.NP;
    const size_t f0 = ((size_t)(bid - d.blk0) * 32 + ql) * 4;
    f32x4 s0 = {0.f, 0.f, 0.f, 0.f}, s1 = s0, s2 = s0, s3 = s0;
    if (f0 < flat) {
      const float* p = d.src + f0;
      int sp = sl;
      for (; sp + 24 < d.nsplit; sp += 32) {
        s0 += *reinterpret_cast<const f32x4*>(p + (size_t)sp * d.sstride);
        s1 += *reinterpret_cast<const f32x4*>(p + (size_t)(sp + 8) * d.sstride);
        s2 += *reinterpret_cast<const f32x4*>(p + (size_t)(sp + 16) * d.sstride);
        s3 += *reinterpret_cast<const f32x4*>(p + (size_t)(sp + 24) * d.sstride);
      }
      for (; sp < d.nsplit; sp += 8) s0 += *reinterpret_cast<const f32x4*>(p + (size_t)sp * d.sstride);
    }
    red4[sl][ql] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    if (sl == 0 && f0 < flat) {
      f32x4 a = red4[0][ql];
#pragma unroll
      for (int k = 1; k < 8; ++k) a += red4[k][ql];
      const unsigned tn = (unsigned)d.T * (unsigned)d.NP;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const unsigned f = (unsigned)f0 + k, m = f / tn, rem = f - m * tn, t = rem / (unsigned)d.NP, n = rem - t * (unsigned)d.NP;
        if ((int)m < d.M && (int)n < d.N) d.dst[((size_t)m * d.N + n) * d.T + t] = a[k];
      }
    }
    return;
  }
  const int tid = threadIdx.x, ol = tid & 31, sl = tid >> 5;
  const long nw = (long)d.M * d.T * d.N;
  const long o = (long)(bid - d.blk0) * 32 + ol;
  const float* p = nullptr;
  size_t sstride = 0;
  long dsti = -1;
  float* dptr = nullptr;
  if (o < nw) {
    const int n = (int)(o % d.N), t = (int)((o / d.N) % d.T), m = (int)(o / ((long)d.N * d.T));
    p = d.src + ((size_t)m * d.T + t) * d.NP + n;
    sstride = d.sstride;
    dptr = d.dst;
    dsti = ((long)m * d.N + n) * d.T + t;
  } else if (d.bdst && o < nw + d.M) {
    const int m = (int)(o - nw);
    p = d.bsrc + m;
    sstride = (size_t)d.MP;
    dptr = d.bdst;
    dsti = m;
  }
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  if (p) {
    int sp = sl;
    for (; sp + 24 < d.nsplit; sp += 32) {
      s0 += p[(size_t)sp * sstride];
      s1 += p[(size_t)(sp + 8) * sstride];
      s2 += p[(size_t)(sp + 16) * sstride];
      s3 += p[(size_t)(sp + 24) * sstride];
    }
    for (; sp < d.nsplit; sp += 8) s0 += p[(size_t)sp * sstride];
  }
  red[sl][ol] = (s0 + s1) + (s2 + s3);
  __syncthreads();
  if (sl == 0 && dptr) {
    float a = 0.f;
#pragma unroll
    for (int k = 0; k < 8; ++k) a += red[k][ol];
    dptr[dsti] = a;
  }
}

// ---------------------------------------------------------------------------------------------------
// column sums (bias gradients etc.): out[c] = sum_r x[r*ld + c], two deterministic stages
// ---------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) colsum_partial_kernel(const float* __restrict__ x, size_t rows, int C, int ld,
                                                             size_t rows_per_block, float* __restrict__ partial) {
  // block (bx, by): rows [bx*rpb, ...), columns [by*64, by*64+64); thread (r = tid/64, c = tid%64)
  __shared__ float red[4][64];
  const int tid = threadIdx.x, cl = tid & 63, rl = tid >> 6;
  const int c = blockIdx.y * 64 + cl;
  const size_t r0 = (size_t)blockIdx.x * rows_per_block;
  const size_t r1 = (r0 + rows_per_block < rows) ? r0 + rows_per_block : rows;
  float s = 0.f;
  if (c < C)
    for (size_t r = r0 + rl; r < r1; r += 4) s += x[r * ld + c];
  red[rl][cl] = s;
  __syncthreads();
  if (rl == 0 && c < C) partial[(size_t)blockIdx.x * C + c] = (red[0][cl] + red[1][cl]) + (red[2][cl] + red[3][cl]);
}
__global__ void colsum_final_kernel(const float* __restrict__ partial, int nblk, int C, float* __restrict__ out) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  float s = 0.f;
  for (int b = 0; b < nblk; ++b) s += partial[(size_t)b * C + c];
  out[c] = s;
}

// dst[(m*N + n)*T + t] = sum over nsplit slabs partial[split][MP][T][NP] (fixed order); optional bias column sums
int launch_split_reduce(const float* partial, float* dst, const float* bias_partial, float* dbias, int nsplit, int M, int N, int T,
                        int MP, int NP, hipStream_t st) {
  const size_t total = (size_t)M * N * T + (dbias ? M : 0);
  hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((unsigned)((total + 31) / 32)), dim3(256), 0, st, partial, dst, bias_partial, dbias,
                     nsplit, M, N, T, MP, NP);
  PIDM_CHECK_LAUNCH("wgrad_reduce_kernel");
  return 0;
}

// PIDM_WGRAD1X1_XCD=0: the split-form 1x1 streams keep the dispatch order (tile fastest) instead of wgrad_xcd_order
static int wgrad_1x1_xcd() {
  const char* e = knob("PIDM_WGRAD1X1_XCD");
  return (e && !atoi(e)) ? 0 : 1;
}

int launch_wgrad_1x1_split_multi(const WgradItem* table_dev, int first, int n, unsigned blk_base, unsigned nblocks, hipStream_t st) {
  if (n <= 0 || nblocks == 0) return 0;
  PIDM_PROF_NAME("conv_wgrad_1x1_split_multi_kernel");
  hipLaunchKernelGGL(conv_wgrad_1x1_split_multi_kernel, dim3(nblocks), dim3(256), 0, st, table_dev + first, n, blk_base, wgrad_1x1_xcd());
  PIDM_CHECK_LAUNCH("conv_wgrad_1x1_split_multi_kernel");
  return 0;
}

int launch_wgrad_1x1_multi(const WgradItem* table_dev, int first, int n, unsigned blk_base, unsigned nblocks, hipStream_t st) {
  if (n <= 0 || nblocks == 0) return 0;
  PIDM_PROF_NAME("conv_wgrad_1x1_multi_kernel");
  hipLaunchKernelGGL(conv_wgrad_1x1_multi_kernel, dim3(nblocks), dim3(256), 0, st, table_dev + first, n, blk_base);
  PIDM_CHECK_LAUNCH("conv_wgrad_1x1_multi_kernel");
  return 0;
}

int launch_reduce_multi(const ReduceDesc* table_dev, int ndesc, unsigned nblocks, hipStream_t st, unsigned blk_base) {
  if (ndesc <= 0 || nblocks == 0) return 0;
  hipLaunchKernelGGL(reduce_multi_kernel, dim3(nblocks), dim3(256), 0, st, table_dev, ndesc, blk_base);
  PIDM_CHECK_LAUNCH("reduce_multi_kernel");
  return 0;
}

// ---- wgrad ------------------------------------------------------------------------------------------
static bool wgrad_smallc(const ConvGeom& g) {
  const int T = g.KH * g.KW;
  return g.nph == 1 && g.C1 == 0 && g.Cin <= 16 && T > 1 && T * g.Cin <= 16 * 32;
}

static int wgrad_taps(const ConvGeom& g) { return g.nph > 1 ? 16 : g.KH * g.KW; }   // taps of the weight tensor

// 0: LDS-staged kernels, 1: LDS-free stream with 32x32 tiles, 2: stream with a 128-wide dY operand, 3: 128-wide X operand
static int wgrad_stream_mode(const ConvGeom& g, int ld_dy) {
  const bool off = knob("PIDM_NO_WGRAD_STREAM") != nullptr;
  const bool off4 = knob("PIDM_NO_WGRAD_STREAM4") != nullptr;
  if (off || g.KH != 1 || g.KW != 1 || g.stride != 1 || g.nph != 1 || g.nz != 1 || wgrad_smallc(g)) return 0;
  if (g.C1 != 0 && g.C0 % 32 != 0) return 0;
  if (off4) return 1;
  const bool dy4 = g.Cout >= 128 && (g.Cout & 3) == 0 && (ld_dy & 3) == 0;
  const bool x4 = g.Cin >= 128 && (g.Cin & 3) == 0 && (g.ld0 & 3) == 0 && (g.C1 == 0 || (g.ld1 & 3) == 0);
  if (dy4 && (g.Cout >= g.Cin || !x4)) return 2;
  if (x4) return 3;
  return 1;
}

// 1x1 problems with a >= 128-channel operand run on the bf16 pipe (conv_wgrad_1x1_split4_body; PIDM_WGRAD1X1_SPLIT=0 or
// PIDM_WGRAD_SPLIT=0: the fp32 streams)
static bool wgrad_1x1_split_on(const ConvGeom& g, int ld_dy) {
  const char* s1e = knob("PIDM_WGRAD1X1_SPLIT");
  const char* wse = knob("PIDM_WGRAD_SPLIT");
  return wgrad_stream_mode(g, ld_dy) >= 2 && !(s1e && !atoi(s1e)) && !(wse && !atoi(wse));
}

static void wgrad_plan(const ConvGeom& g, int ld_dy, WgradGeom* wg) {
  wg->g = g;
  wg->ld_dy = ld_dy;
  const int T = g.KH * g.KW;
  wg->tgs = T < 9 ? T : 9;
  if (T == 16) wg->tgs = 8;
  if (T == 49) wg->tgs = 7;
  if (wgrad_smallc(g)) wg->tgs = T;   // (tap, channel) flattened: one block covers all taps
  wg->ntg = cdiv(T, wg->tgs);
  wg->MP = cdiv(g.Cout, 32) * 32;
  // small-C kernel: (tap, channel) is one flattened GEMM column index -> unpadded partial rows of T*Cin contiguous floats
  // (padded to 32 channels, the init conv's 3136-element gradient cost 103 MB of scattered partial writes and reads)
  wg->NP = wgrad_smallc(g) ? g.Cin : cdiv(g.Cin, 32) * 32;
  int blocks_mn = wgrad_smallc(g) ? (wg->MP / 32) : (wg->MP / 32) * (wg->NP / 32) * wg->ntg * g.nph;
  const int smode = wgrad_stream_mode(g, ld_dy);
  if (smode == 2) blocks_mn = cdiv(wg->MP, 128) * (wg->NP / 32);
  if (smode == 3) blocks_mn = cdiv(wg->NP, 128) * (wg->MP / 32);
  // the 4x4 / stride-2 layers on the row-streaming bf16 kernel (k_wgrad_rs.hip): one workgroup per (split, 32 x 32 block) and CU
  const char* rse = knob("PIDM_WGRAD_RS");
  const bool rs4 = !(rse && !atoi(rse)) && wgrad_rs4_eligible(g, ld_dy);
  if (rs4) blocks_mn = (wg->MP / 32) * (wg->NP / 32);
  // two workgroups per CU are resident: pick tiles-per-split so that the number of workgroup "rounds" over the
  // 512 slots times the per-workgroup work (+ ~1 tile-equivalent of prologue / epilogue) is minimal
  int best_tps = g.tiles_m;
  double best_cost = 1e30;
  for (int tps = 1; tps <= g.tiles_m; ++tps) {
    const long wgs = (long)blocks_mn * cdiv(g.tiles_m, tps);
    if (wgs > 4096 && tps < g.tiles_m) continue;
    const double cost = (T == 9 || rs4) ? (double)((wgs + 255) / 256) * (tps + 1.5) // 3x3 (and 4x4/s2 row-streaming): 1 workgroup per CU
                        : (T == 1) ? (double)((wgs + 1023) / 1024) * (tps + 1.0)    // 1x1: <= 128 registers, 4 workgroups per CU
                                   : (double)((wgs + 511) / 512) * (tps + 1.0);     // 2x2 (phased 4x4/s2): two per CU
    if (cost < best_cost - 1e-9) { best_cost = cost; best_tps = tps; }
  }
  wg->tiles_per_split = best_tps;
  wg->nsplit = cdiv(g.tiles_m, wg->tiles_per_split);
}

size_t wgrad_ws_bytes(const ConvGeom& g) {
  WgradGeom wg;
  wgrad_plan(g, 4, &wg);
  WgradGeom wu;
  wgrad_plan(g, 1, &wu);          // a dY leading dimension that is not a multiple of 4 picks another kernel and split
  const size_t ns = wg.nsplit > wu.nsplit ? wg.nsplit : wu.nsplit;
  return ns * wg.MP * wgrad_taps(g) * wg.NP * sizeof(float) + ns * wg.MP * sizeof(float) + 256;
}

// 3x3 / stride 1 on the bf16 pipe (conv_wgrad_split_kernel) when the geometry allows it; PIDM_WGRAD_SPLIT=0: off.  On success *used
// holds the tiling and split actually launched (never more splits than the plan the workspace was sized for).
static int wgs_xrow_bytes_host(const ConvGeom& g) {
  int xr = ((g.NI * g.IHt * (g.Wv + 8) + 8) * 2 + 15) & ~15;
  if (((xr >> 4) & 1) == 0) xr += 16;
  return xr;
}
static bool launch_wgrad_split(const WgradGeom& plan, const float* src0, const float* src1, const float* dy, int ld_dy, float* partial,
                               float* bias_partial, hipStream_t st, WgradGeom* used) {
  const ConvGeom& g = plan.g;
  const char* se = knob("PIDM_WGRAD_SPLIT");
  if (se && !atoi(se)) return false;
  if (!(g.KH == 3 && g.KW == 3 && g.stride == 1 && g.nph == 1 && g.nz == 1 && g.pad_y[0] == 1 && g.pad_x[0] == 1 && g.Wv == g.Wi &&
        g.Hv == g.Hi && g.Wv >= 8 && (g.Cin % 32 == 0) && (g.C0 % 32 == 0) && (g.Cout % 32 == 0) && (g.C1 == 0 || g.ld1 == g.ld0) &&
        (g.ld0 & 3) == 0 && (ld_dy & 3) == 0 && (reinterpret_cast<size_t>(src0) & 15) == 0 &&
        (!src1 || (reinterpret_cast<size_t>(src1) & 15) == 0) && (reinterpret_cast<size_t>(dy) & 15) == 0))
    return false;
  // the 128-pixel tile first where it measures faster than the 256-pixel one (shorter stage / k-step phases, same work per split):
  // 64-wide images 4-7 % (64->32: 86 -> 80 us), 32- and 16-wide 0-3 %; the 8-wide levels are 1-2 % better on 256 (four whole images per
  // tile instead of two).  PIDM_WGRAD_SPLIT_P = 128 | 256 forces the first try.
  const char* pe = knob("PIDM_WGRAD_SPLIT_P");
  const int p0 = (pe && atoi(pe) == 128) ? 128 : (pe && atoi(pe) == 256) ? 256 : (g.Wv >= 16 ? 128 : 256);
  for (int P = p0; P >= 128; P >>= 1) {
    WgradGeom wg = plan;
    if (!retile_bm(&wg.g, P)) continue;
    const ConvGeom& gp = wg.g;
    const int seg = gp.NI * gp.IHt * gp.Wv, nslot = (P == 256) ? 7 : 4;
    const size_t stage = (size_t)96 * (wgs_xrow_bytes_host(gp) + P * 2 + 16);
    const size_t lds = stage > 12 * 3 * 4096 ? stage : 12 * 3 * 4096;
    if (gp.NI * gp.TH * gp.Wv != P || seg % 8 || seg / 8 + P / 8 > 12 * nslot || lds > 160 * 1024 - 512) continue;
    int ns = plan.nsplit < gp.tiles_m ? plan.nsplit : gp.tiles_m;
    const char* me = knob("PIDM_WGRAD_SPLIT_MAXNS");   // tests: several tiles per split on small problems
    if (me && atoi(me) > 0 && atoi(me) < ns) ns = atoi(me);
    wg.tiles_per_split = cdiv(gp.tiles_m, ns);
    wg.nsplit = cdiv(gp.tiles_m, wg.tiles_per_split);
    const dim3 grid(wg.nsplit, (wg.MP / 32) * (wg.NP / 32), 1);
    static bool attr_ = false;
    if (!attr_) {
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_wgrad_split_kernel<256>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 512);
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_wgrad_split_kernel<128>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 512);
      attr_ = true;
    }
    if (knob("PIDM_TRACE_CONV")) fprintf(stderr, "[pidm]   -> conv_wgrad_split_kernel<%d>, %d splits x %d blocks, %d tiles each, %zu B LDS\n", P, wg.nsplit, grid.y, wg.tiles_per_split, lds);
    PIDM_PROF_NAME("conv_wgrad_split_kernel");
    if (P == 256)
      hipLaunchKernelGGL(HIP_KERNEL_NAME(conv_wgrad_split_kernel<256>), grid, dim3(768), lds, st, wg, src0, src1 ? src1 : src0, dy, partial, bias_partial);
    else
      hipLaunchKernelGGL(HIP_KERNEL_NAME(conv_wgrad_split_kernel<128>), grid, dim3(768), lds, st, wg, src0, src1 ? src1 : src0, dy, partial, bias_partial);
    *used = wg;
    return true;
  }
  return false;
}

// dW[(m*Cin + n)*T + t] written to dw_ref (m over g.Cout = dY channels, n over g.Cin = X channels)
// dbias (may be null) = column sums of dy, fused into the same two launches
int launch_wgrad(const ConvGeom& g, const float* src0, const float* src1, const float* dy, int ld_dy, float* dw_ref,
                 float* dbias, void* workspace, hipStream_t st, ReduceQueue* defer, WgradQueue* wq) {
  if (g.nz != 1) return fail("wgrad: transposed problems must be passed with swapped operands");
  if (!defer) wq = nullptr;      // a queued problem's reduction must be a deferred one as well
  const size_t queued0 = wq ? wq->size() : 0;
  // PIDM_WGRAD_GROUP_FAMS (A/B measurements): bit 0 / 1 / 2 = the 3x3 row-streaming / 4x4-stride-2 row-streaming / 1x1 stream family may queue
  const char* fe = knob("PIDM_WGRAD_GROUP_FAMS");
  const int fams = fe ? atoi(fe) : 15;
  WgradQueue* const wq_rs = (fams & 1) ? wq : nullptr;
  WgradQueue* const wq_rs4 = (fams & 2) ? wq : nullptr;
  WgradQueue* const wq_1x1f = (fams & 4) ? wq : nullptr;
  WgradQueue* const wq_1x1s = (fams & 8) ? wq : nullptr;   // bit 3: the split-form 1x1 streams (their own table and launch)
  WgradGeom wg;
  wgrad_plan(g, ld_dy, &wg);
  const int T = wgrad_taps(g);
  const size_t lds_stage = ((size_t)g.NI * g.IHt * g.IWt + kBM) * 32 * sizeof(float);
  const size_t lds = lds_stage > 16384 ? lds_stage : 16384;
  if (lds > 160 * 1024 - 512) return fail("wgrad: tile needs %zu B of LDS", lds);
  float* partial = reinterpret_cast<float*>(workspace);
  float* bias_partial = dbias ? partial + (size_t)wg.nsplit * wg.MP * T * wg.NP : nullptr;
  const dim3 grid(wg.nsplit, (wg.MP / 32) * (wg.NP / 32) * wg.ntg, 1);
  static bool attr_done = false;
  if (!attr_done) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_wgrad_kernel<9>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 512);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_wgrad_kernel<1>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 512);
    attr_done = true;
  }
  const bool prof = prof_enabled();
  if (prof) {
    char lab[160];
    snprintf(lab, sizeof(lab), "wgrad B%d %dx%d Cin%d Cout%d k%dx%d nph%d", g.B, g.Hv, g.Wv, g.Cin, g.Cout, g.KH, g.KW, g.nph);
    prof_set_label(lab);
    prof_begin_launch(1, 2.0 * g.B * g.Hv * g.Wv * (double)g.Cout * g.Cin * T, st);
  }
  const bool aligned = ((g.ld0 & 3) == 0) && ((g.ld1 & 3) == 0) && ((g.C0 & 3) == 0) && ((g.Cin & 3) == 0) &&
                       ((ld_dy & 3) == 0) && ((g.Cout & 3) == 0);
  const int smode = wgrad_stream_mode(g, ld_dy);
  if (wgrad_smallc(g) && launch_wgrad_rs7(wg, src0, dy, ld_dy, partial, bias_partial, st, &wg)) {
    if (prof) prof_reclass_last(3);   // split form on the bf16 pipe (row-streaming, k_wgrad_rs.hip)
  } else if (wgrad_smallc(g)) {
    // few input channels, many taps (init conv): (tap, channel) flattened into the GEMM N dimension
    const int NJ = T * g.Cin, ntl = cdiv(NJ, 32), maxn = cdiv(ntl, 4);
    const size_t lds2 = ((((size_t)g.NI * g.IHt * g.IWt * g.Cin + 3) & ~(size_t)3) + kBM * 32) * sizeof(float);
    const dim3 grid2(wg.nsplit, wg.MP / 32, 1);
    static bool attr_s = false;
    if (!attr_s) {
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_wgrad_smallc_kernel<1>), hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_wgrad_smallc_kernel<4>), hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
      attr_s = true;
    }
    if (lds2 > 96 * 1024) return fail("wgrad(small-C): tile needs %zu B of LDS", lds2);
    if (maxn <= 1)
      hipLaunchKernelGGL(HIP_KERNEL_NAME(conv_wgrad_smallc_kernel<1>), grid2, dim3(256), lds2, st, wg, src0, dy, partial, bias_partial);
    else
      hipLaunchKernelGGL(HIP_KERNEL_NAME(conv_wgrad_smallc_kernel<4>), grid2, dim3(256), lds2, st, wg, src0, dy, partial, bias_partial);
  } else if (g.nph > 1 && launch_wgrad_rs4(wg, src0, src1, dy, ld_dy, partial, bias_partial, st, &wg, wq_rs4)) {
    if (prof) prof_reclass_last(3);   // split form on the bf16 pipe
  } else if (g.nph > 1) {
    if (!aligned || g.NI * g.IHt * g.IWt * 8 > 9 * 256) return fail("wgrad: phased 4x4/s2 geometry not eligible for the pipelined kernel");
    const dim3 gridp(wg.nsplit, (wg.MP / 32) * (wg.NP / 32), 4);
#define PIDM_LAUNCH_WG6(KH_, KW_, PH_, WIDE_, MINW_, ROWST_, grid_)                                                        \
  {                                                                                                                        \
    static bool attr_ = false;                                                                                             \
    if (!attr_) {                                                                                                          \
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_wgrad_pipe_kernel<KH_, KW_, PH_, WIDE_, MINW_, ROWST_>), \
                                hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);                                    \
      attr_ = true;                                                                                                        \
    }                                                                                                                      \
    PIDM_PROF_NAME("conv_wgrad_pipe_kernel");                                                                              \
    hipLaunchKernelGGL(HIP_KERNEL_NAME(conv_wgrad_pipe_kernel<KH_, KW_, PH_, WIDE_, MINW_, ROWST_>), grid_, dim3(256), lds, st, wg, \
                       src0, src1 ? src1 : src0, dy, partial, bias_partial);                                              \
  }
#define PIDM_LAUNCH_WG(KH_, KW_, PH_, WIDE_, MINW_, grid_) PIDM_LAUNCH_WG6(KH_, KW_, PH_, WIDE_, MINW_, false, grid_)
    if (g.Wv >= 32) PIDM_LAUNCH_WG(2, 2, true, true, 2, gridp)
    else PIDM_LAUNCH_WG(2, 2, true, false, 2, gridp)
  } else if (smode >= 1 && smode <= 3) {
    // the LDS-free pixel streams; grouped (wq): queued with a quarter of the splits (launch_wgrad_rs)
    // wide operands (>= 128 channels) on the bf16 pipe: conv_wgrad_1x1_split4_body (PIDM_WGRAD1X1_SPLIT=0 or PIDM_WGRAD_SPLIT=0: the
    // fp32 streams)
    const bool sp4 = wgrad_1x1_split_on(g, ld_dy);
    WgradQueue* const wq_1x1 = sp4 ? wq_1x1s : wq_1x1f;
    if (wq_1x1) {
      const char* tme = knob("PIDM_WGRAD1X1_SPLIT_DIV");      // the split-form family's own divisor of the planned split count
      const int dv = (sp4 && tme && atoi(tme) > 0) ? atoi(tme) : wgrad_group_splitdiv();
      if (dv > 1) {
        wg.tiles_per_split *= dv;
        if (wg.tiles_per_split > g.tiles_m) wg.tiles_per_split = g.tiles_m;
        wg.nsplit = cdiv(g.tiles_m, wg.tiles_per_split);
      }
    }
    const char* spe = knob("PIDM_WGRAD1X1_SPLIT_PLAN");
    if (sp4 && !wq_1x1 && !(spe && !atoi(spe))) {
      // a launch of its own (large batches): the split form has two workgroups per CU, not the plan's four, and a longer prologue /
      // epilogue per work item - fewer, longer splits (never more than planned: the workspace is sized for the plan).  Batch 256:
      // 466 -> 386 us per step over the 16 launches, their reductions 330 -> 310 (profiles/r06_wgrad1x1_split.txt)
      const int bmn = (smode == 2) ? cdiv(wg.MP, 128) * (wg.NP / 32) : cdiv(wg.NP, 128) * (wg.MP / 32);
      int best = wg.tiles_per_split;
      double bc = 1e30;
      for (int tps = wg.tiles_per_split; tps <= g.tiles_m; ++tps) {
        const long wgs = (long)bmn * cdiv(g.tiles_m, tps);
        const double cost = (double)((wgs + 511) / 512) * (tps + 2.0);
        if (cost < bc - 1e-9) { bc = cost; best = tps; }
      }
      wg.tiles_per_split = best;
      wg.nsplit = cdiv(g.tiles_m, best);
    }
    const dim3 gr = smode == 1   ? dim3(wg.nsplit, (wg.MP / 32) * (wg.NP / 32), 1)
                    : smode == 2 ? dim3(cdiv(wg.MP, 128) * (wg.NP / 32), wg.nsplit, 1)
                                 : dim3(cdiv(wg.NP, 128) * (wg.MP / 32), wg.nsplit, 1);
    const WgradItem it = wgrad_item(wg, src0, src1, dy, partial, bias_partial, gr.x, gr.y,
                                    smode == 1 ? kWgKindStream : smode == 2 ? kWgKindStream4Dy : kWgKindStream4X);
    if (knob("PIDM_TRACE_CONV"))
      fprintf(stderr, "[pidm]   -> 1x1 wgrad stream mode %d: %d pixels, Cin %d (+%d), Cout %d, grid %u x %u, %d tiles per split\n", smode,
              g.B * g.Hv * g.Wv, g.C0, g.C1, g.Cout, gr.x, gr.y, wg.tiles_per_split);
    if (wq_1x1) {
      wq_1x1->push(sp4 ? kWgFam1x1Split : kWgFam1x1, it, sp4 ? 2.0 * g.B * g.Hv * g.Wv * (double)g.Cout * g.Cin : 0.0);
    } else if (sp4) {
      if (smode == 2) {
        PIDM_PROF_NAME("conv_wgrad_1x1_split4_kernel<true>");
        hipLaunchKernelGGL(HIP_KERNEL_NAME(conv_wgrad_1x1_split4_kernel<true>), gr, dim3(256), 0, st, it, wgrad_1x1_xcd());
      } else {
        PIDM_PROF_NAME("conv_wgrad_1x1_split4_kernel<false>");
        hipLaunchKernelGGL(HIP_KERNEL_NAME(conv_wgrad_1x1_split4_kernel<false>), gr, dim3(256), 0, st, it, wgrad_1x1_xcd());
      }
      if (prof) prof_reclass_last(3);   // split form on the bf16 pipe
    } else if (smode == 1) {
      PIDM_PROF_NAME("conv_wgrad_1x1_stream_kernel");
      hipLaunchKernelGGL(conv_wgrad_1x1_stream_kernel, gr, dim3(256), 0, st, it);
    } else if (smode == 2) {
      PIDM_PROF_NAME("conv_wgrad_1x1_stream4_kernel<true>");
      hipLaunchKernelGGL(HIP_KERNEL_NAME(conv_wgrad_1x1_stream4_kernel<true>), gr, dim3(256), 0, st, it);
    } else {
      PIDM_PROF_NAME("conv_wgrad_1x1_stream4_kernel<false>");
      hipLaunchKernelGGL(HIP_KERNEL_NAME(conv_wgrad_1x1_stream4_kernel<false>), gr, dim3(256), 0, st, it);
    }
  } else if (aligned && ((g.KH == 3 && g.KW == 3) || (g.KH == 1 && g.KW == 1)) && wg.ntg == 1 &&
             g.NI * g.IHt * g.IWt * 8 <= (g.KH == 1 ? 4 : 9) * 256 && g.IHt < 1024 && g.IWt < 1024) {
    if (g.KH == 1) {
      if (g.Wv >= 32) PIDM_LAUNCH_WG(1, 1, false, true, 4, grid)
      else PIDM_LAUNCH_WG(1, 1, false, false, 3, grid)   // per-step pixel decode: 3 waves per SIMD without spilling
    } else if (launch_wgrad_rs(wg, src0, src1, dy, ld_dy, partial, bias_partial, st, &wg, wq_rs) ||
               launch_wgrad_split(wg, src0, src1, dy, ld_dy, partial, bias_partial, st, &wg)) {
      if (prof) prof_reclass_last(3);   // split form: counted with the weight gradients and, separately, against the bf16 pipe
      // taken by the bf16-pipe kernel (wg now holds its tiling / split)
    } else {
      // row-aligned staging without vector arithmetic where the geometry allows it (PIDM_WGRAD_ROWST=0: off, for A/B runs)
      const char* re = knob("PIDM_WGRAD_ROWST");
      const int seg = g.NI * g.IHt * g.Wv;
      const bool rowst = !(re && !atoi(re)) && g.stride == 1 && g.pad_y[0] == 1 && g.pad_x[0] == 1 && g.Wv == g.Wi && g.Wv >= 8 &&
                         seg % 32 == 0 && seg <= 256 && (g.Cin % 32 == 0) && (g.C0 % 32 == 0) && (g.Cout % 32 == 0) &&
                         (g.C1 == 0 || g.ld1 == g.ld0) && (ld_dy & 3) == 0;
      if (rowst) {
        if (g.Wv >= 32) PIDM_LAUNCH_WG6(3, 3, false, true, 1, true, grid)
        else PIDM_LAUNCH_WG6(3, 3, false, false, 1, true, grid)
      } else {
        if (g.Wv >= 32) PIDM_LAUNCH_WG(3, 3, false, true, 1, grid)
        else PIDM_LAUNCH_WG(3, 3, false, false, 1, grid)
      }
    }
#undef PIDM_LAUNCH_WG
#undef PIDM_LAUNCH_WG6
  } else if (wg.tgs == 1)
    hipLaunchKernelGGL(HIP_KERNEL_NAME(conv_wgrad_kernel<1>), grid, dim3(256), lds, st, wg, src0, src1 ? src1 : src0, dy, partial, bias_partial);
  else
    hipLaunchKernelGGL(HIP_KERNEL_NAME(conv_wgrad_kernel<9>), grid, dim3(256), lds, st, wg, src0, src1 ? src1 : src0, dy, partial, bias_partial);
  const bool queued = wq && wq->size() > queued0;        // nothing was enqueued: the problem waits for flush_wgrads
  if (queued) {
    // the grouped launch accounts for this row (flush_wgrads declares the FLOPs of its items to the profiling hooks)
    for (auto& q : wq->f)
      if (!q.fl.empty() && q.fl.size() == q.v.size() && q.v.back().partial == partial) q.fl.back() = 2.0 * g.B * g.Hv * g.Wv * (double)g.Cout * g.Cin * T;
    if (prof) prof_cancel_last();
  } else {
    if (prof) prof_end_launch(st);
    PIDM_CHECK_LAUNCH("conv_wgrad_kernel");
  }
  if (defer) {   // the caller keeps `workspace` alive until its reduce_multi launch
    defer->push(partial, dw_ref, bias_partial, dbias, (size_t)wg.MP * T * wg.NP, wg.nsplit, g.Cout, g.Cin, T, wg.MP, wg.NP);
    return 0;
  }
  const size_t total = (size_t)g.Cout * g.Cin * T + (dbias ? g.Cout : 0);
  const int blocks = (int)((total + 31) / 32);
  hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(blocks), dim3(256), 0, st, partial, dw_ref, bias_partial, dbias, wg.nsplit, g.Cout,
                     g.Cin, T, wg.MP, wg.NP);
  PIDM_CHECK_LAUNCH("wgrad_reduce_kernel");
  return 0;
}

static size_t colsum_blocks(size_t rows) {
  size_t nblk = rows / 64;          // >= 64 rows (16 per thread row-lane) per block, at most 1024 blocks
  if (nblk < 1) nblk = 1;
  if (nblk > 1024) nblk = 1024;
  return nblk;
}

size_t colsum_ws_bytes(size_t rows, int C) { return colsum_blocks(rows) * (size_t)C * sizeof(float); }

// `defer` != null: only the per-block partial sums are produced here (into `workspace`, which the caller keeps alive); the
// fixed-order sum over blocks joins the caller's single reduce_multi launch.
int launch_colsum(const float* x, size_t rows, int C, int ld, float* out, void* workspace, hipStream_t st, ReduceQueue* defer) {
  size_t nblk = colsum_blocks(rows);
  const size_t rpb = (rows + nblk - 1) / nblk;
  nblk = (rows + rpb - 1) / rpb;
  float* partial = reinterpret_cast<float*>(workspace);
  hipLaunchKernelGGL(colsum_partial_kernel, dim3((unsigned)nblk, cdiv(C, 64)), dim3(256), 0, st, x, rows, C, ld, rpb, partial);
  PIDM_CHECK_LAUNCH("colsum_partial_kernel");
  if (defer) {
    defer->push(partial, out, nullptr, nullptr, (size_t)C, (int)nblk, 1, C, 1, 1, C);
    return 0;
  }
  hipLaunchKernelGGL(colsum_final_kernel, dim3(cdiv(C, 256)), dim3(256), 0, st, partial, (int)nblk, C, out);
  PIDM_CHECK_LAUNCH("colsum_final_kernel");
  return 0;
}

}  // namespace pidm

using namespace pidm;

extern "C" size_t pidm_conv_wgrad_ws(const pidm_conv_desc* d) {
  ConvGeom g;
  if (d->transposed) {
    if (make_geom(&g, 0, d->B, 2 * d->Hi, 2 * d->Wi, d->Cout, 0, d->Cout, 0, d->C0 + d->C1, 4, 4, 2, 1, 0, 4, 4)) return 0;
  } else if (geom_fwd(d, &g)) {
    return 0;
  }
  return wgrad_ws_bytes(g) + colsum_ws_bytes((size_t)d->B * g.Ho * g.Wo * 4, d->Cout);
}

extern "C" int pidm_conv_wgrad(const pidm_conv_desc* d, const float* src0, const float* src1, const float* dy, int ld_dy,
                               float* dw_ref, float* dbias, void* workspace, void* stream) {
  ConvGeom g;
  hipStream_t st = as_stream(stream);
  int rc;
  size_t dy_rows;
  if (d->transposed) {
    // swapped operands: X' = dy [B,2H,2W,Cout], dY' = x [B,H,W,Cin]; result [Cin][Cout][4][4]
    if (d->C1) return fail("wgrad: transposed conv with two sources is not supported");
    if (make_geom(&g, 0, d->B, 2 * d->Hi, 2 * d->Wi, d->Cout, 0, ld_dy, 0, d->C0, 4, 4, 2, 1, 0, 4, 4)) return -1;
    rc = launch_wgrad(g, dy, nullptr, src0, d->ld0, dw_ref, nullptr, workspace, st);
    dy_rows = (size_t)d->B * 4 * d->Hi * d->Wi;
  } else {
    if (geom_fwd(d, &g)) return -1;
    return launch_wgrad(g, src0, src1, dy, ld_dy, dw_ref, dbias, workspace, st);
  }
  if (rc) return rc;
  if (dbias) {
    char* ws2 = reinterpret_cast<char*>(workspace) + wgrad_ws_bytes(g);
    return launch_colsum(dy, dy_rows, d->Cout, ld_dy, dbias, ws2, st);
  }
  return 0;
}
