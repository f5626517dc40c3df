// Linear attention WITHOUT a qkv tensor ("projected" form), forward + backward, for the full-resolution levels.
//
// Replaces Residual(PreNorm(SpatialLinearAttention)) between the LayerNorm and the residual add (reference
// src/unet_model.py:281-299 + 139-145) and everything autograd replays through it.  The reference materialises
// qkv = to_qkv(xn): 3*heads*32 channels per pixel (805 MB per 64x64 instance at batch 64), written once and read five
// times forward + backward, plus its gradient.  Here q and k are 32x32 (pixel x head-channel) tiles rebuilt on the
// matrix cores from the 32..64-channel xn tile wherever a kernel needs them (C/2 MFMAs per tile and head), and v is never
// formed per pixel at all - linearity moves the to_v projection (and to_out) out of the pixel sums:
//
//   ks[n][d] = softmax_n( xn[n][:] . Wk[d][:] )                      M[d][c]   = sum_n ks[n][d] xn[n][c]
//   ctx[d][e] = 1/N sum_c M[d][c] Wv[e][c]    (= sum_n ks v / N)     P[d][c']  = sum_e ctx[d][e] Wout[c'][h*32+e]
//   qs[n][d] = scale * softmax_d( xn[n][:] . Wq[d][:] )              y[n][c']  = bias[c'] + x[n][c'] + sum_h sum_d qs_h[n][d] P_h[d][c']
//
// and backward (dY = d loss / d y; the residual's own share dY -> dx is added by the LayerNorm backward kernel):
//   G_h[d][c']  = sum_n qs_h[n][d] dY[n][c']        (= dP_h)          dctx = G Wout_h, dWout += G^T ctx, dM = dctx Wv / N,
//   dWv += dctx^T M / N,  rowdot[d] = sum_c dM[d][c] M[d][c]          (= sum_n ks[n][d] dks[n][d]: closed form, no extra pass)
//   dqs[n][d] = sum_c' dY[n][c'] P[d][c'],  dq = qs (dqs - <qs, dqs>/scale)
//   dks[n][d] = sum_c xn[n][c] dM[d][c],    dk = ks (dks - rowdot[d])
//   d_xn[n][c] = sum_h sum_d ( dq Wq + dk Wk + ks dM ),   dWq += dq^T xn,  dWk += dk^T xn        (pixel sums)
//
// Matrix-core layouts (v_mfma_f32_32x32x2_f32: D[i][j] with j = lane & 31, rows i = (r&3) + 8(r>>2) + 4(lane>>5), r < 16):
//   * pixel sums (M, G, dWq, dWk) contract over pixels, so their left operand needs lane = d: the k / q tile is built as
//     D[px][d] (A = xn rows, B = W^T) and fed back as the A operand in its own accumulator-row order ("transposed chaining");
//   * per-pixel work (softmax over d, Jacobians, y, d_xn) wants one pixel per lane: tiles are built as D[d][px] (A = W, B = xn^T),
//     reductions over d are in-lane plus one cross-half exchange, and results chain into the next product as B operands;
//   * dq / dk are needed in both roles: they are produced with lane = px and turned through a wave-private 32x33 LDS tile.
// Kernels: lap_kctx (+ _final), lap_out (forward); lap_g, lap_mid, lap_bwd (backward).  One workgroup = one image's pixel
// range, one wave per head in the pixel-sum kernels (8 waves), one wave per 32-pixel tile in lap_out.
// Supported: C = Cout in {32, 64}, N % 32 == 0, heads <= 8, dim_head = 32.  Everything else keeps the qkv form (k_attn.hip).
#include "pidm_launch.h"

namespace pidm {

static const int kLapDH = 32;
// threads of the two per-(image, head) merge kernels (lap_kctx_final, lap_mid): B * heads workgroups walk five or six dependent
// phases of 32-term dot products; with 1024 threads every output element has its own thread and a phase is one chain long
constexpr int kLapSmallNT = 1024;
static const int kLapTileLd = 33;    // row stride of a wave's 32x32 transposition tile (conflict-free in both directions)

__device__ __forceinline__ float lap_exp(float x) { return __expf(x); }
__device__ __forceinline__ int lap_row(int r, int half) { return (r & 3) + 8 * (r >> 2) + 4 * half; }

// pixels per workgroup of the pixel-sum kernels: the xn (and dY) slab of the range lives in LDS, rows padded to C + 4 floats
// PIDM_LAP_NPER_DIV (1, 2, 4; read per call): the pixel ranges of the split pixel-sum kernels (lap_kctx_split, lap_g_split) divided
// by this - half the LDS per workgroup, two workgroups per CU (one stages while the other computes), twice the range partials
static int lap_nper_div() {
  const char* e = knob("PIDM_LAP_NPER_DIV");
  const int v = e ? atoi(e) : 1;
  return (v == 2 || v == 4) ? v : 1;
}
static int lap_nper(int N, int C, int slabs) {
  int n = (slabs == 1 ? 512 : 256) * 32 / C;
  if (slabs == 3) n = (C == 32) ? 32 : n / 2;   // lap_bwd: C = 32 keeps Wq | Wk | dM of all heads in LDS as well (one tile per staging)
  if (slabs == 2 && n / lap_nper_div() >= 64) n /= lap_nper_div();
  while (n > 32 && N % n) n >>= 1;
  return n;
}

// ---------------------------------------------------------------------------------------------------------------------
// forward 1: per (image, pixel range, head): running column max m[d], Z[d] = sum_n exp(k - m), Mt[d][c] = sum_n exp(k - m) xn[n][c]
// part layout per (b, ns, h): [32][C] Mt rows, then m[32], then Z[32]
// ---------------------------------------------------------------------------------------------------------------------
template <int CB>
__global__ void __launch_bounds__(512) lap_kctx_kernel(const float* __restrict__ xn, const float* __restrict__ wqkv,
                                                       float* __restrict__ part, int N, int heads, int nper) {
  constexpr int C = 32 * CB, CP = C + 4;
  HIP_DYNAMIC_SHARED(float, smem)
  float* xs = smem;                          // [nper][CP]
  float* fs = smem + (size_t)nper * CP;      // [8][32] rescale factors (wave private)
  const int tid = threadIdx.x, lane = tid & 63, half = lane >> 5, l31 = lane & 31;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int NS = N / nper;
  const int b = blockIdx.x / NS, ns = blockIdx.x % NS;
  const int HD = heads * kLapDH;
  const float* xb = xn + ((size_t)b * N + (size_t)ns * nper) * C;
  for (int e = tid; e < nper * (C / 4); e += 512) {
    const int px = e / (C / 4), q = e - px * (C / 4);
    *reinterpret_cast<f32x4*>(xs + (size_t)px * CP + 4 * q) = *reinterpret_cast<const f32x4*>(xb + (size_t)px * C + 4 * q);
  }
  __syncthreads();
  const int h = wave;
  if (h >= heads) return;
  // B operand of the k tile: W_k^T[c][d], lane = d; k-slots permuted so that one 16-byte fetch feeds four MFMAs
  f32x4 wk[C / 8];
  const float* wrow = wqkv + ((size_t)HD + h * kLapDH + l31) * C + 4 * half;
#pragma unroll
  for (int g8 = 0; g8 < C / 8; ++g8) wk[g8] = *reinterpret_cast<const f32x4*>(wrow + 8 * g8);
  f32x16 M[CB];
#pragma unroll
  for (int cb = 0; cb < CB; ++cb)
    for (int r = 0; r < 16; ++r) M[cb][r] = 0.f;
  float mrun = -3.0e38f, zp = 0.f;
  float* fw = fs + wave * 32;
  for (int t = 0; t < nper / 32; ++t) {
    f32x16 kt;
    for (int r = 0; r < 16; ++r) kt[r] = 0.f;
    const float* arow = xs + (size_t)(t * 32 + l31) * CP + 4 * half;
#pragma unroll
    for (int g8 = 0; g8 < C / 8; ++g8) {
      const f32x4 a4 = *reinterpret_cast<const f32x4*>(arow + 8 * g8);
#pragma unroll
      for (int s = 0; s < 4; ++s) kt = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[s], wk[g8][s], kt, 0, 0, 0);
    }
    // kt[px][d]: lane = d, registers = 16 of the tile's pixels.  Online softmax over pixels: column max of the tile first
    float tm = kt[0];
#pragma unroll
    for (int r = 1; r < 16; ++r) tm = fmaxf(tm, kt[r]);
    tm = fmaxf(tm, pidm_other_half(tm));
    if (__any(tm > mrun)) {     // wave-uniform: the max of some column grew - rescale what has been accumulated (rare after the first tiles)
      const float mn = fmaxf(mrun, tm);
      const float f = lap_exp(mrun - mn);
      zp *= f;
      mrun = mn;
      if (half == 0) fw[l31] = f;
      PIDM_WAVE_LDS_SYNC();
      // Mt rows live across registers: row d = lap_row(r, half) needs factor f[d]
#pragma unroll
      for (int q4 = 0; q4 < 4; ++q4) {
        const f32x4 f4 = *reinterpret_cast<const f32x4*>(fw + 8 * q4 + 4 * half);
#pragma unroll
        for (int cb = 0; cb < CB; ++cb)
#pragma unroll
          for (int i = 0; i < 4; ++i) M[cb][4 * q4 + i] *= f4[i];
      }
      PIDM_WAVE_LDS_SYNC();
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      kt[r] = lap_exp(kt[r] - mrun);
      zp += kt[r];
    }
    // Mt[d][c] += sum_px e[px][d] xn[px][c]: A = the exponentials in their own accumulator-row order, B = xn rows from LDS
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const float* brow = xs + (size_t)(t * 32 + lap_row(r, half)) * CP + l31;
#pragma unroll
      for (int cb = 0; cb < CB; ++cb) M[cb] = __builtin_amdgcn_mfma_f32_32x32x2f32(kt[r], brow[32 * cb], M[cb], 0, 0, 0);
    }
  }
  const float z = zp + pidm_other_half(zp);
  float* o = part + ((size_t)blockIdx.x * heads + h) * (size_t)(32 * C + 64);
#pragma unroll
  for (int cb = 0; cb < CB; ++cb)
#pragma unroll
    for (int r = 0; r < 16; ++r) o[(size_t)lap_row(r, half) * C + 32 * cb + l31] = M[cb][r];
  if (half == 0) {
    o[32 * C + l31] = mrun;
    o[32 * C + 32 + l31] = z;
  }
}

// forward 1 in the split form (pidm_common.h: fp32 operands as three bf16 pieces, 6 bf16 MFMAs per fp32 product): the two products
// of a tile and head are 24 MFMAs on the bf16 pipe instead of 32 on the fp32 MFMA, which runs on the vector ALUs and serialises
// with the softmax arithmetic.  The xn slab is staged ONCE per workgroup in both operand layouts (shared by the 8 heads):
//   XA[px][piece][c]   (c contiguous: A operand of k = xn Wk^T, lane = pixel)
//   XT[piece][c][px]   (pixels contiguous: B operand of Mt += e^T xn, lane = channel; 4x4 register transposes at staging)
// The exponentials leave the first product as D[px][d] with lane = d and 16 pixels per lane in accumulator-row order; they are
// the A operand of the second product in exactly that order - k-step s takes rows 8s..8s+7, i.e. pixels 16s + 4 half + {0..3, 8..11}
// - and the B operand follows with two 8-byte reads of XT per piece.
static int lap_nper_split(int N, int C) {
  int n = 256 * 32 / C;
  if (n / lap_nper_div() >= 64) n /= lap_nper_div();
  while (n > 32 && N % n) n >>= 1;
  return n;
}
// Tile groups of lap_bwd: the pixel-sum kernels run one wave per head, so with heads <= 4 (attn_heads below the model's default of 8)
// half of the workgroup's 8 wave slots would idle after staging while the others run their dependent MFMA chains alone on their SIMD.
// With G groups wave w takes head w % heads and every G-th tile of the slab (group w / heads); the groups' weight-gradient shares are
// summed through LDS at the end.  Measured at 4 heads, batch 64: lap_bwd 224 -> 181 us (64x64, C = 32), 220 -> 178 us (32x32, C = 64).
// lap_kctx_split / lap_g_split stay at one wave per head: groups there mean G partials per range, and what the pixel-sum kernels
// gain (2 - 4 us) their merge kernels lose twice over (+8 us lap_kctx_final, +19 us lap_mid).
static int lap_groups(int heads, int ntiles) {
  const char* e = knob("PIDM_LAP_GROUPS");      // 1: one wave per head, the other slots idle (A/B measurements)
  int g = 8 / heads;
  if (e && atoi(e) > 0 && atoi(e) < g) g = atoi(e);
  while (g > 1 && ntiles % g) --g;
  return g < 1 ? 1 : g;
}
template <int CB>
__global__ void __launch_bounds__(512) lap_kctx_split_kernel(const float* __restrict__ xn, const float* __restrict__ wqkv,
                                                             float* __restrict__ part, int N, int heads, int nper) {
  constexpr int C = 32 * CB, RA = 6 * C + 16;     // bytes per XA row (pixel)
  HIP_DYNAMIC_SHARED(float, smemf)
  char* smem = reinterpret_cast<char*>(smemf);
  const int RT = nper * 2 + 16;                   // bytes per XT row (piece, channel)
  char* XA = smem;
  char* XT = smem + (size_t)nper * RA;
  float* fs = reinterpret_cast<float*>(XT + (size_t)3 * C * RT);     // [8][32] rescale factors (wave private)
  const int tid = threadIdx.x, lane = tid & 63, half = lane >> 5, l31 = lane & 31;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int NS = N / nper;
  const int b = blockIdx.x / NS, ns = blockIdx.x % NS;
  const int HD = heads * kLapDH;
  const float* xb = xn + ((size_t)b * N + (size_t)ns * nper) * C;
  {
    // staging: a wave-slot = (64 / (C/4)) groups of 4 pixels x all channel quads; lane = (quad, group bit, pixel bits 4-5)
    constexpr int QN = C / 4, GP = 16 / QN;       // quads per pixel, 4-pixel groups per wave-slot (2 for C = 32, 1 for C = 64)
    const int q = lane & (QN - 1), pb = (lane & 15) / QN, la = lane >> 4;
    for (int sl = wave; sl < nper / (4 * GP); sl += 8) {
      const int px = (sl * GP + pb) * 4 + la;
      const f32x4 v = *reinterpret_cast<const f32x4*>(xb + (size_t)px * C + 4 * q);
      typedef unsigned u32x2_t __attribute__((ext_vector_type(2)));
      unsigned a0, a1, a2, b0, b1, b2;
      pidm_split3_pk(v[0], v[1], a0, a1, a2);
      pidm_split3_pk(v[2], v[3], b0, b1, b2);
      char* da = XA + (size_t)px * RA + 8 * q;
      *reinterpret_cast<u32x2_t*>(da) = u32x2_t{a0, b0};
      *reinterpret_cast<u32x2_t*>(da + 2 * C) = u32x2_t{a1, b1};
      *reinterpret_cast<u32x2_t*>(da + 4 * C) = u32x2_t{a2, b2};
      unsigned t0 = __float_as_uint(v[0]), t1 = __float_as_uint(v[1]), t2 = __float_as_uint(v[2]), t3 = __float_as_uint(v[3]);
      {
        const auto s02 = __builtin_amdgcn_permlane32_swap(t0, t2, false, false);
        const auto s13 = __builtin_amdgcn_permlane32_swap(t1, t3, false, false);
        const auto s01 = __builtin_amdgcn_permlane16_swap(s02[0], s13[0], false, false);
        const auto s23 = __builtin_amdgcn_permlane16_swap(s02[1], s13[1], false, false);
        t0 = s01[0]; t1 = s01[1]; t2 = s23[0]; t3 = s23[1];
      }
      // now: channel 4 q + la, pixels (sl GP + pb) 4 + 0..3
      pidm_split3_pk(__uint_as_float(t0), __uint_as_float(t1), a0, a1, a2);
      pidm_split3_pk(__uint_as_float(t2), __uint_as_float(t3), b0, b1, b2);
      char* dt = XT + (size_t)(4 * q + la) * RT + (size_t)((sl * GP + pb) * 4) * 2;
      *reinterpret_cast<u32x2_t*>(dt) = u32x2_t{a0, b0};
      *reinterpret_cast<u32x2_t*>(dt + (size_t)C * RT) = u32x2_t{a1, b1};
      *reinterpret_cast<u32x2_t*>(dt + (size_t)2 * C * RT) = u32x2_t{a2, b2};
    }
  }
  __syncthreads();
  const int h = wave;
  if (h >= heads) return;
  // B operand of the k tile: Wk_h[d][c], lane = d, k-step s = channels 16 s + 8 half .. + 7; pre-split once
  u32x4 wk[C / 16][3];
  {
    const float* wrow = wqkv + ((size_t)HD + h * kLapDH + l31) * C + 8 * half;
#pragma unroll
    for (int s = 0; s < C / 16; ++s) {
      const f32x4 w0 = *reinterpret_cast<const f32x4*>(wrow + 16 * s), w1 = *reinterpret_cast<const f32x4*>(wrow + 16 * s + 4);
      unsigned p0[4], p1[4], p2[4];
      pidm_split3_pk(w0[0], w0[1], p0[0], p1[0], p2[0]);
      pidm_split3_pk(w0[2], w0[3], p0[1], p1[1], p2[1]);
      pidm_split3_pk(w1[0], w1[1], p0[2], p1[2], p2[2]);
      pidm_split3_pk(w1[2], w1[3], p0[3], p1[3], p2[3]);
      wk[s][0] = u32x4{p0[0], p0[1], p0[2], p0[3]};
      wk[s][1] = u32x4{p1[0], p1[1], p1[2], p1[3]};
      wk[s][2] = u32x4{p2[0], p2[1], p2[2], p2[3]};
    }
  }
  f32x16 M[CB];
#pragma unroll
  for (int cb = 0; cb < CB; ++cb)
    for (int r = 0; r < 16; ++r) M[cb][r] = 0.f;
  float mrun = -3.0e38f, zp = 0.f;
  float* fw = fs + wave * 32;
#define PIDM_LAP_SIX(acc_, a_, b_)                                                                                 \
  acc_ = pidm_mfma_bf16_32x32x16(a_[2], b_[0], acc_);                                                              \
  acc_ = pidm_mfma_bf16_32x32x16(a_[0], b_[2], acc_);                                                              \
  acc_ = pidm_mfma_bf16_32x32x16(a_[1], b_[1], acc_);                                                              \
  acc_ = pidm_mfma_bf16_32x32x16(a_[1], b_[0], acc_);                                                              \
  acc_ = pidm_mfma_bf16_32x32x16(a_[0], b_[1], acc_);                                                              \
  acc_ = pidm_mfma_bf16_32x32x16(a_[0], b_[0], acc_);
  for (int t = 0; t < nper / 32; ++t) {
    f32x16 kt;
    for (int r = 0; r < 16; ++r) kt[r] = 0.f;
    const char* arow = XA + (size_t)(t * 32 + l31) * RA + 16 * half;
#pragma unroll
    for (int s = 0; s < C / 16; ++s) {
      u32x4 xa[3];
#pragma unroll
      for (int pc = 0; pc < 3; ++pc) xa[pc] = *reinterpret_cast<const u32x4*>(arow + pc * 2 * C + 32 * s);
      PIDM_LAP_SIX(kt, xa, wk[s])
    }
    // kt[px][d]: lane = d, registers = 16 of the tile's pixels.  Online softmax over pixels: column max of the tile first
    float tm = kt[0];
#pragma unroll
    for (int r = 1; r < 16; ++r) tm = fmaxf(tm, kt[r]);
    tm = fmaxf(tm, pidm_other_half(tm));
    if (__any(tm > mrun)) {
      const float mn = fmaxf(mrun, tm);
      const float f = lap_exp(mrun - mn);
      zp *= f;
      mrun = mn;
      if (half == 0) fw[l31] = f;
      PIDM_WAVE_LDS_SYNC();
#pragma unroll
      for (int q4 = 0; q4 < 4; ++q4) {
        const f32x4 f4 = *reinterpret_cast<const f32x4*>(fw + 8 * q4 + 4 * half);
#pragma unroll
        for (int cb = 0; cb < CB; ++cb)
#pragma unroll
          for (int i = 0; i < 4; ++i) M[cb][4 * q4 + i] *= f4[i];
      }
      PIDM_WAVE_LDS_SYNC();
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      kt[r] = lap_exp(kt[r] - mrun);
      zp += kt[r];
    }
    // Mt[d][c] += sum_px e[px][d] xn[px][c]: A = the exponentials (rows 8 s .. 8 s + 7 of the accumulator = pixels
    // 16 s + 4 half + {0..3, 8..11}), B = XT rows read in the same pixel order
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      u32x4 ea[3];
      {
        unsigned p0[4], p1[4], p2[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) pidm_split3_pk(kt[8 * s + 2 * j], kt[8 * s + 2 * j + 1], p0[j], p1[j], p2[j]);
        ea[0] = u32x4{p0[0], p0[1], p0[2], p0[3]};
        ea[1] = u32x4{p1[0], p1[1], p1[2], p1[3]};
        ea[2] = u32x4{p2[0], p2[1], p2[2], p2[3]};
      }
#pragma unroll
      for (int cb = 0; cb < CB; ++cb) {
        u32x4 xb4[3];
        const char* brow = XT + (size_t)(32 * cb + l31) * RT + (size_t)(t * 32 + 16 * s + 4 * half) * 2;
#pragma unroll
        for (int pc = 0; pc < 3; ++pc) {
          typedef unsigned u32x2_t __attribute__((ext_vector_type(2)));
          const u32x2_t lo = *reinterpret_cast<const u32x2_t*>(brow + (size_t)pc * C * RT);
          const u32x2_t hi = *reinterpret_cast<const u32x2_t*>(brow + (size_t)pc * C * RT + 16);
          xb4[pc] = u32x4{lo[0], lo[1], hi[0], hi[1]};
        }
        PIDM_LAP_SIX(M[cb], ea, xb4)
      }
    }
  }
#undef PIDM_LAP_SIX
  const float z = zp + pidm_other_half(zp);
  float* o = part + ((size_t)blockIdx.x * heads + h) * (size_t)(32 * C + 64);
#pragma unroll
  for (int cb = 0; cb < CB; ++cb)
#pragma unroll
    for (int r = 0; r < 16; ++r) o[(size_t)lap_row(r, half) * C + 32 * cb + l31] = M[cb][r];
  if (half == 0) {
    o[32 * C + l31] = mrun;
    o[32 * C + 32 + l31] = z;
  }
}

// forward 1b, per (image, head): merge the pixel ranges; M = Mt / Z, kst = (m, 1/Z), ctx = M Wv^T / N, P = ctx Wout_h^T
__global__ void __launch_bounds__(kLapSmallNT) lap_kctx_final_kernel(const float* __restrict__ part, const float* __restrict__ wqkv,
                                                             const float* __restrict__ wout, float* __restrict__ kst,
                                                             float* __restrict__ Mmat, float* __restrict__ ctx,
                                                             float* __restrict__ P, int N, int heads, int C, int NS) {
  HIP_DYNAMIC_SHARED(float, smem)
  float* sM = smem;                 // [32][C]
  float* sW = sM + 32 * C;          // [32][C+1] Wv_h rows, later Wout[:, h*32 .. +32] as [C][33]
  float* sC = sW + 32 * (C + 1) + C;   // [32][33] ctx
  float* sw = sC + 32 * 33;         // [NS][32] merge weights
  float* sst = sw + NS * 32;        // [NS][64] the ranges' (m, Z) rows
  __shared__ float sm_[32], siz[32];
  constexpr int NT = kLapSmallNT;
  const int b = blockIdx.x / heads, h = blockIdx.x % heads, tid = threadIdx.x;
  const int HD = heads * kLapDH;
  const size_t pstride = (size_t)(32 * C + 64);
  const float* p0 = part + ((size_t)b * NS * heads + h) * pstride;
  // all threads fetch the ranges' statistics rows (2 x 32 floats each) at once; the 32 merging threads then walk them in LDS
  for (int e = tid; e < NS * 64; e += NT) sst[e] = p0[(size_t)(e >> 6) * heads * pstride + 32 * C + (e & 63)];
  __syncthreads();
  if (tid < 32) {
    float m = -3.0e38f;
    for (int k = 0; k < NS; ++k) m = fmaxf(m, sst[k * 64 + tid]);
    float z = 0.f;
    for (int k = 0; k < NS; ++k) {
      const float w = lap_exp(sst[k * 64 + tid] - m);
      sw[k * 32 + tid] = w;
      z += sst[k * 64 + 32 + tid] * w;
    }
    sm_[tid] = m;
    siz[tid] = 1.f / z;
    kst[((size_t)blockIdx.x * 32 + tid) * 2] = m;
    kst[((size_t)blockIdx.x * 32 + tid) * 2 + 1] = 1.f / z;
  }
  __syncthreads();
  for (int e = tid; e < 32 * C; e += NT) {
    const int d = e / C;
    float v = 0.f;
    for (int k = 0; k < NS; ++k) v += p0[(size_t)k * heads * pstride + e] * sw[k * 32 + d];
    v *= siz[d];
    sM[e] = v;
    Mmat[(size_t)blockIdx.x * 32 * C + e] = v;
    const int c = e - d * C;
    sW[d * (C + 1) + c] = wqkv[((size_t)2 * HD + h * kLapDH + d) * C + c];     // Wv_h[e = d][c]
  }
  __syncthreads();
  const float invN = 1.f / (float)N;
  for (int e = tid; e < 1024; e += NT) {
    const int d = e >> 5, ee = e & 31;
    float v = 0.f;
    for (int c = 0; c < C; ++c) v = fmaf(sM[d * C + c], sW[ee * (C + 1) + c], v);
    v *= invN;
    sC[d * 33 + ee] = v;
    ctx[(size_t)blockIdx.x * 1024 + e] = v;
  }
  __syncthreads();
  for (int e = tid; e < 32 * C; e += NT) {     // sW <- Wout[c'][h*32 + e] as [c'][33]
    const int cc = e >> 5, ee = e & 31;
    sW[cc * 33 + ee] = wout[(size_t)cc * HD + h * kLapDH + ee];
  }
  __syncthreads();
  for (int e = tid; e < 32 * C; e += NT) {
    const int d = e / C, cc = e - d * C;
    float v = 0.f;
#pragma unroll
    for (int ee = 0; ee < 32; ++ee) v = fmaf(sC[d * 33 + ee], sW[cc * 33 + ee], v);
    P[(size_t)blockIdx.x * 32 * C + e] = v;
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// forward 2: y = bias + x + sum_h qs_h P_h, one wave per 32-pixel tile, heads in a loop; also leaves qstat = (max, 1/sum) of the
// q softmax per (pixel, head) for the backward pixel sums.  LDS: W_q of all heads [HD][C+4] | P of this image [heads*32][C]
// ---------------------------------------------------------------------------------------------------------------------
template <int CB>
__global__ void __launch_bounds__(256) lap_out_kernel(const float* __restrict__ xn, const float* __restrict__ wqkv,
                                                      const float* __restrict__ P, const float* __restrict__ bias,
                                                      const float* __restrict__ resid, float* __restrict__ y,
                                                      float* __restrict__ qstat, int N, int heads, int tiles_per_wg, float scale) {
  constexpr int C = 32 * CB, CP = C + 4;
  HIP_DYNAMIC_SHARED(float, smem)
  const int HD = heads * kLapDH;
  float* sWq = smem;                        // [HD][CP]
  float* sP = smem + (size_t)HD * CP;       // [HD][C]
  const int tid = threadIdx.x, lane = tid & 63, half = lane >> 5, l31 = lane & 31;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wgs_per_img = (N / 32 + tiles_per_wg - 1) / tiles_per_wg;
  const int b = blockIdx.x / wgs_per_img, wg = blockIdx.x % wgs_per_img;
  for (int e = tid; e < HD * (C / 4); e += 256) {
    const int row = e / (C / 4), q = e - row * (C / 4);
    *reinterpret_cast<f32x4*>(sWq + (size_t)row * CP + 4 * q) = *reinterpret_cast<const f32x4*>(wqkv + (size_t)row * C + 4 * q);
    *reinterpret_cast<f32x4*>(sP + (size_t)row * C + 4 * q) = *reinterpret_cast<const f32x4*>(P + ((size_t)b * HD + row) * C + 4 * q);
  }
  __syncthreads();
  const int t_end = (wg + 1) * tiles_per_wg < N / 32 ? (wg + 1) * tiles_per_wg : N / 32;
  for (int t = wg * tiles_per_wg + wave; t < t_end; t += 4) {
    const size_t pix = (size_t)b * N + (size_t)t * 32 + l31;
    // B operand of every head's q tile: xn^T[c][px], lane = px
    f32x4 xt[C / 8];
#pragma unroll
    for (int g8 = 0; g8 < C / 8; ++g8) xt[g8] = *reinterpret_cast<const f32x4*>(xn + pix * C + 8 * g8 + 4 * half);
    f32x16 yacc[CB];
#pragma unroll
    for (int cb = 0; cb < CB; ++cb)
      for (int r = 0; r < 16; ++r) yacc[cb][r] = 0.f;
    for (int h = 0; h < heads; ++h) {
      f32x16 qt;
      for (int r = 0; r < 16; ++r) qt[r] = 0.f;
      const float* wrow = sWq + (size_t)(h * kLapDH + l31) * CP + 4 * half;
#pragma unroll
      for (int g8 = 0; g8 < C / 8; ++g8) {
        const f32x4 w4 = *reinterpret_cast<const f32x4*>(wrow + 8 * g8);
#pragma unroll
        for (int s = 0; s < 4; ++s) qt = __builtin_amdgcn_mfma_f32_32x32x2f32(w4[s], xt[g8][s], qt, 0, 0, 0);
      }
      // qt[d][px]: lane = pixel, registers = 16 of the 32 head channels; softmax over d = in-lane + the other half
      float mx = qt[0];
#pragma unroll
      for (int r = 1; r < 16; ++r) mx = fmaxf(mx, qt[r]);
      mx = fmaxf(mx, pidm_other_half(mx));
      float sm = 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        qt[r] = lap_exp(qt[r] - mx);
        sm += qt[r];
      }
      sm += pidm_other_half(sm);
      const float inv = 1.f / sm;
      if (half == 0) *reinterpret_cast<float2*>(qstat + (pix * heads + h) * 2) = make_float2(mx, inv);
      const float sc = inv * scale;
      // y^T[c'][px] += P_h^T[c'][d] qs^T[d][px]: B = the softmax tile in its own row order, A = P_h rows read in that order
      const float* prow = sP + (size_t)(h * kLapDH) * C + l31;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float qv = qt[r] * sc;
        const float* pr = prow + (size_t)lap_row(r, half) * C;
#pragma unroll
        for (int cb = 0; cb < CB; ++cb) yacc[cb] = __builtin_amdgcn_mfma_f32_32x32x2f32(pr[32 * cb], qv, yacc[cb], 0, 0, 0);
      }
    }
    // yacc[cb][r] = y[px][32 cb + lap_row(r, half)]: four consecutive channels per register quad -> 16-byte accesses
#pragma unroll
    for (int cb = 0; cb < CB; ++cb)
#pragma unroll
      for (int q4 = 0; q4 < 4; ++q4) {
        const int c0 = 32 * cb + 8 * q4 + 4 * half;
        const f32x4 bv = *reinterpret_cast<const f32x4*>(bias + c0);
        const f32x4 rv = *reinterpret_cast<const f32x4*>(resid + pix * C + c0);
        f32x4 o;
#pragma unroll
        for (int i = 0; i < 4; ++i) o[i] = yacc[cb][4 * q4 + i] + bv[i] + rv[i];
        *reinterpret_cast<f32x4*>(y + pix * C + c0) = o;
      }
  }
}

// forward 2 in the split form (C = 32): Wq of all heads as [row][piece][c] and this image's P transposed as [piece][c'][d] live in
// LDS as bf16 pieces (staged once per workgroup); a wave splits its tile's xn rows once for all heads; the softmax tile chains
// into y^T += P_h^T qs^T as the B operand in accumulator-row order, the A operand follows with two 8-byte reads per piece.
__global__ void __launch_bounds__(512) lap_out_split_kernel(const float* __restrict__ xn, const float* __restrict__ wqkv,
                                                            const float* __restrict__ P, const float* __restrict__ bias,
                                                            const float* __restrict__ resid, float* __restrict__ y,
                                                            float* __restrict__ qstat, int N, int heads, int tiles_per_wg, float scale) {
  constexpr int C = 32, RA = 6 * C + 16;
  HIP_DYNAMIC_SHARED(float, smemf)
  char* smem = reinterpret_cast<char*>(smemf);
  const int HD = heads * kLapDH;
  const int RT = HD * 2 + 16;               // bytes per PT row (piece, c')
  char* WA = smem;                          // [HD][piece][C]
  char* PT = smem + (size_t)HD * RA;        // [piece][C][HD]
  const int tid = threadIdx.x, lane = tid & 63, half = lane >> 5, l31 = lane & 31;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wgs_per_img = (N / 32 + tiles_per_wg - 1) / tiles_per_wg;
  const int b = blockIdx.x / wgs_per_img, wg = blockIdx.x % wgs_per_img;
  typedef unsigned u32x2_t __attribute__((ext_vector_type(2)));
  {
    const int q = lane & 7, pb = (lane >> 3) & 1, la = lane >> 4;
    for (int sl = wave; sl < HD / 8; sl += 8) {          // 8 rows d x 32 channels per wave-slot
      const int d = (sl * 2 + pb) * 4 + la;
      const f32x4 v = *reinterpret_cast<const f32x4*>(wqkv + (size_t)d * C + 4 * q);
      const f32x4 w = *reinterpret_cast<const f32x4*>(P + ((size_t)b * HD + d) * C + 4 * q);
      unsigned a0, a1, a2, b0, b1, b2;
      pidm_split3_pk(v[0], v[1], a0, a1, a2);
      pidm_split3_pk(v[2], v[3], b0, b1, b2);
      char* da = WA + (size_t)d * RA + 8 * q;
      *reinterpret_cast<u32x2_t*>(da) = u32x2_t{a0, b0};
      *reinterpret_cast<u32x2_t*>(da + 2 * C) = u32x2_t{a1, b1};
      *reinterpret_cast<u32x2_t*>(da + 4 * C) = u32x2_t{a2, b2};
      unsigned t0 = __float_as_uint(w[0]), t1 = __float_as_uint(w[1]), t2 = __float_as_uint(w[2]), t3 = __float_as_uint(w[3]);
      {
        const auto s02 = __builtin_amdgcn_permlane32_swap(t0, t2, false, false);
        const auto s13 = __builtin_amdgcn_permlane32_swap(t1, t3, false, false);
        const auto s01 = __builtin_amdgcn_permlane16_swap(s02[0], s13[0], false, false);
        const auto s23 = __builtin_amdgcn_permlane16_swap(s02[1], s13[1], false, false);
        t0 = s01[0]; t1 = s01[1]; t2 = s23[0]; t3 = s23[1];
      }
      // now: channel c' = 4 q + la, rows d = (2 sl + pb) 4 + 0..3
      pidm_split3_pk(__uint_as_float(t0), __uint_as_float(t1), a0, a1, a2);
      pidm_split3_pk(__uint_as_float(t2), __uint_as_float(t3), b0, b1, b2);
      char* dt = PT + (size_t)(4 * q + la) * RT + (size_t)((sl * 2 + pb) * 4) * 2;
      *reinterpret_cast<u32x2_t*>(dt) = u32x2_t{a0, b0};
      *reinterpret_cast<u32x2_t*>(dt + (size_t)C * RT) = u32x2_t{a1, b1};
      *reinterpret_cast<u32x2_t*>(dt + (size_t)2 * C * RT) = u32x2_t{a2, b2};
    }
  }
  __syncthreads();
#define PIDM_LAP_SIX(acc_, a_, b_)                                                                                 \
  acc_ = pidm_mfma_bf16_32x32x16(a_[2], b_[0], acc_);                                                              \
  acc_ = pidm_mfma_bf16_32x32x16(a_[0], b_[2], acc_);                                                              \
  acc_ = pidm_mfma_bf16_32x32x16(a_[1], b_[1], acc_);                                                              \
  acc_ = pidm_mfma_bf16_32x32x16(a_[1], b_[0], acc_);                                                              \
  acc_ = pidm_mfma_bf16_32x32x16(a_[0], b_[1], acc_);                                                              \
  acc_ = pidm_mfma_bf16_32x32x16(a_[0], b_[0], acc_);
  const int t_end = (wg + 1) * tiles_per_wg < N / 32 ? (wg + 1) * tiles_per_wg : N / 32;
  for (int t = wg * tiles_per_wg + wave; t < t_end; t += 8) {
    const size_t pix = (size_t)b * N + (size_t)t * 32 + l31;
    // B operand of every head's q tile: xn[px][16 s + 8 half .. + 7], lane = px, split once
    u32x4 xb[C / 16][3];
#pragma unroll
    for (int s = 0; s < C / 16; ++s) {
      const f32x4 x0 = *reinterpret_cast<const f32x4*>(xn + pix * C + 16 * s + 8 * half);
      const f32x4 x1 = *reinterpret_cast<const f32x4*>(xn + pix * C + 16 * s + 8 * half + 4);
      unsigned p0[4], p1[4], p2[4];
      pidm_split3_pk(x0[0], x0[1], p0[0], p1[0], p2[0]);
      pidm_split3_pk(x0[2], x0[3], p0[1], p1[1], p2[1]);
      pidm_split3_pk(x1[0], x1[1], p0[2], p1[2], p2[2]);
      pidm_split3_pk(x1[2], x1[3], p0[3], p1[3], p2[3]);
      xb[s][0] = u32x4{p0[0], p0[1], p0[2], p0[3]};
      xb[s][1] = u32x4{p1[0], p1[1], p1[2], p1[3]};
      xb[s][2] = u32x4{p2[0], p2[1], p2[2], p2[3]};
    }
    f32x16 yacc;
    for (int r = 0; r < 16; ++r) yacc[r] = 0.f;
    for (int h = 0; h < heads; ++h) {
      f32x16 qt;
      for (int r = 0; r < 16; ++r) qt[r] = 0.f;
      const char* wrow = WA + (size_t)(h * kLapDH + l31) * RA + 16 * half;
#pragma unroll
      for (int s = 0; s < C / 16; ++s) {
        u32x4 wa[3];
#pragma unroll
        for (int pc = 0; pc < 3; ++pc) wa[pc] = *reinterpret_cast<const u32x4*>(wrow + pc * 2 * C + 32 * s);
        PIDM_LAP_SIX(qt, wa, xb[s])
      }
      // qt[d][px]: lane = pixel, registers = 16 of the 32 head channels; softmax over d = in-lane + the other half
      float mx = qt[0];
#pragma unroll
      for (int r = 1; r < 16; ++r) mx = fmaxf(mx, qt[r]);
      mx = fmaxf(mx, pidm_other_half(mx));
      float sm = 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        qt[r] = lap_exp(qt[r] - mx);
        sm += qt[r];
      }
      sm += pidm_other_half(sm);
      const float inv = 1.f / sm;
      if (half == 0) *reinterpret_cast<float2*>(qstat + (pix * heads + h) * 2) = make_float2(mx, inv);
      const float sc = inv * scale;
      // y^T[c'][px] += P_h^T[c'][d] qs^T[d][px]: B = the softmax tile (rows 8 s .. 8 s + 7 = d 16 s + 4 half + {0..3, 8..11})
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        u32x4 qb[3], pa[3];
        {
          unsigned p0[4], p1[4], p2[4];
#pragma unroll
          for (int j = 0; j < 4; ++j) pidm_split3_pk(qt[8 * s + 2 * j] * sc, qt[8 * s + 2 * j + 1] * sc, p0[j], p1[j], p2[j]);
          qb[0] = u32x4{p0[0], p0[1], p0[2], p0[3]};
          qb[1] = u32x4{p1[0], p1[1], p1[2], p1[3]};
          qb[2] = u32x4{p2[0], p2[1], p2[2], p2[3]};
        }
        const char* prow = PT + (size_t)l31 * RT + (size_t)(h * kLapDH + 16 * s + 4 * half) * 2;
#pragma unroll
        for (int pc = 0; pc < 3; ++pc) {
          const u32x2_t lo = *reinterpret_cast<const u32x2_t*>(prow + (size_t)pc * C * RT);
          const u32x2_t hi = *reinterpret_cast<const u32x2_t*>(prow + (size_t)pc * C * RT + 16);
          pa[pc] = u32x4{lo[0], lo[1], hi[0], hi[1]};
        }
        PIDM_LAP_SIX(yacc, pa, qb)
      }
    }
#pragma unroll
    for (int q4 = 0; q4 < 4; ++q4) {
      const int c0 = 8 * q4 + 4 * half;
      const f32x4 bv = *reinterpret_cast<const f32x4*>(bias + c0);
      const f32x4 rv = *reinterpret_cast<const f32x4*>(resid + pix * C + c0);
      f32x4 o;
#pragma unroll
      for (int i = 0; i < 4; ++i) o[i] = yacc[4 * q4 + i] + bv[i] + rv[i];
      *reinterpret_cast<f32x4*>(y + pix * C + c0) = o;
    }
  }
#undef PIDM_LAP_SIX
}

// ---------------------------------------------------------------------------------------------------------------------
// backward 1: G_h[d][c'] = sum_n qs_h[n][d] dY[n][c'] per (image, pixel range, head); part layout [32][C] per (b, ns, h)
// ---------------------------------------------------------------------------------------------------------------------
template <int CB>
__global__ void __launch_bounds__(512) lap_g_kernel(const float* __restrict__ xn, const float* __restrict__ dy,
                                                    const float* __restrict__ wqkv, const float* __restrict__ qstat,
                                                    float* __restrict__ part, int N, int heads, int nper, float scale) {
  constexpr int C = 32 * CB, CP = C + 4;
  HIP_DYNAMIC_SHARED(float, smem)
  float* xs = smem;                           // [nper][CP]
  float* ys = smem + (size_t)nper * CP;       // [nper][CP]
  const int tid = threadIdx.x, lane = tid & 63, half = lane >> 5, l31 = lane & 31;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int NS = N / nper;
  const int b = blockIdx.x / NS, ns = blockIdx.x % NS;
  const size_t pix0 = (size_t)b * N + (size_t)ns * nper;
  for (int e = tid; e < nper * (C / 4); e += 512) {
    const int px = e / (C / 4), q = e - px * (C / 4);
    *reinterpret_cast<f32x4*>(xs + (size_t)px * CP + 4 * q) = *reinterpret_cast<const f32x4*>(xn + (pix0 + px) * C + 4 * q);
    *reinterpret_cast<f32x4*>(ys + (size_t)px * CP + 4 * q) = *reinterpret_cast<const f32x4*>(dy + (pix0 + px) * C + 4 * q);
  }
  __syncthreads();
  const int h = wave;
  if (h >= heads) return;
  f32x4 wq[C / 8];
  const float* wrow = wqkv + ((size_t)h * kLapDH + l31) * C + 4 * half;
#pragma unroll
  for (int g8 = 0; g8 < C / 8; ++g8) wq[g8] = *reinterpret_cast<const f32x4*>(wrow + 8 * g8);
  f32x16 G[CB];
#pragma unroll
  for (int cb = 0; cb < CB; ++cb)
    for (int r = 0; r < 16; ++r) G[cb][r] = 0.f;
  for (int t = 0; t < nper / 32; ++t) {
    f32x16 qt;
    for (int r = 0; r < 16; ++r) qt[r] = 0.f;
    const float* arow = xs + (size_t)(t * 32 + l31) * CP + 4 * half;
#pragma unroll
    for (int g8 = 0; g8 < C / 8; ++g8) {
      const f32x4 a4 = *reinterpret_cast<const f32x4*>(arow + 8 * g8);
#pragma unroll
      for (int s = 0; s < 4; ++s) qt = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[s], wq[g8][s], qt, 0, 0, 0);
    }
    // qt[px][d] (lane = d): the per-pixel softmax constants come from the forward (same address across a lane half: broadcast)
    const float* st = qstat + ((pix0 + (size_t)t * 32) * heads + h) * 2;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int px = lap_row(r, half);
      const float2 s2 = *reinterpret_cast<const float2*>(st + (size_t)px * heads * 2);
      const float qv = lap_exp(qt[r] - s2.x) * s2.y * scale;
      const float* brow = ys + (size_t)(t * 32 + px) * CP + l31;
#pragma unroll
      for (int cb = 0; cb < CB; ++cb) G[cb] = __builtin_amdgcn_mfma_f32_32x32x2f32(qv, brow[32 * cb], G[cb], 0, 0, 0);
    }
  }
  float* o = part + ((size_t)blockIdx.x * heads + h) * (size_t)(32 * C);
#pragma unroll
  for (int cb = 0; cb < CB; ++cb)
#pragma unroll
    for (int r = 0; r < 16; ++r) o[(size_t)lap_row(r, half) * C + 32 * cb + l31] = G[cb][r];
}

// backward 1 in the split form (same scheme as lap_kctx_split_kernel): xn is staged as [px][piece][c] (A operand of q = xn Wq^T),
// dY as [piece][c'][px] (B operand of G += qs^T dY); qs leaves the first product with lane = d and chains into the second.
template <int CB>
__global__ void __launch_bounds__(512) lap_g_split_kernel(const float* __restrict__ xn, const float* __restrict__ dy,
                                                          const float* __restrict__ wqkv, const float* __restrict__ qstat,
                                                          float* __restrict__ part, int N, int heads, int nper, float scale) {
  constexpr int C = 32 * CB, RA = 6 * C + 16;
  HIP_DYNAMIC_SHARED(float, smemf)
  char* smem = reinterpret_cast<char*>(smemf);
  const int RT = nper * 2 + 16;
  char* XA = smem;
  char* YT = smem + (size_t)nper * RA;
  const int tid = threadIdx.x, lane = tid & 63, half = lane >> 5, l31 = lane & 31;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int NS = N / nper;
  const int b = blockIdx.x / NS, ns = blockIdx.x % NS;
  const size_t pix0 = (size_t)b * N + (size_t)ns * nper;
  {
    constexpr int QN = C / 4, GP = 16 / QN;
    const int q = lane & (QN - 1), pb = (lane & 15) / QN, la = lane >> 4;
    typedef unsigned u32x2_t __attribute__((ext_vector_type(2)));
    for (int sl = wave; sl < nper / (4 * GP); sl += 8) {
      const int px = (sl * GP + pb) * 4 + la;
      const f32x4 v = *reinterpret_cast<const f32x4*>(xn + (pix0 + px) * C + 4 * q);
      const f32x4 w = *reinterpret_cast<const f32x4*>(dy + (pix0 + px) * C + 4 * q);
      unsigned a0, a1, a2, b0, b1, b2;
      pidm_split3_pk(v[0], v[1], a0, a1, a2);
      pidm_split3_pk(v[2], v[3], b0, b1, b2);
      char* da = XA + (size_t)px * RA + 8 * q;
      *reinterpret_cast<u32x2_t*>(da) = u32x2_t{a0, b0};
      *reinterpret_cast<u32x2_t*>(da + 2 * C) = u32x2_t{a1, b1};
      *reinterpret_cast<u32x2_t*>(da + 4 * C) = u32x2_t{a2, b2};
      unsigned t0 = __float_as_uint(w[0]), t1 = __float_as_uint(w[1]), t2 = __float_as_uint(w[2]), t3 = __float_as_uint(w[3]);
      {
        const auto s02 = __builtin_amdgcn_permlane32_swap(t0, t2, false, false);
        const auto s13 = __builtin_amdgcn_permlane32_swap(t1, t3, false, false);
        const auto s01 = __builtin_amdgcn_permlane16_swap(s02[0], s13[0], false, false);
        const auto s23 = __builtin_amdgcn_permlane16_swap(s02[1], s13[1], false, false);
        t0 = s01[0]; t1 = s01[1]; t2 = s23[0]; t3 = s23[1];
      }
      pidm_split3_pk(__uint_as_float(t0), __uint_as_float(t1), a0, a1, a2);
      pidm_split3_pk(__uint_as_float(t2), __uint_as_float(t3), b0, b1, b2);
      char* dt = YT + (size_t)(4 * q + la) * RT + (size_t)((sl * GP + pb) * 4) * 2;
      *reinterpret_cast<u32x2_t*>(dt) = u32x2_t{a0, b0};
      *reinterpret_cast<u32x2_t*>(dt + (size_t)C * RT) = u32x2_t{a1, b1};
      *reinterpret_cast<u32x2_t*>(dt + (size_t)2 * C * RT) = u32x2_t{a2, b2};
    }
  }
  __syncthreads();
  const int h = wave;
  if (h >= heads) return;
  u32x4 wq[C / 16][3];
  {
    const float* wrow = wqkv + ((size_t)h * kLapDH + l31) * C + 8 * half;
#pragma unroll
    for (int s = 0; s < C / 16; ++s) {
      const f32x4 w0 = *reinterpret_cast<const f32x4*>(wrow + 16 * s), w1 = *reinterpret_cast<const f32x4*>(wrow + 16 * s + 4);
      unsigned p0[4], p1[4], p2[4];
      pidm_split3_pk(w0[0], w0[1], p0[0], p1[0], p2[0]);
      pidm_split3_pk(w0[2], w0[3], p0[1], p1[1], p2[1]);
      pidm_split3_pk(w1[0], w1[1], p0[2], p1[2], p2[2]);
      pidm_split3_pk(w1[2], w1[3], p0[3], p1[3], p2[3]);
      wq[s][0] = u32x4{p0[0], p0[1], p0[2], p0[3]};
      wq[s][1] = u32x4{p1[0], p1[1], p1[2], p1[3]};
      wq[s][2] = u32x4{p2[0], p2[1], p2[2], p2[3]};
    }
  }
  f32x16 G[CB];
#pragma unroll
  for (int cb = 0; cb < CB; ++cb)
    for (int r = 0; r < 16; ++r) G[cb][r] = 0.f;
#define PIDM_LAP_SIX(acc_, a_, b_)                                                                                 \
  acc_ = pidm_mfma_bf16_32x32x16(a_[2], b_[0], acc_);                                                              \
  acc_ = pidm_mfma_bf16_32x32x16(a_[0], b_[2], acc_);                                                              \
  acc_ = pidm_mfma_bf16_32x32x16(a_[1], b_[1], acc_);                                                              \
  acc_ = pidm_mfma_bf16_32x32x16(a_[1], b_[0], acc_);                                                              \
  acc_ = pidm_mfma_bf16_32x32x16(a_[0], b_[1], acc_);                                                              \
  acc_ = pidm_mfma_bf16_32x32x16(a_[0], b_[0], acc_);
  for (int t = 0; t < nper / 32; ++t) {
    f32x16 qt;
    for (int r = 0; r < 16; ++r) qt[r] = 0.f;
    const char* arow = XA + (size_t)(t * 32 + l31) * RA + 16 * half;
#pragma unroll
    for (int s = 0; s < C / 16; ++s) {
      u32x4 xa[3];
#pragma unroll
      for (int pc = 0; pc < 3; ++pc) xa[pc] = *reinterpret_cast<const u32x4*>(arow + pc * 2 * C + 32 * s);
      PIDM_LAP_SIX(qt, xa, wq[s])
    }
    // qt[px][d] (lane = d): the per-pixel softmax constants come from the forward (same address across a lane half: broadcast)
    const float* st = qstat + ((pix0 + (size_t)t * 32) * heads + h) * 2;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const float2 s2 = *reinterpret_cast<const float2*>(st + (size_t)lap_row(r, half) * heads * 2);
      qt[r] = lap_exp(qt[r] - s2.x) * s2.y * scale;
    }
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      u32x4 ea[3];
      {
        unsigned p0[4], p1[4], p2[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) pidm_split3_pk(qt[8 * s + 2 * j], qt[8 * s + 2 * j + 1], p0[j], p1[j], p2[j]);
        ea[0] = u32x4{p0[0], p0[1], p0[2], p0[3]};
        ea[1] = u32x4{p1[0], p1[1], p1[2], p1[3]};
        ea[2] = u32x4{p2[0], p2[1], p2[2], p2[3]};
      }
#pragma unroll
      for (int cb = 0; cb < CB; ++cb) {
        u32x4 yb4[3];
        const char* brow = YT + (size_t)(32 * cb + l31) * RT + (size_t)(t * 32 + 16 * s + 4 * half) * 2;
#pragma unroll
        for (int pc = 0; pc < 3; ++pc) {
          typedef unsigned u32x2_t __attribute__((ext_vector_type(2)));
          const u32x2_t lo = *reinterpret_cast<const u32x2_t*>(brow + (size_t)pc * C * RT);
          const u32x2_t hi = *reinterpret_cast<const u32x2_t*>(brow + (size_t)pc * C * RT + 16);
          yb4[pc] = u32x4{lo[0], lo[1], hi[0], hi[1]};
        }
        PIDM_LAP_SIX(G[cb], ea, yb4)
      }
    }
  }
#undef PIDM_LAP_SIX
  float* o = part + ((size_t)blockIdx.x * heads + h) * (size_t)(32 * C);
#pragma unroll
  for (int cb = 0; cb < CB; ++cb)
#pragma unroll
    for (int r = 0; r < 16; ++r) o[(size_t)lap_row(r, half) * C + 32 * cb + l31] = G[cb][r];
}

// backward 1b, per (image, head): G = sum of the ranges; dctx = G Wout_h; dWout share of this image; dM = dctx Wv / N;
// dWv share of this image; rowdot[d] = sum_c dM[d][c] M[d][c]
__global__ void __launch_bounds__(kLapSmallNT) lap_mid_kernel(const float* __restrict__ gpart, const float* __restrict__ wqkv,
                                                      const float* __restrict__ wout, const float* __restrict__ ctx,
                                                      const float* __restrict__ Mmat, float* __restrict__ dMmat,
                                                      float* __restrict__ rowdot, float* __restrict__ dwout_part,
                                                      float* __restrict__ dwv_part, int N, int heads, int C, int NS) {
  HIP_DYNAMIC_SHARED(float, smem)
  float* sG = smem;                       // [32][C+1]
  float* sW = sG + 32 * (C + 1);          // [C][33] Wout[c'][h*32+e], later Wv_h [32][C+1]
  float* sD = sW + (C > 33 ? C : 33) * 33 + 32 * (C + 1);   // [32][33] dctx
  float* sM = sD + 32 * 33;               // [32][C+1] M, later dM
  constexpr int NT = kLapSmallNT;
  const int b = blockIdx.x / heads, h = blockIdx.x % heads, tid = threadIdx.x;
  const int HD = heads * kLapDH;
  const float* g0 = gpart + ((size_t)b * NS * heads + h) * (size_t)(32 * C);
  for (int e = tid; e < 32 * C; e += NT) {
    const int d = e / C, c = e - d * C;
    // the NS range partials in order, eight loads in flight (the plain loop compiled to load - s_waitcnt vmcnt(0) - add - branch:
    // one memory round trip per range, 16 in a row at the 64 x 64 level; 20.2 -> 19.4 us per launch.  The same in lap_kctx_final_kernel,
    // whose 1024 threads already keep 1024 loads in flight: 14.9 -> 20.1 us, not kept)
    float v = 0.f;
    for (int k0 = 0; k0 < NS; k0 += 8) {
      float t[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) t[u] = g0[(size_t)(k0 + u < NS ? k0 + u : 0) * heads * 32 * C + e];
#pragma unroll
      for (int u = 0; u < 8; ++u) v += (k0 + u < NS) ? t[u] : 0.f;
    }
    sG[d * (C + 1) + c] = v;
    sM[d * (C + 1) + c] = Mmat[(size_t)blockIdx.x * 32 * C + e];
    const int cc = e >> 5, ee = e & 31;
    sW[cc * 33 + ee] = wout[(size_t)cc * HD + h * kLapDH + ee];
  }
  __syncthreads();
  for (int e = tid; e < 1024; e += NT) {           // dctx[d][e] = sum_c' G[d][c'] Wout[c'][h*32+e]
    const int d = e >> 5, ee = e & 31;
    float v = 0.f;
    for (int c = 0; c < C; ++c) v = fmaf(sG[d * (C + 1) + c], sW[c * 33 + ee], v);
    sD[d * 33 + ee] = v;
  }
  // dWout[c'][h*32+e] share of this image = sum_d G[d][c'] ctx[d][e]
  for (int e = tid; e < 32 * C; e += NT) {
    const int cc = e >> 5, ee = e & 31;
    float v = 0.f;
#pragma unroll 8
    for (int d = 0; d < 32; ++d) v = fmaf(sG[d * (C + 1) + cc], ctx[(size_t)blockIdx.x * 1024 + d * 32 + ee], v);
    dwout_part[((size_t)b * C + cc) * HD + h * kLapDH + ee] = v;
  }
  __syncthreads();
  for (int e = tid; e < 32 * C; e += NT) {         // sW <- Wv_h[e][c] as [32][C+1]
    const int ee = e / C, c = e - ee * C;
    sW[ee * (C + 1) + c] = wqkv[((size_t)2 * HD + h * kLapDH + ee) * C + c];
  }
  __syncthreads();
  const float invN = 1.f / (float)N;
  // dWv[h*32+e][c] share of this image = sum_d dctx[d][e] M[d][c] / N
  for (int e = tid; e < 32 * C; e += NT) {
    const int ee = e / C, c = e - ee * C;
    float v = 0.f;
#pragma unroll 8
    for (int d = 0; d < 32; ++d) v = fmaf(sD[d * 33 + ee], sM[d * (C + 1) + c], v);
    dwv_part[((size_t)b * HD + h * kLapDH + ee) * C + c] = v * invN;
  }
  __syncthreads();
  // dM[d][c] = sum_e dctx[d][e] Wv[e][c] / N; rowdot[d] = sum_c dM[d][c] M[d][c]
  for (int e = tid; e < 32 * C; e += NT) {
    const int d = e / C, c = e - d * C;
    float v = 0.f;
#pragma unroll 8
    for (int ee = 0; ee < 32; ++ee) v = fmaf(sD[d * 33 + ee], sW[ee * (C + 1) + c], v);
    v *= invN;
    dMmat[(size_t)blockIdx.x * 32 * C + e] = v;
    sG[d * (C + 1) + c] = v * sM[d * (C + 1) + c];
  }
  __syncthreads();
  if (tid < 32) {
    float v = 0.f;
    for (int c = 0; c < C; ++c) v += sG[tid * (C + 1) + c];
    rowdot[(size_t)blockIdx.x * 32 + tid] = v;
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// backward 2: per (image, pixel range): d_xn of the range (all heads, summed across the 8 waves through LDS) and this range's
// share of dWq, dWk (one head per wave, accumulated over the range's tiles).  dw part layout: [b*NS+ns][2*HD][C] (q rows, k rows)
// ---------------------------------------------------------------------------------------------------------------------
// SP (C = 32 only, needs split_dw): the four per-pixel PROJECTIONS of a head - qt = Wq xn^T, dqs = P dY^T, kt = Wk xn^T,
// dks = dM xn^T: 64 of the 112 fp32 MFMAs per head and tile - run on the bf16 pipe with 3-piece operands as well: the head's four
// 32 x 32 operand matrices are split ONCE per workgroup into registers (96 VGPRs, A operands with lane = d), the xn / dY tile is
// split once per tile by the whole workgroup into a pixel-major LDS image ([px][k-step][half][piece] - the B operands, lane = px).
// The fp32 slabs of xn / dY and the padding of the Wq | Wk | dM rows in LDS (only the register fragments read them row-wise now)
// make room for it.  The d_xn products keep the fp32 MFMA (their B operands chain from the accumulators).
#ifndef PIDM_LAP_TRACE_BUILD
#define PIDM_LAP_TRACE_BUILD 0
#endif
// PIDM_LAP_TRACE=1 in a build with -DPIDM_LAP_TRACE_BUILD=1: cycle stamps of lap_bwd, workgroup 0, waves 0 and 4 (the two waves of one SIMD): [wave slot][tile round < 8][16]
// 0 round start | 1 first half: projections consumed (softmax done) | 2 its d_xn products issued | 3 its weight-gradient share done |
// 4-6 the same for the second half | 7 d_xn share written | 8 past barrier 1 | 9 head sum stored | 10 past barrier 2
__device__ unsigned long long g_lap_trace[2 * 8 * 16];
template <int CB, bool SP>
__global__ void __launch_bounds__(512) lap_bwd_kernel(const float* __restrict__ xn, const float* __restrict__ dy,
                                                      const float* __restrict__ wqkv, const float* __restrict__ P,
                                                      const float* __restrict__ kst, const float* __restrict__ dMmat,
                                                      const float* __restrict__ rowdot, float* __restrict__ dxn,
                                                      float* __restrict__ dw_part, int N, int heads, int nper, int nsub,
                                                      float scale, int split_dw, int G) {
  constexpr int C = 32 * CB, CP = C + 4;
  static_assert(!SP || CB == 1, "the split projections keep the head's operand matrices in registers: C = 32 only");
  constexpr int WCP = SP ? C : CP;                    // row stride of Wq | Wk | dM in LDS
  constexpr int PR = 208;                             // SP: bytes per pixel row of the xn / dY piece images (13 x 16: odd)
  HIP_DYNAMIC_SHARED(float, smem)
  float* xs = smem;                                   // [nper][CP]   (not SP)
  float* ys = xs + (SP ? 0 : (size_t)nper * CP);      // [nper][CP]   (not SP)
  constexpr int TSZ = CB * 32 * kLapTileLd;           // floats of a wave's tile region
  float* tiles = ys + (SP ? 0 : (size_t)nper * CP);   // [8][TSZ]: transposition tile / d_xn share of each wave
  float* cst = tiles + (size_t)8 * TSZ;               // [8][96]: k max, k 1/Z, rowdot of each wave's head
  // C = 32: the operand matrices that are read row-wise (lane = c) 48 times per tile and head - Wq, Wk (all heads) and this image's
  // dM - live in LDS, rows padded to CP: fetched through L1 with the register file full, every one of those MFMAs waited for its
  // own load (measured: 52 % of the matrix-core rate for the whole kernel)
  constexpr bool WLDS = (CB == 1);
  float* wl = cst + 8 * 96;                           // [3][heads*32][WCP] when WLDS
  // split_dw: the weight-gradient pixel sums dWq += dq xn, dWk += dk xn run on the bf16 pipe (3-piece operands): xn of the range also
  // as pieces [piece][c][px] (B operand, pixels contiguous), dq / dk through the wave's tile as before (A operand: 8 pixels of a row)
  const int RT = nper * 2 + (WLDS ? 0 : 16);          // bytes per XT row (C = 32: no room for the conflict-avoiding pad)
  const int HD = heads * kLapDH;
  char* XT = reinterpret_cast<char*>(wl + (WLDS ? 3 * HD * WCP : 0));
  char* XP = XT + (size_t)3 * C * RT;                 // SP: [nper][PR] xn pieces, then the same for dY
  char* YP = XP + (size_t)nper * PR;
  const int tid = threadIdx.x, lane = tid & 63, half = lane >> 5, l31 = lane & 31;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int NW = N / (nper * nsub);                  // workgroups per image: each walks nsub consecutive pixel ranges
  const int b = blockIdx.x / NW, nw = blockIdx.x % NW;
  if (WLDS) {
    for (int e = tid; e < 3 * HD * (C / 4); e += 512) {
      const int m = e / (HD * (C / 4)), rem = e - m * (HD * (C / 4));
      const int row = rem / (C / 4), q = rem - row * (C / 4);
      const float* src = (m < 2) ? wqkv + ((size_t)m * HD + row) * C : dMmat + ((size_t)b * HD + row) * C;
      *reinterpret_cast<f32x4*>(wl + ((size_t)m * HD + row) * WCP + 4 * q) = *reinterpret_cast<const f32x4*>(src + 4 * q);
    }
  }
  // wave -> (tile group, head): group tg takes tiles tg, tg + G, .. of every slab (lap_groups above)
  const int tg = wave / heads, h = wave - tg * heads;
  const bool act = tg < G;
  float* tw = tiles + (size_t)wave * TSZ;
  float* cw = cst + wave * 96;
  if (act && half == 0) {
    cw[l31] = kst[(((size_t)b * heads + h) * 32 + l31) * 2];
    cw[32 + l31] = kst[(((size_t)b * heads + h) * 32 + l31) * 2 + 1];
    cw[64 + l31] = rowdot[((size_t)b * heads + h) * 32 + l31];
  }
  // A operands with lane = d (rows of Wq_h, Wk_h, P_h, dM_h; 16 bytes per lane and 8-channel group) are re-fetched per tile
  // through L1 instead of living in registers for the whole range: with them resident the kernel needs > 256 registers
  // (two waves per SIMD in a 512-thread workgroup: 256 is the ceiling) and spills into the MFMA loops
  f32x16 dWq[CB], dWk[CB];
  const size_t bh = (size_t)b * heads + (act ? h : 0);
  const int hrow = (act ? h : 0) * kLapDH + l31;
  const float* wq_p = WLDS ? wl + (size_t)hrow * WCP + 4 * half : wqkv + (size_t)hrow * C + 4 * half;
  const float* wk_p = WLDS ? wl + (size_t)(HD + hrow) * WCP + 4 * half : wqkv + ((size_t)HD + hrow) * C + 4 * half;
  const float* dm_p = WLDS ? wl + (size_t)(2 * HD + hrow) * WCP + 4 * half : dMmat + (bh * 32 + l31) * C + 4 * half;
  const float* pp_p = P + (bh * 32 + l31) * C + 4 * half;
  // SP: the head's operand rows as bf16 pieces in registers: f..[k-step][piece], lane = (d = l31, k = 16 ks + 8 half + 0..7)
  u32x4 fwq[SP ? C / 16 : 1][3], fwk[SP ? C / 16 : 1][3], fpp[SP ? C / 16 : 1][3], fdm[SP ? C / 16 : 1][3];
  if constexpr (SP) {
    __syncthreads();                                  // Wq | Wk | dM are in LDS
#pragma unroll
    for (int ks = 0; ks < C / 16; ++ks) {
      const float* srcs[4] = {wl + (size_t)hrow * WCP + 16 * ks + 8 * half, wl + (size_t)(HD + hrow) * WCP + 16 * ks + 8 * half,
                              P + (bh * 32 + l31) * C + 16 * ks + 8 * half, wl + (size_t)(2 * HD + hrow) * WCP + 16 * ks + 8 * half};
#pragma unroll
      for (int m = 0; m < 4; ++m) {
        const f32x4 v0 = *reinterpret_cast<const f32x4*>(srcs[m]), v1 = *reinterpret_cast<const f32x4*>(srcs[m] + 4);
        unsigned q0[4], q1[4], q2[4];
        pidm_split3_pk(v0[0], v0[1], q0[0], q1[0], q2[0]);
        pidm_split3_pk(v0[2], v0[3], q0[1], q1[1], q2[1]);
        pidm_split3_pk(v1[0], v1[1], q0[2], q1[2], q2[2]);
        pidm_split3_pk(v1[2], v1[3], q0[3], q1[3], q2[3]);
        u32x4(&f)[3] = (m == 0) ? fwq[ks] : (m == 1) ? fwk[ks] : (m == 2) ? fpp[ks] : fdm[ks];
        f[0] = u32x4{q0[0], q0[1], q0[2], q0[3]};
        f[1] = u32x4{q1[0], q1[1], q1[2], q1[3]};
        f[2] = u32x4{q2[0], q2[1], q2[2], q2[3]};
      }
    }
  }
  if (act) {
#pragma unroll
    for (int cb = 0; cb < CB; ++cb)
      for (int r = 0; r < 16; ++r) { dWq[cb][r] = 0.f; dWk[cb][r] = 0.f; }
  }
  // Wq_h[d][c = l31 + 32 cb] etc. for the d_xn products (A operand, lane = c); row stride ldT
  const int ldT = WLDS ? WCP : C;
  const int hT = act ? h : 0;
  const float* wqT = WLDS ? wl + (size_t)(hT * kLapDH) * WCP + l31 : wqkv + (size_t)hT * kLapDH * C + l31;
  const float* wkT = WLDS ? wl + (size_t)(HD + hT * kLapDH) * WCP + l31 : wqkv + ((size_t)HD + hT * kLapDH) * C + l31;
  const float* dmT = WLDS ? wl + (size_t)(2 * HD + hT * kLapDH) * WCP + l31 : dMmat + ((size_t)b * heads + hT) * 32 * C + l31;

  const float rscale = 1.f / scale;
  // dW[d][c] += sum_px g[d][px] xn[px][c] for the tile: g (lane = px, 16 rows d per lane) is turned through the wave's LDS tile
#define PIDM_LAP_DW(g_, dW_)                                                                                       \
  if (split_dw) {                                                                                                  \
    _Pragma("unroll") for (int r = 0; r < 16; ++r) tw[lap_row(r, half) * kLapTileLd + l31] = g_[r];                \
    PIDM_WAVE_LDS_SYNC();                                                                                          \
    _Pragma("unroll") for (int s = 0; s < 2; ++s) {                                                                \
      float g0[4], g1[4];                                                                                          \
      _Pragma("unroll") for (int e = 0; e < 4; ++e) {                                                              \
        g0[e] = tw[l31 * kLapTileLd + 16 * s + 8 * half + e];                                                      \
        g1[e] = tw[l31 * kLapTileLd + 16 * s + 8 * half + 4 + e];                                                  \
      }                                                                                                            \
      unsigned p0[4], p1[4], p2[4];                                                                                \
      pidm_split3_pk(g0[0], g0[1], p0[0], p1[0], p2[0]);                                                           \
      pidm_split3_pk(g0[2], g0[3], p0[1], p1[1], p2[1]);                                                           \
      pidm_split3_pk(g1[0], g1[1], p0[2], p1[2], p2[2]);                                                           \
      pidm_split3_pk(g1[2], g1[3], p0[3], p1[3], p2[3]);                                                           \
      u32x4 ga[3] = {u32x4{p0[0], p0[1], p0[2], p0[3]}, u32x4{p1[0], p1[1], p1[2], p1[3]}, u32x4{p2[0], p2[1], p2[2], p2[3]}}; \
      _Pragma("unroll") for (int cb = 0; cb < CB; ++cb) {                                                          \
        u32x4 xb4[3];                                                                                              \
        const char* brow = XT + (size_t)(32 * cb + l31) * RT + (size_t)(t * 32 + 16 * s + 8 * half) * 2;           \
        _Pragma("unroll") for (int pc = 0; pc < 3; ++pc) xb4[pc] = *reinterpret_cast<const u32x4*>(brow + (size_t)pc * C * RT); \
        dW_[cb] = pidm_mfma_bf16_32x32x16(ga[2], xb4[0], dW_[cb]);                                                 \
        dW_[cb] = pidm_mfma_bf16_32x32x16(ga[0], xb4[2], dW_[cb]);                                                 \
        dW_[cb] = pidm_mfma_bf16_32x32x16(ga[1], xb4[1], dW_[cb]);                                                 \
        dW_[cb] = pidm_mfma_bf16_32x32x16(ga[1], xb4[0], dW_[cb]);                                                 \
        dW_[cb] = pidm_mfma_bf16_32x32x16(ga[0], xb4[1], dW_[cb]);                                                 \
        dW_[cb] = pidm_mfma_bf16_32x32x16(ga[0], xb4[0], dW_[cb]);                                                 \
      }                                                                                                            \
    }                                                                                                              \
    PIDM_WAVE_LDS_SYNC();                                                                                          \
  } else {                                                                                                         \
    _Pragma("unroll") for (int r = 0; r < 16; ++r) tw[lap_row(r, half) * kLapTileLd + l31] = g_[r];                \
    PIDM_WAVE_LDS_SYNC();                                                                                          \
    _Pragma("unroll") for (int s = 0; s < 16; ++s) {                                                               \
      const float a = tw[l31 * kLapTileLd + 2 * s + half];                                                         \
      const float* brow = xs + (size_t)(t * 32 + 2 * s + half) * CP + l31;                                         \
      _Pragma("unroll") for (int cb = 0; cb < CB; ++cb) dW_[cb] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, brow[32 * cb], dW_[cb], 0, 0, 0); \
    }                                                                                                              \
    PIDM_WAVE_LDS_SYNC();                                                                                          \
  }
#define PIDM_LAP_SIX(acc_, fa_, fb_)                                                                               \
  acc_ = pidm_mfma_bf16_32x32x16(fa_[2], fb_[0], acc_);                                                            \
  acc_ = pidm_mfma_bf16_32x32x16(fa_[0], fb_[2], acc_);                                                            \
  acc_ = pidm_mfma_bf16_32x32x16(fa_[1], fb_[1], acc_);                                                            \
  acc_ = pidm_mfma_bf16_32x32x16(fa_[1], fb_[0], acc_);                                                            \
  acc_ = pidm_mfma_bf16_32x32x16(fa_[0], fb_[1], acc_);                                                            \
  acc_ = pidm_mfma_bf16_32x32x16(fa_[0], fb_[0], acc_);
  // the two waves of a SIMD (w and w + 4) walk the two halves of a tile's work in opposite order: the per-tile barriers keep all
  // waves in step, and in the same order their matrix-pipe phases and their vector phases coincide instead of overlapping
  const bool order_flip = (split_dw & 2) != 0;
  // the second-dispatched half of the workgroup (waves 4-7: the younger wave of every SIMD) loses the issue arbitration against its
  // older partner on every phase - stamps: 16 700 against 13 400 busy cycles per tile round, the older wave then waits 3300 cycles at
  // the round's barrier.  Static priority for that half (split_dw bit 3; PIDM_LAP_PRIO=1) only swaps the roles: the round stays at
  // 18 500 cycles - the SIMD's vector pipe (fp32 MFMAs + softmax + splitting of BOTH waves) is what is full (C = 32 -1.8 %, C = 64
  // +6 %: profiles/r06_lap_bwd_stamps.txt).  Off by default.
  if ((split_dw & 8) && wave >= 4) __builtin_amdgcn_s_setprio(1);
  // (the stamps exist in measurement builds only, -DPIDM_LAP_TRACE_BUILD=1: the C = 32 instantiation sits at 256 registers without a
  // spill, and even scalar stamp code pushed 40 registers into scratch)
#if PIDM_LAP_TRACE_BUILD
  const bool tr_on = (split_dw & 4) && blockIdx.x == 0 && (wave & 3) == 0;
  int tr_round = 0;
#define PIDM_LAP_STAMP(i_)                                                                                         \
  if (tr_on && tr_round < 8) {                                                                                     \
    const unsigned long long c__ = clock64();                                                                      \
    if (lane == 0) g_lap_trace[(wave >> 2) * 128 + 16 * tr_round + (i_)] = c__;                                    \
  }
#define PIDM_LAP_TRACE_ONLY(x_) x_
#else
#define PIDM_LAP_STAMP(i_)
#define PIDM_LAP_TRACE_ONLY(x_)
#endif
  split_dw &= 1;
  const bool k_first = (wave & 4) != 0 && order_flip;
  for (int sub = 0; sub < nsub; ++sub) {
  const size_t pix0 = (size_t)b * N + ((size_t)nw * nsub + sub) * nper;
  // (re)stage the range's xn and dY slabs; the loop below ends on a barrier, so nobody still reads the previous range
  for (int e = tid; e < nper * (C / 4); e += 512) {
    const int px = e / (C / 4), q = e - px * (C / 4);
    const f32x4 vx = *reinterpret_cast<const f32x4*>(xn + (pix0 + px) * C + 4 * q);
    const f32x4 vy = *reinterpret_cast<const f32x4*>(dy + (pix0 + px) * C + 4 * q);
    if constexpr (SP) {
      // 4 channels of one pixel -> 3 pieces x 2 dwords at [px][k-step = q / 4][half = (q / 2) & 1][piece][low / high 4 channels]
      typedef unsigned u32x2_t __attribute__((ext_vector_type(2)));
      const size_t o = (size_t)px * PR + ((q >> 2) * 2 + ((q >> 1) & 1)) * 48 + (q & 1) * 8;
      unsigned a0, a1, a2, b0, b1, b2;
      pidm_split3_pk(vx[0], vx[1], a0, a1, a2);
      pidm_split3_pk(vx[2], vx[3], b0, b1, b2);
      *reinterpret_cast<u32x2_t*>(XP + o) = u32x2_t{a0, b0};
      *reinterpret_cast<u32x2_t*>(XP + o + 16) = u32x2_t{a1, b1};
      *reinterpret_cast<u32x2_t*>(XP + o + 32) = u32x2_t{a2, b2};
      pidm_split3_pk(vy[0], vy[1], a0, a1, a2);
      pidm_split3_pk(vy[2], vy[3], b0, b1, b2);
      *reinterpret_cast<u32x2_t*>(YP + o) = u32x2_t{a0, b0};
      *reinterpret_cast<u32x2_t*>(YP + o + 16) = u32x2_t{a1, b1};
      *reinterpret_cast<u32x2_t*>(YP + o + 32) = u32x2_t{a2, b2};
    } else {
      *reinterpret_cast<f32x4*>(xs + (size_t)px * CP + 4 * q) = vx;
      *reinterpret_cast<f32x4*>(ys + (size_t)px * CP + 4 * q) = vy;
    }
  }
  if (split_dw) {
    constexpr int QN = C / 4, GP = 16 / QN;
    const int q = lane & (QN - 1), pb = (lane & 15) / QN, la = lane >> 4;
    typedef unsigned u32x2_t __attribute__((ext_vector_type(2)));
    for (int sl = wave; sl < nper / (4 * GP); sl += 8) {
      const int px = (sl * GP + pb) * 4 + la;
      const f32x4 w = *reinterpret_cast<const f32x4*>(xn + (pix0 + px) * C + 4 * q);
      unsigned t0 = __float_as_uint(w[0]), t1 = __float_as_uint(w[1]), t2 = __float_as_uint(w[2]), t3 = __float_as_uint(w[3]);
      {
        const auto s02 = __builtin_amdgcn_permlane32_swap(t0, t2, false, false);
        const auto s13 = __builtin_amdgcn_permlane32_swap(t1, t3, false, false);
        const auto s01 = __builtin_amdgcn_permlane16_swap(s02[0], s13[0], false, false);
        const auto s23 = __builtin_amdgcn_permlane16_swap(s02[1], s13[1], false, false);
        t0 = s01[0]; t1 = s01[1]; t2 = s23[0]; t3 = s23[1];
      }
      unsigned a0, a1, a2, b0, b1, b2;
      pidm_split3_pk(__uint_as_float(t0), __uint_as_float(t1), a0, a1, a2);
      pidm_split3_pk(__uint_as_float(t2), __uint_as_float(t3), b0, b1, b2);
      char* dt = XT + (size_t)(4 * q + la) * RT + (size_t)((sl * GP + pb) * 4) * 2;
      *reinterpret_cast<u32x2_t*>(dt) = u32x2_t{a0, b0};
      *reinterpret_cast<u32x2_t*>(dt + (size_t)C * RT) = u32x2_t{a1, b1};
      *reinterpret_cast<u32x2_t*>(dt + (size_t)2 * C * RT) = u32x2_t{a2, b2};
    }
  }
  __syncthreads();
  for (int t0 = 0; t0 < nper / 32; t0 += G) {
    const int t = t0 + tg;                 // this wave's tile of the round (act waves only)
    f32x16 dx[CB];
    PIDM_LAP_STAMP(0)
    PIDM_LAP_TRACE_ONLY(int tr_half = 0;)  // 0 while the first of q_part / k_part runs
    if (act) {
      int z0 = 0;
      if (!WLDS) PIDM_OPAQUE_I32(z0);      // keeps the operand fetches below inside the tile loop
#pragma unroll
      for (int cb = 0; cb < CB; ++cb)
        for (int r = 0; r < 16; ++r) dx[cb][r] = 0.f;
      // B operands with lane = px: xn^T and dY^T rows of the tile, fetched from the LDS slabs where they are used (holding them
      // across the whole tile costs C/2 registers each)
      const float* xrow = xs + (size_t)(t * 32 + l31) * CP + 4 * half;
      const float* yrow = ys + (size_t)(t * 32 + l31) * CP + 4 * half;
      // SP: B operands of the projections: this lane's pixel, 8 channels per k-step half, three pieces each
      const char* xpr = XP + (size_t)(t * 32 + l31) * PR + 48 * half;
      const char* ypr = YP + (size_t)(t * 32 + l31) * PR + 48 * half;
      // ---- q: qs^T[d][px], dqs^T[d][px] = P_h dY^T, softmax Jacobian per pixel (in-lane + other half) ----
      auto q_part = [&]() __attribute__((always_inline)) {
      f32x16 qt, dq;
      for (int r = 0; r < 16; ++r) { qt[r] = 0.f; dq[r] = 0.f; }
      if constexpr (SP) {
#pragma unroll
        for (int ks = 0; ks < C / 16; ++ks) {
          u32x4 xb[3], yb[3];
#pragma unroll
          for (int pc = 0; pc < 3; ++pc) {
            xb[pc] = *reinterpret_cast<const u32x4*>(xpr + 96 * ks + 16 * pc);
            yb[pc] = *reinterpret_cast<const u32x4*>(ypr + 96 * ks + 16 * pc);
          }
          PIDM_LAP_SIX(qt, fwq[ks], xb)
          PIDM_LAP_SIX(dq, fpp[ks], yb)
        }
      } else {
#pragma unroll 2
      for (int g8 = 0; g8 < C / 8; ++g8) {
        const f32x4 w4 = *reinterpret_cast<const f32x4*>(wq_p + z0 + 8 * g8);
        const f32x4 p4 = *reinterpret_cast<const f32x4*>(pp_p + z0 + 8 * g8);
        const f32x4 x4 = *reinterpret_cast<const f32x4*>(xrow + 8 * g8);
        const f32x4 y4 = *reinterpret_cast<const f32x4*>(yrow + 8 * g8);
#pragma unroll
        for (int s = 0; s < 4; ++s) {
          qt = __builtin_amdgcn_mfma_f32_32x32x2f32(w4[s], x4[s], qt, 0, 0, 0);
          dq = __builtin_amdgcn_mfma_f32_32x32x2f32(p4[s], y4[s], dq, 0, 0, 0);
        }
      }
      }
      float mx = qt[0];
#pragma unroll
      for (int r = 1; r < 16; ++r) mx = fmaxf(mx, qt[r]);
      mx = fmaxf(mx, pidm_other_half(mx));
      float sm = 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        qt[r] = lap_exp(qt[r] - mx);
        sm += qt[r];
      }
      sm += pidm_other_half(sm);
      const float sc = scale / sm;
      float jd = 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        qt[r] *= sc;                     // qs
        jd += qt[r] * dq[r];
      }
      jd = (jd + pidm_other_half(jd)) * rscale;
#pragma unroll
      for (int r = 0; r < 16; ++r) dq[r] = qt[r] * (dq[r] - jd);
      PIDM_LAP_STAMP(1 + 3 * tr_half)
      // d_xn^T[c][px] += Wq_h^T[c][d] dq^T[d][px]  (operand rows fetched as one batch, then the MFMAs)
      if constexpr (WLDS) {
        float wa[16][CB];
#pragma unroll
        for (int r = 0; r < 16; ++r)
#pragma unroll
          for (int cb = 0; cb < CB; ++cb) wa[r][cb] = (wqT + (size_t)lap_row(r, half) * ldT)[32 * cb];
#pragma unroll
        for (int r = 0; r < 16; ++r)
#pragma unroll
          for (int cb = 0; cb < CB; ++cb) dx[cb] = __builtin_amdgcn_mfma_f32_32x32x2f32(wa[r][cb], dq[r], dx[cb], 0, 0, 0);
      } else {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float* wr = wqT + (size_t)lap_row(r, half) * ldT;
#pragma unroll
          for (int cb = 0; cb < CB; ++cb) dx[cb] = __builtin_amdgcn_mfma_f32_32x32x2f32(wr[32 * cb], dq[r], dx[cb], 0, 0, 0);
        }
      }
      PIDM_LAP_STAMP(2 + 3 * tr_half)
      // dWq_h[d][c] += sum_px dq[px][d] xn[px][c]: turn dq^T through the wave's LDS tile (write [d][px], read lane = d)
      PIDM_LAP_DW(dq, dWq)
      PIDM_LAP_STAMP(3 + 3 * tr_half)
      PIDM_LAP_TRACE_ONLY(tr_half = 1;)
      };
      // ---- k: ks^T[d][px] from the saved column statistics, dks^T = dM_h xn^T, dk = ks (dks - rowdot) ----
      auto k_part = [&]() __attribute__((always_inline)) {
      f32x16 kt, dk;
      for (int r = 0; r < 16; ++r) { kt[r] = 0.f; dk[r] = 0.f; }
      if constexpr (SP) {
#pragma unroll
        for (int ks = 0; ks < C / 16; ++ks) {
          u32x4 xb[3];
#pragma unroll
          for (int pc = 0; pc < 3; ++pc) xb[pc] = *reinterpret_cast<const u32x4*>(xpr + 96 * ks + 16 * pc);
          PIDM_LAP_SIX(kt, fwk[ks], xb)
          PIDM_LAP_SIX(dk, fdm[ks], xb)
        }
      } else {
#pragma unroll 2
      for (int g8 = 0; g8 < C / 8; ++g8) {
        const f32x4 w4 = *reinterpret_cast<const f32x4*>(wk_p + z0 + 8 * g8);
        const f32x4 m4 = *reinterpret_cast<const f32x4*>(dm_p + z0 + 8 * g8);
        const f32x4 x4 = *reinterpret_cast<const f32x4*>(xrow + 8 * g8);
#pragma unroll
        for (int s = 0; s < 4; ++s) {
          kt = __builtin_amdgcn_mfma_f32_32x32x2f32(w4[s], x4[s], kt, 0, 0, 0);
          dk = __builtin_amdgcn_mfma_f32_32x32x2f32(m4[s], x4[s], dk, 0, 0, 0);
        }
      }
      }
      // per-head column constants in accumulator-row order (row d = lap_row(r, half)): k max, k 1/Z, rowdot from the wave's LDS block
#pragma unroll
      for (int q4 = 0; q4 < 4; ++q4) {
        const f32x4 mx4 = *reinterpret_cast<const f32x4*>(cw + 8 * q4 + 4 * half);
        const f32x4 iz4 = *reinterpret_cast<const f32x4*>(cw + 32 + 8 * q4 + 4 * half);
        const f32x4 rd4 = *reinterpret_cast<const f32x4*>(cw + 64 + 8 * q4 + 4 * half);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int r = 4 * q4 + i;
          kt[r] = lap_exp(kt[r] - mx4[i]) * iz4[i];     // ks
          dk[r] = kt[r] * (dk[r] - rd4[i]);
        }
      }
      PIDM_LAP_STAMP(1 + 3 * tr_half)
      // d_xn^T[c][px] += Wk_h^T[c][d] dk^T[d][px] + dM_h^T[c][d] ks^T[d][px]
      if constexpr (WLDS) {
        float wa[16][CB], ma[16][CB];
#pragma unroll
        for (int r = 0; r < 16; ++r)
#pragma unroll
          for (int cb = 0; cb < CB; ++cb) {
            wa[r][cb] = (wkT + (size_t)lap_row(r, half) * ldT)[32 * cb];
            ma[r][cb] = (dmT + (size_t)lap_row(r, half) * ldT)[32 * cb];
          }
#pragma unroll
        for (int r = 0; r < 16; ++r)
#pragma unroll
          for (int cb = 0; cb < CB; ++cb) {
            dx[cb] = __builtin_amdgcn_mfma_f32_32x32x2f32(wa[r][cb], dk[r], dx[cb], 0, 0, 0);
            dx[cb] = __builtin_amdgcn_mfma_f32_32x32x2f32(ma[r][cb], kt[r], dx[cb], 0, 0, 0);
          }
      } else {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float* wr = wkT + (size_t)lap_row(r, half) * ldT;
          const float* mr = dmT + (size_t)lap_row(r, half) * ldT;
#pragma unroll
          for (int cb = 0; cb < CB; ++cb) {
            dx[cb] = __builtin_amdgcn_mfma_f32_32x32x2f32(wr[32 * cb], dk[r], dx[cb], 0, 0, 0);
            dx[cb] = __builtin_amdgcn_mfma_f32_32x32x2f32(mr[32 * cb], kt[r], dx[cb], 0, 0, 0);
          }
        }
      }
      PIDM_LAP_STAMP(2 + 3 * tr_half)
      PIDM_LAP_DW(dk, dWk)
      PIDM_LAP_STAMP(3 + 3 * tr_half)
      PIDM_LAP_TRACE_ONLY(tr_half = 1;)
      };
      if (k_first) { k_part(); q_part(); } else { q_part(); k_part(); }
      // this head's d_xn^T share -> the wave's tile(s): [cb][c][px]
#pragma unroll
      for (int cb = 0; cb < CB; ++cb)
#pragma unroll
        for (int r = 0; r < 16; ++r) tw[(cb * 32 + lap_row(r, half)) * kLapTileLd + l31] = dx[cb][r];
    }
    PIDM_LAP_STAMP(7)
    __syncthreads();
    PIDM_LAP_STAMP(8)
    // sum over the heads (waves of a group) and store d_xn rows of the round's G tiles: thread -> (pixel, 4 consecutive channels)
    for (int e = tid; e < G * 32 * (C / 4); e += 512) {
      const int px = e / (C / 4), c0 = 4 * (e - px * (C / 4));      // px: 0 .. 32 G - 1 (group = px / 32)
      const int gq = px >> 5, p31 = px & 31;
      f32x4 o = {0.f, 0.f, 0.f, 0.f};
      for (int w = gq * heads; w < (gq + 1) * heads; ++w) {
        const float* tp = tiles + (size_t)w * TSZ + (size_t)c0 * kLapTileLd + p31;
#pragma unroll
        for (int i = 0; i < 4; ++i) o[i] += tp[i * kLapTileLd];
      }
      *reinterpret_cast<f32x4*>(dxn + (pix0 + (size_t)t0 * 32 + px) * C + c0) = o;
    }
    PIDM_LAP_STAMP(9)
    __syncthreads();
    PIDM_LAP_STAMP(10)
    PIDM_LAP_TRACE_ONLY(++tr_round;)
  }
  }
#undef PIDM_LAP_STAMP
#undef PIDM_LAP_TRACE_ONLY
  // the groups' dWq / dWk shares of a head are summed in group order through the waves' tiles (free after the last barrier above)
  for (int m = 0; m < 2 && G > 1; ++m) {
    if (act && tg > 0) {
#pragma unroll
      for (int cb = 0; cb < CB; ++cb)
#pragma unroll
        for (int r = 0; r < 16; ++r) tw[(cb * 32 + lap_row(r, half)) * kLapTileLd + l31] = m ? dWk[cb][r] : dWq[cb][r];
    }
    __syncthreads();
    if (act && tg == 0) {
      for (int g = 1; g < G; ++g) {
        const float* og = tiles + (size_t)(g * heads + h) * TSZ;
#pragma unroll
        for (int cb = 0; cb < CB; ++cb)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const float v = og[(cb * 32 + lap_row(r, half)) * kLapTileLd + l31];
            if (m) dWk[cb][r] += v; else dWq[cb][r] += v;
          }
      }
    }
    __syncthreads();
  }
  if (act && tg == 0) {
    float* oq = dw_part + ((size_t)blockIdx.x * 2 * HD + h * kLapDH) * C;
    float* ok = oq + (size_t)HD * C;
#pragma unroll
    for (int cb = 0; cb < CB; ++cb)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        oq[(size_t)lap_row(r, half) * C + 32 * cb + l31] = dWq[cb][r];
        ok[(size_t)lap_row(r, half) * C + 32 * cb + l31] = dWk[cb][r];
      }
  }
#undef PIDM_LAP_DW
#undef PIDM_LAP_SIX
}

// ---------------------------------------------------------------------------------------------------------------------
// launchers
// ---------------------------------------------------------------------------------------------------------------------
bool lap_ok(int N, int heads, int C, int Cout) {
  return Cout == C && (C == 32 || C == 64) && N % 32 == 0 && N >= 32 && heads >= 1 && heads <= 8;
}

// floats of caller scratch (partials of the pixel-range kernels; the larger of forward and backward needs)
size_t lap_scratch_floats(int B, int N, int heads, int C) {
  const size_t fw = (size_t)B * (N / lap_nper_split(N, C)) * heads * (32 * C + 64);   // the split kernel's ranges are the smaller ones
  const size_t bw = (size_t)B * (N / lap_nper(N, C, 2)) * heads * 32 * C;
  return (fw > bw ? fw : bw) + 64;
}
// floats of the saved statistics block: kst | M | ctx | P  (qstat is separate: B*N*heads*2)
size_t lap_saved_floats(int B, int heads, int C) { return (size_t)B * heads * (64 + 32 * C + 1024 + 32 * C); }
// partial weight-gradient buffers of one backward: dWq|dWk per pixel range, dWv per image, dWout per image
// ranges one lap_bwd workgroup walks (its dWq / dWk share is written once): keeps the partial buffers small while the
// launch still has >= 8 workgroups per image
static int lap_nsub(int N, int C) {
  const int ns = N / lap_nper(N, C, 3);
  int k = 1;
  while (k < 16 && ns % (2 * k) == 0 && ns / (2 * k) >= 8) k *= 2;
  return k;
}
size_t lap_dw_ranges(int N, int C) { return (size_t)(N / lap_nper(N, C, 3) / lap_nsub(N, C)); }

template <int CB>
static int lap_forward_t(const float* xn, const float* wqkv, const float* wout, const float* bias, const float* resid, float* y,
                         float* saved, float* qstat, int B, int N, int heads, float* scratch, hipStream_t st) {
  constexpr int C = 32 * CB;
  const char* spe = knob("PIDM_LAP_SPLIT");                 // 0: the fp32-MFMA pixel-sum kernel
  const bool split1 = !(spe && !atoi(spe));
  const int nper = split1 ? lap_nper_split(N, C) : lap_nper(N, C, 1), NS = N / nper;
  float* kst = saved;
  float* Mmat = kst + (size_t)B * heads * 64;
  float* ctx = Mmat + (size_t)B * heads * 32 * C;
  float* P = ctx + (size_t)B * heads * 1024;
  const size_t lds1 = ((size_t)nper * (C + 4) + 8 * 32) * sizeof(float);
  static bool attr = false;
  if (!attr) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&lap_kctx_kernel<CB>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 256);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&lap_out_kernel<CB>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 256);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&lap_kctx_final_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
    attr = true;
  }
  if (split1) {
    const size_t ldss = (size_t)nper * (6 * C + 16) + (size_t)3 * C * (nper * 2 + 16) + 8 * 32 * sizeof(float);
    static bool attr_s = false;
    if (!attr_s) {
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&lap_kctx_split_kernel<CB>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 256);
      attr_s = true;
    }
    PIDM_PROF_NAME("lap_kctx_split_kernel");
    hipLaunchKernelGGL(HIP_KERNEL_NAME(lap_kctx_split_kernel<CB>), dim3(B * NS), dim3(512), ldss, st, xn, wqkv, scratch, N, heads, nper);
  } else {
    hipLaunchKernelGGL(HIP_KERNEL_NAME(lap_kctx_kernel<CB>), dim3(B * NS), dim3(512), lds1, st, xn, wqkv, scratch, N, heads, nper);
  }
  PIDM_CHECK_LAUNCH("lap_kctx_kernel");
  const size_t lds2 = ((size_t)32 * C + 32 * (C + 1) + C + 32 * 33 + (size_t)NS * 96 + 64) * sizeof(float);
  hipLaunchKernelGGL(lap_kctx_final_kernel, dim3(B * heads), dim3(kLapSmallNT), lds2, st, scratch, wqkv, wout, kst, Mmat, ctx, P, N, heads, C, NS);
  PIDM_CHECK_LAUNCH("lap_kctx_final_kernel");
  const int HD = heads * kLapDH;
  const size_t lds3 = ((size_t)HD * (C + 4) + (size_t)HD * C) * sizeof(float);
  // tiles per workgroup: enough workgroups to fill the chip twice, at least one tile per wave
  int tpw = 4;
  while ((long)B * ((N / 32 + tpw - 1) / tpw) > 2048 && tpw < 64) tpw *= 2;
  const int wgs = B * ((N / 32 + tpw - 1) / tpw);
  if (split1 && CB == 1 && HD % 8 == 0) {
    // 8 waves per workgroup (the pieces of Wq and P^T take 104 KB: one workgroup per CU)
    int tp8 = 8;
    while ((long)B * ((N / 32 + tp8 - 1) / tp8) > 1024 && tp8 < 64) tp8 *= 2;
    const int wg8 = B * ((N / 32 + tp8 - 1) / tp8);
    const size_t ldso = (size_t)HD * (6 * C + 16) + (size_t)3 * C * (HD * 2 + 16);
    static bool attr_o = false;
    if (!attr_o) {
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&lap_out_split_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 256);
      attr_o = true;
    }
    PIDM_PROF_NAME("lap_out_split_kernel");
    hipLaunchKernelGGL(lap_out_split_kernel, dim3(wg8), dim3(512), ldso, st, xn, wqkv, P, bias, resid, y, qstat, N, heads, tp8, 0.17677669529663687f);
  } else {
    hipLaunchKernelGGL(HIP_KERNEL_NAME(lap_out_kernel<CB>), dim3(wgs), dim3(256), lds3, st, xn, wqkv, P, bias, resid, y, qstat, N, heads, tpw,
                       0.17677669529663687f);
  }
  PIDM_CHECK_LAUNCH("lap_out_kernel");
  return 0;
}

int launch_lap_forward(const float* xn, const float* wqkv, const float* wout, const float* bias, const float* resid, float* y,
                       float* saved, float* qstat, int C, int B, int N, int heads, float* scratch, hipStream_t st) {
  if (!lap_ok(N, heads, C, C)) return fail("projected linear attention: unsupported shape (N=%d heads=%d C=%d)", N, heads, C);
  if (!bias || !resid) return fail("projected linear attention: bias and residual are required");
  if (C == 32) return lap_forward_t<1>(xn, wqkv, wout, bias, resid, y, saved, qstat, B, N, heads, scratch, st);
  return lap_forward_t<2>(xn, wqkv, wout, bias, resid, y, saved, qstat, B, N, heads, scratch, st);
}

template <int CB>
static int lap_backward_t(const float* xn, const float* dy, const float* wqkv, const float* wout, const float* saved, const float* qstat,
                          float* dxn, float* dwqk_part, float* dwv_part, float* dwout_part, float* tmp, int B, int N, int heads,
                          float* scratch, hipStream_t st) {
  constexpr int C = 32 * CB;
  const float scale = 0.17677669529663687f;
  const float* kst = saved;
  const float* Mmat = kst + (size_t)B * heads * 64;
  const float* ctx = Mmat + (size_t)B * heads * 32 * C;
  const float* P = ctx + (size_t)B * heads * 1024;
  float* dMmat = tmp;                                    // [B][heads][32][C]
  float* rowdot = tmp + (size_t)B * heads * 32 * C;      // [B][heads][32]
  static bool attr = false;
  if (!attr) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&lap_g_kernel<CB>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 256);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&lap_bwd_kernel<CB, false>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 256);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&lap_mid_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
    attr = true;
  }
  const int np2 = lap_nper(N, C, 2), NS2 = N / np2;
  const size_t lds1 = (size_t)2 * np2 * (C + 4) * sizeof(float);
  const char* spe = knob("PIDM_LAP_SPLIT");                 // 0: the fp32-MFMA pixel-sum kernel
  if (!(spe && !atoi(spe))) {
    const size_t ldss = (size_t)np2 * (6 * C + 16) + (size_t)3 * C * (np2 * 2 + 16);
    static bool attr_s = false;
    if (!attr_s) {
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&lap_g_split_kernel<CB>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 256);
      attr_s = true;
    }
    PIDM_PROF_NAME("lap_g_split_kernel");
    hipLaunchKernelGGL(HIP_KERNEL_NAME(lap_g_split_kernel<CB>), dim3(B * NS2), dim3(512), ldss, st, xn, dy, wqkv, qstat, scratch, N, heads, np2, scale);
  } else {
    hipLaunchKernelGGL(HIP_KERNEL_NAME(lap_g_kernel<CB>), dim3(B * NS2), dim3(512), lds1, st, xn, dy, wqkv, qstat, scratch, N, heads, np2, scale);
  }
  PIDM_CHECK_LAUNCH("lap_g_kernel");
  const size_t lds2 = ((size_t)32 * (C + 1) + (size_t)(C > 33 ? C : 33) * 33 + 32 * (C + 1) + 32 * 33 + 32 * (C + 1) + 64) * sizeof(float);
  hipLaunchKernelGGL(lap_mid_kernel, dim3(B * heads), dim3(kLapSmallNT), lds2, st, scratch, wqkv, wout, ctx, Mmat, dMmat, rowdot, dwout_part, dwv_part, N,
                     heads, C, NS2);
  PIDM_CHECK_LAUNCH("lap_mid_kernel");
  const int np3 = lap_nper(N, C, 3), nsub = lap_nsub(N, C), NS3 = N / np3 / nsub;
  // two tile groups when the heads leave wave slots free: the staged slab then holds two tiles (C = 32: two ranges per staging)
  int G3 = lap_groups(heads, 2), nslab = np3, nsl = nsub;
  if (G3 > 1 && (np3 / 32) % G3) {
    if (nsub % G3 == 0) { nslab = np3 * G3; nsl = nsub / G3; } else G3 = 1;
  }
  const char* ofe = knob("PIDM_LAP_ORDER_FLIP");            // 0: every wave runs a tile's q half before its k half
  const int oflip = (ofe && !atoi(ofe)) ? 0 : 2;
  const int split_dw = !(spe && !atoi(spe)) ? 1 : 0;
  const char* pre = knob("PIDM_LAP_PRIO");                 // 1: static priority for the younger half of lap_bwd's waves (measured: no gain)
  const int trbit = (knob("PIDM_LAP_TRACE") ? 4 : 0) | ((pre && atoi(pre)) ? 8 : 0);
  const char* ppe = knob("PIDM_LAP_SPLIT_PROJ");            // 0: the four per-pixel projections of lap_bwd stay on the fp32 MFMA
  if (CB == 1 && split_dw && !(ppe && !atoi(ppe))) {
    const size_t lds3p = ((size_t)8 * 32 * kLapTileLd + 8 * 96 + (size_t)3 * heads * kLapDH * C) * sizeof(float) + (size_t)3 * C * (nslab * 2) +
                         (size_t)2 * nslab * 208;
    if (lds3p > 160 * 1024 - 256) return fail("lap_bwd: %zu B of LDS", lds3p);
    static bool attr_p = false;
    if (!attr_p) {
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&lap_bwd_kernel<1, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 256);
      attr_p = true;
    }
    PIDM_PROF_NAME("lap_bwd_kernel<1, true>");
    hipLaunchKernelGGL(HIP_KERNEL_NAME(lap_bwd_kernel<1, true>), dim3(B * NS3), dim3(512), lds3p, st, xn, dy, wqkv, P, kst, dMmat, rowdot, dxn,
                       dwqk_part, N, heads, nslab, nsl, scale, 1 | oflip | trbit, G3);
    PIDM_CHECK_LAUNCH("lap_bwd_kernel");
    return 0;
  }
  const size_t lds3 = ((size_t)2 * nslab * (C + 4) + (size_t)8 * CB * 32 * kLapTileLd + 8 * 96 + (CB == 1 ? 3 * heads * kLapDH * (C + 4) : 0)) * sizeof(float) +
                      (size_t)3 * C * (nslab * 2 + (CB == 1 ? 0 : 16));
  if (lds3 > 160 * 1024 - 256) return fail("lap_bwd: %zu B of LDS", lds3);
  hipLaunchKernelGGL(HIP_KERNEL_NAME(lap_bwd_kernel<CB, false>), dim3(B * NS3), dim3(512), lds3, st, xn, dy, wqkv, P, kst, dMmat, rowdot, dxn, dwqk_part,
                     N, heads, nslab, nsl, scale, split_dw | oflip | trbit, G3);
  PIDM_CHECK_LAUNCH("lap_bwd_kernel");
  return 0;
}

size_t lap_bwd_tmp_floats(int B, int heads, int C) { return (size_t)B * heads * (32 * C + 32) + 64; }

// dwqk_part: [B * lap_dw_ranges(N, C)][2*HD][C]; dwv_part: [B][HD][C]; dwout_part: [B][C][HD]; tmp: lap_bwd_tmp_floats
int launch_lap_backward(const float* xn, const float* dy, const float* wqkv, const float* wout, const float* saved, const float* qstat,
                        float* dxn, float* dwqk_part, float* dwv_part, float* dwout_part, float* tmp, int C, int B, int N, int heads,
                        float* scratch, hipStream_t st) {
  if (!lap_ok(N, heads, C, C)) return fail("projected linear attention: unsupported shape (N=%d heads=%d C=%d)", N, heads, C);
  if (C == 32)
    return lap_backward_t<1>(xn, dy, wqkv, wout, saved, qstat, dxn, dwqk_part, dwv_part, dwout_part, tmp, B, N, heads, scratch, st);
  return lap_backward_t<2>(xn, dy, wqkv, wout, saved, qstat, dxn, dwqk_part, dwv_part, dwout_part, tmp, B, N, heads, scratch, st);
}

}  // namespace pidm

using namespace pidm;

// ---- unit-level C ABI (parity tests, tools/bench_attn.py) ---------------------------------------------------------------------
extern "C" size_t pidm_lap_ws(int B, int N, int heads, int C) {
  // scratch | backward tmp | dWq,dWk ranges | dWv per image | dWout per image
  const size_t HD = (size_t)heads * kLapDH;
  return (lap_scratch_floats(B, N, heads, C) + lap_bwd_tmp_floats(B, heads, C) + (size_t)B * lap_dw_ranges(N, C) * 2 * HD * C +
          (size_t)B * HD * C * 2 + 256) * sizeof(float);
}
extern "C" size_t pidm_lap_saved_floats(int B, int heads, int C) { return lap_saved_floats(B, heads, C); }

extern "C" int pidm_lap_forward(const float* xn, const float* w_qkv, const float* w_out, const float* bias, const float* resid, float* y,
                                float* saved, float* qstat, int C, int B, int N, int heads, void* workspace, void* stream) {
  if (!xn || !w_qkv || !w_out || !y || !saved || !qstat || !workspace) return fail("lap_forward: null buffer");
  return launch_lap_forward(xn, w_qkv, w_out, bias, resid, y, saved, qstat, C, B, N, heads, reinterpret_cast<float*>(workspace), as_stream(stream));
}

// d_w_qkv [3*HD][C], d_w_out [C][HD] are WRITTEN (fixed-order sums over images / pixel ranges); d_xn [B][N][C]
extern "C" int pidm_lap_backward(const float* xn, const float* dy, const float* w_qkv, const float* w_out, const float* saved,
                                 const float* qstat, float* d_xn, float* d_w_qkv, float* d_w_out, int C, int B, int N, int heads,
                                 void* workspace, void* stream) {
  if (!xn || !dy || !w_qkv || !w_out || !saved || !qstat || !d_xn || !d_w_qkv || !d_w_out || !workspace) return fail("lap_backward: null buffer");
  if (!lap_ok(N, heads, C, C)) return fail("projected linear attention: unsupported shape (N=%d heads=%d C=%d)", N, heads, C);
  hipStream_t st = as_stream(stream);
  const int HD = heads * kLapDH;
  float* scratch = reinterpret_cast<float*>(workspace);
  float* tmp = scratch + lap_scratch_floats(B, N, heads, C);
  float* dwqk = tmp + lap_bwd_tmp_floats(B, heads, C);
  const int nr = (int)lap_dw_ranges(N, C);
  float* dwv = dwqk + (size_t)B * nr * 2 * HD * C;
  float* dwo = dwv + (size_t)B * HD * C;
  if (launch_lap_backward(xn, dy, w_qkv, w_out, saved, qstat, d_xn, dwqk, dwv, dwo, tmp, C, B, N, heads, scratch, st)) return -1;
  if (launch_split_reduce(dwqk, d_w_qkv, nullptr, nullptr, B * nr, 2 * HD, C, 1, 2 * HD, C, st)) return -1;
  if (launch_split_reduce(dwv, d_w_qkv + (size_t)2 * HD * C, nullptr, nullptr, B, HD, C, 1, HD, C, st)) return -1;
  return launch_split_reduce(dwo, d_w_out, nullptr, nullptr, B, C, HD, 1, C, HD, st);
}

// measurement aid: the cycle stamps lap_bwd_kernel left (PIDM_LAP_TRACE=1; see g_lap_trace)
extern "C" int pidm_debug_lap_trace(unsigned long long* out256) {
  return hipMemcpyFromSymbol(out256, HIP_SYMBOL(pidm::g_lap_trace), sizeof(unsigned long long) * 256) == hipSuccess ? 0 : -1;
}
