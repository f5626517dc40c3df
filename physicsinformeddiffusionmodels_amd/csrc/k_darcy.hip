// Darcy PDE residual, its adjoint, and the fused PIDM loss (forward + d loss / d x0_pred) for gfx950.
//
// Replaces (reference paths): ResidualsDarcy.compute_residual src/residuals_darcy.py:137-183, the 54
// depthwise convolutions + 60 slice scatters of StencilGradientComputation src/grad_utils.py:64-146 and
// the loss algebra of DenoisingDiffusion.model_estimation_loss src/denoising_utils.py:666-692.
//
// One workgroup per (sample, band of rows).  The band's rows of both fields (p, K) plus their stencil halo are staged once
// in LDS, every derivative is a 3/4-tap read of LDS with the one-sided boundary rows selected per pixel, and the adjoint
// is a GATHER over the transposed stencil (no atomics, run-to-run deterministic).  HBM traffic is the
// compulsory 32 KB read + 48 KB residual write + 32 KB gradient write per 64x64 sample (halo rows are re-read from L2).
#include <stdlib.h>

#include "pidm_common.h"

namespace pidm {

// acc-2 finite-difference coefficient tables along one axis, already divided by h^order (fp32, as the
// reference stores them in its conv kernels).  class 0 = low edge (taps at +0,+1,+2[,+3]),
// class 1 = centre (taps at -1,0,+1), class 2 = high edge (taps at -0,-1,-2[,-3]).
struct FdAxis {
  float c1[3][4];
  float c2[3][4];
};

__device__ __forceinline__ float fd_apply(const float (&c)[3][4], const float* a, int i, int P, int s, int ntap_edge) {
  // (D a)[i] for a 1-D line a[k*s], k in [0,P)
  float r = 0.f;
  if (i == 0) {
    for (int k = 0; k < ntap_edge; ++k) r = fmaf(c[0][k], a[k * s], r);
  } else if (i == P - 1) {
    for (int k = 0; k < ntap_edge; ++k) r = fmaf(c[2][k], a[(P - 1 - k) * s], r);
  } else {
    for (int k = 0; k < 3; ++k) r = fmaf(c[1][k], a[(i - 1 + k) * s], r);
  }
  return r;
}

__device__ __forceinline__ float fd_coef(const float (&c)[3][4], int i, int m, int P, int ntap_edge) {
  // D[i][m]
  if (i == 0) return (m < ntap_edge) ? c[0][m] : 0.f;
  if (i == P - 1) return (P - 1 - m < ntap_edge) ? c[2][P - 1 - m] : 0.f;
  int k = m - (i - 1);
  return (k >= 0 && k < 3) ? c[1][k] : 0.f;
}

template <typename F>
__device__ __forceinline__ float fd_apply_T(const float (&c)[3][4], F a_at, int m, int P, int ntap_edge) {
  // (D^T a)[m] = sum_i D[i][m] a[i]; a_at(i) returns a[i]
  float r = 0.f;
  if (m < ntap_edge) r = fmaf(c[0][m], a_at(0), r);
  if (P - 1 - m < ntap_edge) r = fmaf(c[2][P - 1 - m], a_at(P - 1), r);
  for (int i = m - 1; i <= m + 1; ++i) {
    if (i >= 1 && i <= P - 2) r = fmaf(c[1][m - (i - 1)], a_at(i), r);
  }
  return r;
}

enum { DARCY_RES_ONLY = 0, DARCY_BWD = 1, DARCY_LOSS = 2 };

// Row bands.  One workgroup per (sample, band of R rows): with one workgroup per sample a batch of 64 occupied 64 of the 256
// CUs with one wave per SIMD each (112 KB of LDS per workgroup) and the kernel was pure latency (64 us for 7 MB).  A band
// [r0, r1) of gradient rows gathers (transposed stencils) from the intermediate fields of rows [r0-1, r1] plus row 0 / row P-1
// when it lies within 3 rows of them (the one-sided 4-tap edge stencils), and those rows need p and K one row further out
// (rows 0..3 / P-4..P-1 for the edge rows): the halo rows are recomputed by both neighbours, nothing is exchanged.
// R >= 4 keeps the extra rows of the edge stencils inside the first / last band's own halo.
struct DarcyBands {
  int nb, R;
};
// one workgroup per sample (darcy_full_kernel): 64 x 64 fields at batches that fill the chip twice over on their own
static bool darcy_full_on(int B, int P) {
  const char* e = knob("PIDM_DARCY_FULL");     // smallest batch that takes it (0: never)
  const int min_b = e ? atoi(e) : 512;
  return P == 64 && min_b > 0 && B >= min_b;
}
static bool darcy_stream_on() {
  const char* e = knob("PIDM_DARCY_STREAM");     // 0: one workgroup per sample, no cross-sample prefetch (darcy_full_kernel)
  return !(e && !atoi(e));
}
static DarcyBands darcy_bands(int B, int P, bool res_only = false) {
  if (!res_only && darcy_full_on(B, P)) return DarcyBands{1, P};
  int nb = (1024 + B - 1) / B;              // >= 4 workgroups per CU where the batch alone does not provide them
  static int max_rows = 0;                  // rows per band at large batches (PIDM_DARCY_ROWS, measurement knob; >= 4)
  if (!max_rows) {
    const char* e = knob("PIDM_DARCY_ROWS");
    max_rows = e ? atoi(e) : 32;       // four-pixel kernel at batch 4096: 8 rows 343 us, 16 257, 24 275, 32 241, 64 265 (batch 1024: 60.7 / 61.9 / 54.8 / 61.1 for 16 / 24 / 32 / 64)
    if (max_rows < 4) max_rows = 4;
  }
  const int nb_min = (P + max_rows - 1) / max_rows;
  if (nb < nb_min) nb = nb_min;
  int nb_max = P / 4;
  if (nb_max < 1) nb_max = 1;
  if (nb > nb_max) nb = nb_max;
  DarcyBands d;
  d.R = (P + nb - 1) / nb;
  d.nb = (P + d.R - 1) / d.R;
  return d;
}
static int darcy_lds_rows(int P, int R) { return (R + 7 < P) ? R + 7 : P; }

// per-sample loss weights: either gathered by the caller (t == nullptr: p2w[b], inv_var[b]) or looked up here from the
// schedule tables by the sample's time step (p2w = p2_loss_weight table, inv_var = posterior_variance_clipped table; the
// reciprocal is the correctly rounded fp32 division torch's `1.0 / var[t]` performs)
__device__ __forceinline__ float darcy_p2w(const float* __restrict__ p2w, const long long* __restrict__ t, int b) {
  return t ? p2w[t[b]] : p2w[b];
}
__device__ __forceinline__ float darcy_inv_var(const float* __restrict__ v, const long long* __restrict__ t, int b) {
  return t ? 1.0f / v[t[b]] : v[b];
}

// Branch-free stencil taps.  Forward: (D a)[i] = sum_k w[k] a[idx[k]] with 4 (position, weight) pairs chosen by the class of i
// (low edge: rows 0..3, high edge: rows P-1..P-4 in that order, interior: i-1, i, i+1 and a zero-weight fourth tap) - the same
// taps in the same order as fd_apply above, without its three code paths.  Transposed: (D^T a)[m] = sum over the (at most 5)
// stencil rows that touch column m: the interior rows m-1, m, m+1 and the two edge rows 0 and P-1; rows that do not touch m get
// weight 0.  Indices of zero-weight taps are clamped into [lo, hi] (the band's window of LDS rows).
struct FdTaps4 {
  int idx[4];
  float w1[4], w2[4];    // first / second derivative weights
};
__device__ __forceinline__ FdTaps4 fd_taps(const FdAxis& ax, int i, int P, int lo, int hi) {
  FdTaps4 t;
  const bool low = i == 0, high = i == P - 1;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    int ix = high ? P - 1 - k : (low ? k : i - 1 + k);
    ix = ix < lo ? lo : (ix > hi ? hi : ix);
    t.idx[k] = ix;
    const float c1i = (k < 3) ? ax.c1[1][k] : 0.f, c2i = (k < 3) ? ax.c2[1][k] : 0.f;
    t.w1[k] = high ? ax.c1[2][k] : (low ? ax.c1[0][k] : c1i);
    t.w2[k] = high ? ax.c2[2][k] : (low ? ax.c2[0][k] : c2i);
  }
  return t;
}
struct FdTaps5 {
  int idx[5];
  float w1[5], w2[5];
};
__device__ __forceinline__ FdTaps5 fd_taps_T(const FdAxis& ax, int m, int P, int lo, int hi) {
  FdTaps5 t;
  // same order as fd_apply_T: row 0, row P-1, then the interior rows m-1, m, m+1
  const int rows[5] = {0, P - 1, m - 1, m, m + 1};
#pragma unroll
  for (int k = 0; k < 5; ++k) {
    const int i = rows[k];
    float a1 = 0.f, a2 = 0.f;
    if (k == 0) {
      const int q = m < 4 ? m : 3;
      a1 = (m < 3) ? ax.c1[0][q] : 0.f;
      a2 = (m < 4) ? ax.c2[0][q] : 0.f;
    } else if (k == 1) {
      const int d = P - 1 - m, q = d < 4 ? d : 3;
      a1 = (d < 3) ? ax.c1[2][q] : 0.f;
      a2 = (d < 4) ? ax.c2[2][q] : 0.f;
    } else {
      const bool ok = (i >= 1) & (i <= P - 2);
      const int q = m - (i - 1);          // 2, 1, 0 for k = 2, 3, 4
      a1 = ok ? ax.c1[1][q] : 0.f;
      a2 = ok ? ax.c2[1][q] : 0.f;
    }
    t.w1[k] = a1;
    t.w2[k] = a2;
    t.idx[k] = i < lo ? lo : (i > hi ? hi : i);
  }
  return t;
}

// dynamic LDS: MODE 0: 2 fields; MODE 1/2: 8 fields of darcy_lds_rows(P, R) x P floats
template <int MODE>
__global__ void __launch_bounds__(256) darcy_kernel(const float* __restrict__ x0,       // target (MODE 2) [B,2,P,P]
                                                    const float* __restrict__ pred,     // x0_pred [B,2,P,P]
                                                    const float* __restrict__ f_s,      // [P*P]
                                                    const float* __restrict__ grad_res, // MODE 1: [B,P*P,3]
                                                    const float* __restrict__ p2w, const float* __restrict__ inv_var,
                                                    const long long* __restrict__ tsteps,
                                                    float c_data, float c_res, float bc1_sign, FdAxis ax0, FdAxis ax1,
                                                    float* __restrict__ residual, float* __restrict__ grad_pred,
                                                    double* __restrict__ partial,       // MODE 2: [B][nb][4]
                                                    int B, int P, int R, int nb, int lds_rows) {
  HIP_DYNAMIC_SHARED(float, smem)
  const int N = P * P;
  const int b = blockIdx.x / nb, band = blockIdx.x - b * nb;
  const int tid = threadIdx.x;
  const int r0 = band * R, r1 = (r0 + R < P) ? r0 + R : P;
  // rows of intermediates / of p and K this band works on (see above)
  int ilo = r0, ihi = r1;
  if (MODE != DARCY_RES_ONLY) {
    ilo = (r0 < 4) ? 0 : r0 - 1;
    ihi = (r1 > P - 4) ? P : r1 + 1;
  }
  int dlo = (ilo > 0) ? ilo - 1 : 0, dhi = (ihi < P) ? ihi + 1 : P;
  if (ilo == 0 && dhi < 4) dhi = 4;
  if (ihi == P && dlo > P - 4) dlo = P - 4;
  const int F = lds_rows * P;
  // field pointers addressed with ABSOLUTE pixel indices i*P + j, i in [dlo, dhi)
  float* sp = smem - dlo * P;
  float* sK = sp + F;
  float* skg = sp + 2 * F;    // -K g
  float* sa0 = sp + 3 * F;
  float* sa1 = sp + 4 * F;
  float* sb0 = sp + 5 * F;
  float* sb1 = sp + 6 * F;
  float* sd = sp + 7 * F;
  const float* pb = pred + (size_t)b * 2 * N;
  // a thread's column advances by 256 % P and its row by 256 / P (+ carry) per iteration: no division in the loops
  const int dj = 256 % P, di = 256 / P;

  double acc_data = 0.0, acc_r2 = 0.0, acc_rabs = 0.0;
  for (int n = dlo * P + tid; n < dhi * P; n += 256) {
    float vp = pb[n], vK = pb[N + n];
    sp[n] = vp;
    sK[n] = vK;
    if (MODE == DARCY_LOSS && n >= r0 * P && n < r1 * P) {
      float d0 = x0[(size_t)b * 2 * N + n] - vp, d1 = x0[(size_t)b * 2 * N + N + n] - vK;
      acc_data += (double)(d0 * d0) + (double)(d1 * d1);
    }
  }
  __syncthreads();

  const float gscale = (MODE == DARCY_LOSS) ? c_res * darcy_inv_var(inv_var, tsteps, b) / ((float)B * (float)N * 3.0f) : 0.f;
  {
    int i = ilo + tid / P, j = tid - (tid / P) * P;
    FdTaps4 tj = fd_taps(ax1, j, P, 0, P - 1);
    for (int n = ilo * P + tid; n < ihi * P; n += 256) {
      const bool own = (i >= r0) & (i < r1);   // rows this band reports (residual, loss sums); the others are halo recomputation
      const FdTaps4 ti = fd_taps(ax0, i, P, dlo, dhi - 1);
      float p0 = 0.f, p00 = 0.f, K0 = 0.f, p1 = 0.f, p11 = 0.f, K1 = 0.f;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const float pa = sp[ti.idx[k] * P + j], pr = sp[i * P + tj.idx[k]];
        p0 = fmaf(ti.w1[k], pa, p0);
        p00 = fmaf(ti.w2[k], pa, p00);
        p1 = fmaf(tj.w1[k], pr, p1);
        p11 = fmaf(tj.w2[k], pr, p11);
        if (k < 3) {        // the first-derivative stencils have three taps everywhere
          K0 = fmaf(ti.w1[k], sK[ti.idx[k] * P + j], K0);
          K1 = fmaf(tj.w1[k], sK[i * P + tj.idx[k]], K1);
        }
      }
      const float Kv = sK[n];
      // reference op order: vj00 = -K*p00 - K0*p0 ; vj11 = -K*p11 - K1*p1 ; eq = vj00 + vj11 - f_s
      const float vj00 = -Kv * p00 - K0 * p0;
      const float vj11 = -Kv * p11 - K1 * p1;
      const float eq = vj00 + vj11 - f_s[n];
      // boundary rows: bc0 = -p0 (row 0), +p0 (row P-1); bc1 = +p1 (col 0), -p1 (col P-1) when reverse_d1
      // (d1 < 0, bc1_sign = +1), opposite signs otherwise (src/residuals_darcy.py:173-180)
      const float s0 = (i == 0) ? -1.f : ((i == P - 1) ? 1.f : 0.f);
      const float s1 = (j == 0) ? bc1_sign : ((j == P - 1) ? -bc1_sign : 0.f);
      const float bc0 = s0 * p0, bc1 = s1 * p1;
      if (MODE != DARCY_BWD && own) {
        float* r = residual + ((size_t)b * N + n) * 3;
        r[0] = eq;
        r[1] = bc0;
        r[2] = bc1;
      }
      if (MODE != DARCY_RES_ONLY) {
        float g, gb0, gb1;
        if (MODE == DARCY_BWD) {
          const float* gr = grad_res + ((size_t)b * N + n) * 3;
          g = gr[0];
          gb0 = gr[1];
          gb1 = gr[2];
        } else {
          if (own) {
            acc_r2 += (double)(eq * eq) + (double)(bc0 * bc0) + (double)(bc1 * bc1);
            acc_rabs += (double)fabsf(eq) + (double)fabsf(bc0) + (double)fabsf(bc1);
          }
          g = gscale * eq;
          gb0 = gscale * bc0;
          gb1 = gscale * bc1;
        }
        skg[n] = -Kv * g;              // D00^T / D11^T act on -K g
        sa0[n] = -K0 * g + s0 * gb0;   // coefficient on p0
        sa1[n] = -K1 * g + s1 * gb1;   // coefficient on p1
        sb0[n] = -p0 * g;              // coefficient on K0
        sb1[n] = -p1 * g;              // coefficient on K1
        sd[n] = -(p00 + p11) * g;      // direct dependence of eq on K at the same pixel
      }
      j += dj;
      i += di;
      if (j >= P) {
        j -= P;
        ++i;
        tj = fd_taps(ax1, j, P, 0, P - 1);
      } else if (dj) {
        tj = fd_taps(ax1, j, P, 0, P - 1);
      }
    }
  }
  if (MODE == DARCY_RES_ONLY) return;
  __syncthreads();

  const float dscale = (MODE == DARCY_LOSS) ? 2.f * c_data * darcy_p2w(p2w, tsteps, b) / ((float)B * 2.f * (float)N) : 0.f;
  {
    int i = r0 + tid / P, j = tid - (tid / P) * P;
    FdTaps5 tj = fd_taps_T(ax1, j, P, 0, P - 1);
    for (int n = r0 * P + tid; n < r1 * P; n += 256) {
      const FdTaps5 ti = fd_taps_T(ax0, i, P, ilo, ihi - 1);
      // grad wrt p: D00^T(-K g) + D11^T(-K g) + D0^T a0 + D1^T a1;  grad wrt K: direct + D0^T b0 + D1^T b1
      float g00 = 0.f, g11 = 0.f, ga0 = 0.f, ga1 = 0.f, gb0 = 0.f, gb1 = 0.f;
#pragma unroll
      for (int k = 0; k < 5; ++k) {
        const int c0 = ti.idx[k] * P + j, c1 = i * P + tj.idx[k];
        g00 = fmaf(ti.w2[k], skg[c0], g00);
        ga0 = fmaf(ti.w1[k], sa0[c0], ga0);
        gb0 = fmaf(ti.w1[k], sb0[c0], gb0);
        g11 = fmaf(tj.w2[k], skg[c1], g11);
        ga1 = fmaf(tj.w1[k], sa1[c1], ga1);
        gb1 = fmaf(tj.w1[k], sb1[c1], gb1);
      }
      float gp = ((g00 + g11) + ga0) + ga1;
      float gK = (sd[n] + gb0) + gb1;
      if (MODE == DARCY_LOSS) {
        gp += dscale * (sp[n] - x0[(size_t)b * 2 * N + n]);
        gK += dscale * (sK[n] - x0[(size_t)b * 2 * N + N + n]);
      }
      grad_pred[(size_t)b * 2 * N + n] = gp;
      grad_pred[(size_t)b * 2 * N + N + n] = gK;
      j += dj;
      i += di;
      if (j >= P) {
        j -= P;
        ++i;
        tj = fd_taps_T(ax1, j, P, 0, P - 1);
      } else if (dj) {
        tj = fd_taps_T(ax1, j, P, 0, P - 1);
      }
    }
  }

  if (MODE == DARCY_LOSS) {
    // deterministic block reduction of the three partial sums
    __shared__ double red[3][4];
    double v[3] = {acc_data, acc_r2, acc_rabs};
    for (int q = 0; q < 3; ++q) {
      double x = v[q];
      for (int off = 32; off > 0; off >>= 1) x += __shfl_down(x, off);
      if ((tid & 63) == 0) red[q][tid >> 6] = x;
    }
    __syncthreads();
    if (tid < 3) {
      double s = red[tid][0] + red[tid][1] + red[tid][2] + red[tid][3];
      partial[(size_t)blockIdx.x * 4 + tid] = s;
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------------------
// The same kernel with FOUR pixels of a row per thread (P / 4 a power of two: the 64 x 64 fields of every configuration): 16-byte
// LDS and global accesses (the residual's 4 x 3 floats leave as three float4 stores instead of twelve scattered dwords), the
// column-direction taps (rows i-1, i, i+1 or the edge rows) are shared by the four pixels, the row-direction taps come from a
// six-value window a[j0-1 .. j0+4] (one 16-byte read + two scalars) - only the first and the last quad of a row contain an edge
// pixel, and those are handled by selects on pixel 0 / pixel 3.  ~4x fewer LDS instructions and ~3x fewer VALU instructions per
// pixel than darcy_kernel: at batch 4096 the scalar kernel was LDS / VALU bound at 1.3 TB/s of algorithmic traffic.
// Same bands, same LDS layout, same summation order within each stencil as darcy_kernel.
// ---------------------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ float fd_edge3(const float (&c)[4], float a0, float a1, float a2) { return fmaf(c[2], a2, fmaf(c[1], a1, c[0] * a0)); }
__device__ __forceinline__ float fd_edge4(const float (&c)[4], float a0, float a1, float a2, float a3) {
  return fmaf(c[3], a3, fmaf(c[2], a2, fmaf(c[1], a1, c[0] * a0)));
}
// row-direction derivatives of the quad's four pixels from the window w[0..5] = a[j0-1 .. j0+4]
__device__ __forceinline__ void quad_d1(const FdAxis& ax, const float (&w)[6], bool lowq, bool highq, float (&d)[4]) {
#pragma unroll
  for (int m = 0; m < 4; ++m) d[m] = fd_edge3(ax.c1[1], w[m], w[m + 1], w[m + 2]);
  const float lo = fd_edge3(ax.c1[0], w[1], w[2], w[3]), hi = fd_edge3(ax.c1[2], w[4], w[3], w[2]);
  d[0] = lowq ? lo : d[0];
  d[3] = highq ? hi : d[3];
}
__device__ __forceinline__ void quad_d2(const FdAxis& ax, const float (&w)[6], bool lowq, bool highq, float (&d)[4]) {
#pragma unroll
  for (int m = 0; m < 4; ++m) d[m] = fd_edge3(ax.c2[1], w[m], w[m + 1], w[m + 2]);
  const float lo = fd_edge4(ax.c2[0], w[1], w[2], w[3], w[4]), hi = fd_edge4(ax.c2[2], w[4], w[3], w[2], w[1]);
  d[0] = lowq ? lo : d[0];
  d[3] = highq ? hi : d[3];
}
// transposed row-direction stencil: weights of the quad's four columns on a[0], a[P-1] and the window positions c-1, c, c+1
struct QuadT {
  float e0[4], eP[4], m1[4], z0[4], p1[4];
};
__device__ __forceinline__ QuadT quad_T(const float (&c)[3][4], int j0, int P, int nt) {
  QuadT t;
#pragma unroll
  for (int m = 0; m < 4; ++m) {
    const int col = j0 + m, d = P - 1 - col;
    const int qa = col < 4 ? col : 3, qb = d < 4 ? d : 3;
    t.e0[m] = (col < nt) ? c[0][qa] : 0.f;
    t.eP[m] = (d < nt) ? c[2][qb] : 0.f;
    t.m1[m] = (col - 1 >= 1 && col - 1 <= P - 2) ? c[1][2] : 0.f;
    t.z0[m] = (col >= 1 && col <= P - 2) ? c[1][1] : 0.f;
    t.p1[m] = (col + 1 >= 1 && col + 1 <= P - 2) ? c[1][0] : 0.f;
  }
  return t;
}
__device__ __forceinline__ void quad_gather(const QuadT& t, float a0, float aP, const float (&w)[6], float (&g)[4]) {
#pragma unroll
  for (int m = 0; m < 4; ++m) {
    float r = t.e0[m] * a0;                  // same order as fd_apply_T: row 0, row P-1, then m-1, m, m+1
    r = fmaf(t.eP[m], aP, r);
    r = fmaf(t.m1[m], w[m], r);
    r = fmaf(t.z0[m], w[m + 1], r);
    g[m] = fmaf(t.p1[m], w[m + 2], r);
  }
}
// window a[j0-1 .. j0+4] of row `rowp` (pointer to the row's column 0); the two outer values are clamped at the row ends
// (their weights are zero there)
__device__ __forceinline__ void quad_window(const float* rowp, int j0, int P, float (&w)[6]) {
  const f32x4 v = *reinterpret_cast<const f32x4*>(rowp + j0);
  w[0] = rowp[j0 > 0 ? j0 - 1 : 0];
  w[1] = v[0]; w[2] = v[1]; w[3] = v[2]; w[4] = v[3];
  w[5] = rowp[j0 + 4 < P ? j0 + 4 : P - 1];
}

template <int MODE>
__global__ void __launch_bounds__(512) darcy_quad_kernel(const float* __restrict__ x0, const float* __restrict__ pred,
                                                         const float* __restrict__ f_s, const float* __restrict__ grad_res,
                                                         const float* __restrict__ p2w, const float* __restrict__ inv_var,
                                                         const long long* __restrict__ tsteps, float c_data, float c_res, float bc1_sign,
                                                         FdAxis ax0, FdAxis ax1, float* __restrict__ residual, float* __restrict__ grad_pred,
                                                         double* __restrict__ partial, int B, int P, int R, int nb, int lds_rows) {
  HIP_DYNAMIC_SHARED(float, smem)
  const int N = P * P, QP = P >> 2;           // quads per row (a power of two <= 256)
  const int b = blockIdx.x / nb, band = blockIdx.x - b * nb;
  const int tid = threadIdx.x;
  const int r0 = band * R, r1 = (r0 + R < P) ? r0 + R : P;
  int ilo = r0, ihi = r1;
  if (MODE != DARCY_RES_ONLY) {
    ilo = (r0 < 4) ? 0 : r0 - 1;
    ihi = (r1 > P - 4) ? P : r1 + 1;
  }
  int dlo = (ilo > 0) ? ilo - 1 : 0, dhi = (ihi < P) ? ihi + 1 : P;
  if (ilo == 0 && dhi < 4) dhi = 4;
  if (ihi == P && dlo > P - 4) dlo = P - 4;
  const int F = lds_rows * P;
  float* sp = smem - dlo * P;
  float* sK = sp + F;
  float* skg = sp + 2 * F;
  float* sa0 = sp + 3 * F;
  float* sa1 = sp + 4 * F;
  float* sb0 = sp + 5 * F;
  float* sb1 = sp + 6 * F;
  float* sd = sp + 7 * F;
  const float* pb = pred + (size_t)b * 2 * N;
  const float* tb = x0 + (size_t)b * 2 * N;
  // block size = quads per row x (rows of the band + halo), so that every pass is ONE sweep (no second, nearly empty iteration)
  const int q = tid & (QP - 1), j0 = 4 * q, rstep = (int)blockDim.x / QP, rt = tid / QP;   // this thread's quad column; rows rt, rt + rstep, ...
  const bool lowq = q == 0, highq = q == QP - 1;

  double acc_data = 0.0, acc_r2 = 0.0, acc_rabs = 0.0;
  for (int i = dlo + rt; i < dhi; i += rstep) {
    const int n = i * P + j0;
    const f32x4 vp = *reinterpret_cast<const f32x4*>(pb + n), vK = *reinterpret_cast<const f32x4*>(pb + N + n);
    *reinterpret_cast<f32x4*>(sp + n) = vp;
    *reinterpret_cast<f32x4*>(sK + n) = vK;
    if (MODE == DARCY_LOSS && i >= r0 && i < r1) {
      const f32x4 t0 = *reinterpret_cast<const f32x4*>(tb + n), t1 = *reinterpret_cast<const f32x4*>(tb + N + n);
#pragma unroll
      for (int m = 0; m < 4; ++m) {
        const float d0 = t0[m] - vp[m], d1 = t1[m] - vK[m];
        acc_data += (double)(d0 * d0) + (double)(d1 * d1);
      }
    }
  }
  __syncthreads();

  const float gscale = (MODE == DARCY_LOSS) ? c_res * darcy_inv_var(inv_var, tsteps, b) / ((float)B * (float)N * 3.0f) : 0.f;
  for (int i = ilo + rt; i < ihi; i += rstep) {
    const int n = i * P + j0;
    const bool own = (i >= r0) & (i < r1);
    const FdTaps4 ti = fd_taps(ax0, i, P, dlo, dhi - 1);
    float p0[4] = {0.f, 0.f, 0.f, 0.f}, p00[4] = {0.f, 0.f, 0.f, 0.f}, K0[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const f32x4 pa = *reinterpret_cast<const f32x4*>(sp + ti.idx[k] * P + j0);
#pragma unroll
      for (int m = 0; m < 4; ++m) {
        p0[m] = fmaf(ti.w1[k], pa[m], p0[m]);
        p00[m] = fmaf(ti.w2[k], pa[m], p00[m]);
      }
      if (k < 3) {
        const f32x4 ka = *reinterpret_cast<const f32x4*>(sK + ti.idx[k] * P + j0);
#pragma unroll
        for (int m = 0; m < 4; ++m) K0[m] = fmaf(ti.w1[k], ka[m], K0[m]);
      }
    }
    float wp[6], wk[6], p1[4], p11[4], K1[4];
    quad_window(sp + i * P, j0, P, wp);
    quad_window(sK + i * P, j0, P, wk);
    quad_d1(ax1, wp, lowq, highq, p1);
    quad_d2(ax1, wp, lowq, highq, p11);
    quad_d1(ax1, wk, lowq, highq, K1);
    const f32x4 fs4 = *reinterpret_cast<const f32x4*>(f_s + n);
    const float s0 = (i == 0) ? -1.f : ((i == P - 1) ? 1.f : 0.f);
    float eq[4], bc0[4], bc1[4];
#pragma unroll
    for (int m = 0; m < 4; ++m) {
      const float Kv = wk[m + 1];
      // reference op order: vj00 = -K*p00 - K0*p0 ; vj11 = -K*p11 - K1*p1 ; eq = vj00 + vj11 - f_s
      const float vj00 = -Kv * p00[m] - K0[m] * p0[m];
      const float vj11 = -Kv * p11[m] - K1[m] * p1[m];
      eq[m] = vj00 + vj11 - fs4[m];
      const float s1 = (m == 0 && lowq) ? bc1_sign : ((m == 3 && highq) ? -bc1_sign : 0.f);
      bc0[m] = s0 * p0[m];
      bc1[m] = s1 * p1[m];
    }
    if (MODE != DARCY_BWD && own) {
      f32x4* r = reinterpret_cast<f32x4*>(residual + ((size_t)b * N + n) * 3);
      r[0] = f32x4{eq[0], bc0[0], bc1[0], eq[1]};
      r[1] = f32x4{bc0[1], bc1[1], eq[2], bc0[2]};
      r[2] = f32x4{bc1[2], eq[3], bc0[3], bc1[3]};
    }
    if (MODE != DARCY_RES_ONLY) {
      float g[4], gb0[4], gb1[4];
      if (MODE == DARCY_BWD) {
        const f32x4* gr = reinterpret_cast<const f32x4*>(grad_res + ((size_t)b * N + n) * 3);
        const f32x4 a = gr[0], c = gr[1], e = gr[2];
        g[0] = a[0]; gb0[0] = a[1]; gb1[0] = a[2]; g[1] = a[3];
        gb0[1] = c[0]; gb1[1] = c[1]; g[2] = c[2]; gb0[2] = c[3];
        gb1[2] = e[0]; g[3] = e[1]; gb0[3] = e[2]; gb1[3] = e[3];
      } else {
#pragma unroll
        for (int m = 0; m < 4; ++m) {
          if (own) {
            acc_r2 += (double)(eq[m] * eq[m]) + (double)(bc0[m] * bc0[m]) + (double)(bc1[m] * bc1[m]);
            acc_rabs += (double)fabsf(eq[m]) + (double)fabsf(bc0[m]) + (double)fabsf(bc1[m]);
          }
          g[m] = gscale * eq[m];
          gb0[m] = gscale * bc0[m];
          gb1[m] = gscale * bc1[m];
        }
      }
      f32x4 o_kg, o_a0, o_a1, o_b0, o_b1, o_d;
#pragma unroll
      for (int m = 0; m < 4; ++m) {
        const float s1 = (m == 0 && lowq) ? bc1_sign : ((m == 3 && highq) ? -bc1_sign : 0.f);
        o_kg[m] = -wk[m + 1] * g[m];
        o_a0[m] = -K0[m] * g[m] + s0 * gb0[m];
        o_a1[m] = -K1[m] * g[m] + s1 * gb1[m];
        o_b0[m] = -p0[m] * g[m];
        o_b1[m] = -p1[m] * g[m];
        o_d[m] = -(p00[m] + p11[m]) * g[m];
      }
      *reinterpret_cast<f32x4*>(skg + n) = o_kg;
      *reinterpret_cast<f32x4*>(sa0 + n) = o_a0;
      *reinterpret_cast<f32x4*>(sa1 + n) = o_a1;
      *reinterpret_cast<f32x4*>(sb0 + n) = o_b0;
      *reinterpret_cast<f32x4*>(sb1 + n) = o_b1;
      *reinterpret_cast<f32x4*>(sd + n) = o_d;
    }
  }
  if (MODE == DARCY_RES_ONLY) return;
  __syncthreads();

  const float dscale = (MODE == DARCY_LOSS) ? 2.f * c_data * darcy_p2w(p2w, tsteps, b) / ((float)B * 2.f * (float)N) : 0.f;
  const QuadT t1 = quad_T(ax1.c1, j0, P, 3), t2 = quad_T(ax1.c2, j0, P, 4);
  for (int i = r0 + rt; i < r1; i += rstep) {
    const int n = i * P + j0;
    const FdTaps5 ti = fd_taps_T(ax0, i, P, ilo, ihi - 1);
    float g00[4] = {0.f, 0.f, 0.f, 0.f}, ga0[4] = {0.f, 0.f, 0.f, 0.f}, gb0[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int k = 0; k < 5; ++k) {
      const int c0 = ti.idx[k] * P + j0;
      const f32x4 vkg = *reinterpret_cast<const f32x4*>(skg + c0), va = *reinterpret_cast<const f32x4*>(sa0 + c0),
                  vb = *reinterpret_cast<const f32x4*>(sb0 + c0);
#pragma unroll
      for (int m = 0; m < 4; ++m) {
        g00[m] = fmaf(ti.w2[k], vkg[m], g00[m]);
        ga0[m] = fmaf(ti.w1[k], va[m], ga0[m]);
        gb0[m] = fmaf(ti.w1[k], vb[m], gb0[m]);
      }
    }
    float w[6], g11[4], ga1[4], gb1[4];
    const int row = i * P;
    quad_window(skg + row, j0, P, w);
    quad_gather(t2, skg[row], skg[row + P - 1], w, g11);
    quad_window(sa1 + row, j0, P, w);
    quad_gather(t1, sa1[row], sa1[row + P - 1], w, ga1);
    quad_window(sb1 + row, j0, P, w);
    quad_gather(t1, sb1[row], sb1[row + P - 1], w, gb1);
    const f32x4 vd = *reinterpret_cast<const f32x4*>(sd + n);
    f32x4 gp, gK;
#pragma unroll
    for (int m = 0; m < 4; ++m) {
      gp[m] = ((g00[m] + g11[m]) + ga0[m]) + ga1[m];
      gK[m] = (vd[m] + gb0[m]) + gb1[m];
    }
    if (MODE == DARCY_LOSS) {
      const f32x4 vp = *reinterpret_cast<const f32x4*>(sp + n), vK = *reinterpret_cast<const f32x4*>(sK + n);
      const f32x4 t0 = *reinterpret_cast<const f32x4*>(tb + n), t1v = *reinterpret_cast<const f32x4*>(tb + N + n);
#pragma unroll
      for (int m = 0; m < 4; ++m) {
        gp[m] += dscale * (vp[m] - t0[m]);
        gK[m] += dscale * (vK[m] - t1v[m]);
      }
    }
    *reinterpret_cast<f32x4*>(grad_pred + (size_t)b * 2 * N + n) = gp;
    *reinterpret_cast<f32x4*>(grad_pred + (size_t)b * 2 * N + N + n) = gK;
  }

  if (MODE == DARCY_LOSS) {
    __shared__ double red[3][8];
    double v[3] = {acc_data, acc_r2, acc_rabs};
    for (int qq = 0; qq < 3; ++qq) {
      double x = v[qq];
      for (int off = 32; off > 0; off >>= 1) x += __shfl_down(x, off);
      if ((tid & 63) == 0) red[qq][tid >> 6] = x;
    }
    __syncthreads();
    if (tid < 3) {
      double sv = 0.0;
      for (int w = 0; w < (int)blockDim.x >> 6; ++w) sv += red[tid][w];     // fixed order
      partial[(size_t)blockIdx.x * 4 + tid] = sv;
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------------------
// Large batches (round 6): ONE workgroup per 64 x 64 sample, no bands - no halo rows to reload and recompute - and only FIVE fields
// in LDS (p, K and the three intermediates whose transposed stencils run across rows: -K g, the coefficients on p0 and K0): 80 KB.
// Everything a pixel's OWN thread needs later stays in its registers - thread (row slot rt, quad q) owns rows rt, rt + NT / 16, ...
// in every pass: the intermediates of the row-direction stencils (their neighbours come by 16-lane shuffles: the 16 quads of a row
// are 16 consecutive lanes), the direct K term, prediction - target (read once, for the data loss AND its gradient).
// Same arithmetic in the same order as darcy_quad_kernel (same taps, same windows, same gathers): residual and gradients are
// bit-identical; the loss partial sums are per sample instead of per band (one more grouping of the same doubles).
// Replaces the same reference lines as darcy_quad_kernel for B >= 512 (PIDM_DARCY_FULL=<smallest batch>, 0: off).
// ---------------------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ void quad_window_regs(const float (&v)[4], bool lowq, bool highq, float (&w)[6]) {
  const float left = __shfl_up(v[3], 1, 16), right = __shfl_down(v[0], 1, 16);
  w[0] = lowq ? v[0] : left;
  w[1] = v[0]; w[2] = v[1]; w[3] = v[2]; w[4] = v[3];
  w[5] = highq ? v[3] : right;
}
template <int MODE, int NT>
__global__ void __launch_bounds__(NT, 2) darcy_full_kernel(const float* __restrict__ x0, const float* __restrict__ pred,
                                                        const float* __restrict__ f_s, const float* __restrict__ grad_res,
                                                        const float* __restrict__ p2w, const float* __restrict__ inv_var,
                                                        const long long* __restrict__ tsteps, float c_data, float c_res, float bc1_sign,
                                                        FdAxis ax0, FdAxis ax1, float* __restrict__ residual, float* __restrict__ grad_pred,
                                                        double* __restrict__ partial, int B) {
  constexpr int P = 64, N = P * P, RS = NT / 16, NIT = P / RS;     // row step, rows per thread
  HIP_DYNAMIC_SHARED(float, smem)
  float* sp = smem;
  float* sK = sp + N;
  float* skg = sp + 2 * N;
  float* sa0 = sp + 3 * N;
  float* sb0 = sp + 4 * N;
  const int b = blockIdx.x, tid = threadIdx.x;
  const int q = tid & 15, j0 = 4 * q, rt = tid >> 4;
  const bool lowq = q == 0, highq = q == 15;
  const float* pb = pred + (size_t)b * 2 * N;
  const float* tb = x0 + (size_t)b * 2 * N;
  f32x4 vp[NIT], vK[NIT], t0[NIT], t1[NIT];
#pragma unroll
  for (int it = 0; it < NIT; ++it) {
    const int n = (rt + RS * it) * P + j0;
    vp[it] = *reinterpret_cast<const f32x4*>(pb + n);
    vK[it] = *reinterpret_cast<const f32x4*>(pb + N + n);
    if (MODE == DARCY_LOSS) {
      t0[it] = *reinterpret_cast<const f32x4*>(tb + n);
      t1[it] = *reinterpret_cast<const f32x4*>(tb + N + n);
    }
  }
  double acc_data = 0.0, acc_r2 = 0.0, acc_rabs = 0.0;
#pragma unroll
  for (int it = 0; it < NIT; ++it) {
    const int n = (rt + RS * it) * P + j0;
    *reinterpret_cast<f32x4*>(sp + n) = vp[it];
    *reinterpret_cast<f32x4*>(sK + n) = vK[it];
    if (MODE == DARCY_LOSS) {
#pragma unroll
      for (int m = 0; m < 4; ++m) {
        // kept as prediction - target for the gradient below; its square is the reference's (target - prediction)^2 bit for bit
        t0[it][m] = vp[it][m] - t0[it][m];
        t1[it][m] = vK[it][m] - t1[it][m];
        acc_data += (double)(t0[it][m] * t0[it][m]) + (double)(t1[it][m] * t1[it][m]);
      }
    }
  }
  __syncthreads();

  const float gscale = (MODE == DARCY_LOSS) ? c_res * darcy_inv_var(inv_var, tsteps, b) / ((float)B * (float)N * 3.0f) : 0.f;
  // own-pixel values kept for the last pass: -K g, coefficients on p1 / K1, the direct K term
  float rkg[NIT][4], ra1[NIT][4], rb1[NIT][4], rd[NIT][4];
#pragma unroll
  for (int it = 0; it < NIT; ++it) {
    const int i = rt + RS * it, n = i * P + j0;
    const FdTaps4 ti = fd_taps(ax0, i, P, 0, P - 1);
    float p0[4] = {0.f, 0.f, 0.f, 0.f}, p00[4] = {0.f, 0.f, 0.f, 0.f}, K0[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const f32x4 pa = *reinterpret_cast<const f32x4*>(sp + ti.idx[k] * P + j0);
#pragma unroll
      for (int m = 0; m < 4; ++m) {
        p0[m] = fmaf(ti.w1[k], pa[m], p0[m]);
        p00[m] = fmaf(ti.w2[k], pa[m], p00[m]);
      }
      if (k < 3) {
        const f32x4 ka = *reinterpret_cast<const f32x4*>(sK + ti.idx[k] * P + j0);
#pragma unroll
        for (int m = 0; m < 4; ++m) K0[m] = fmaf(ti.w1[k], ka[m], K0[m]);
      }
    }
    float wp[6], wk[6], p1[4], p11[4], K1[4];
    const float pv[4] = {vp[it][0], vp[it][1], vp[it][2], vp[it][3]}, kv[4] = {vK[it][0], vK[it][1], vK[it][2], vK[it][3]};
    quad_window_regs(pv, lowq, highq, wp);
    quad_window_regs(kv, lowq, highq, wk);
    quad_d1(ax1, wp, lowq, highq, p1);
    quad_d2(ax1, wp, lowq, highq, p11);
    quad_d1(ax1, wk, lowq, highq, K1);
    const f32x4 fs4 = *reinterpret_cast<const f32x4*>(f_s + n);
    const float s0 = (i == 0) ? -1.f : ((i == P - 1) ? 1.f : 0.f);
    float eq[4], bc0[4], bc1[4];
#pragma unroll
    for (int m = 0; m < 4; ++m) {
      const float Kv = wk[m + 1];
      const float vj00 = -Kv * p00[m] - K0[m] * p0[m];
      const float vj11 = -Kv * p11[m] - K1[m] * p1[m];
      eq[m] = vj00 + vj11 - fs4[m];
      const float s1 = (m == 0 && lowq) ? bc1_sign : ((m == 3 && highq) ? -bc1_sign : 0.f);
      bc0[m] = s0 * p0[m];
      bc1[m] = s1 * p1[m];
    }
    if (MODE != DARCY_BWD) {
      f32x4* r = reinterpret_cast<f32x4*>(residual + ((size_t)b * N + n) * 3);
      r[0] = f32x4{eq[0], bc0[0], bc1[0], eq[1]};
      r[1] = f32x4{bc0[1], bc1[1], eq[2], bc0[2]};
      r[2] = f32x4{bc1[2], eq[3], bc0[3], bc1[3]};
    }
    float g[4], gb0[4], gb1[4];
    if (MODE == DARCY_BWD) {
      const f32x4* gr = reinterpret_cast<const f32x4*>(grad_res + ((size_t)b * N + n) * 3);
      const f32x4 a = gr[0], c = gr[1], e = gr[2];
      g[0] = a[0]; gb0[0] = a[1]; gb1[0] = a[2]; g[1] = a[3];
      gb0[1] = c[0]; gb1[1] = c[1]; g[2] = c[2]; gb0[2] = c[3];
      gb1[2] = e[0]; g[3] = e[1]; gb0[3] = e[2]; gb1[3] = e[3];
    } else {
#pragma unroll
      for (int m = 0; m < 4; ++m) {
        // (eq^2 + bc0^2) + bc1^2 in double, as the band kernel: bc0 is an exact zero off the first / last row and bc1 off the
        // first / last column, and adding those zeros changes no bit - they are skipped
        double r2 = (double)(eq[m] * eq[m]), ra = (double)fabsf(eq[m]);
        if (s0 != 0.f) {
          r2 += (double)(bc0[m] * bc0[m]);
          ra += (double)fabsf(bc0[m]);
        }
        if (m == 0 || m == 3) {
          r2 += (double)(bc1[m] * bc1[m]);
          ra += (double)fabsf(bc1[m]);
        }
        acc_r2 += r2;
        acc_rabs += ra;
        g[m] = gscale * eq[m];
        gb0[m] = gscale * bc0[m];
        gb1[m] = gscale * bc1[m];
      }
    }
    f32x4 o_kg, o_a0, o_b0;
#pragma unroll
    for (int m = 0; m < 4; ++m) {
      const float s1 = (m == 0 && lowq) ? bc1_sign : ((m == 3 && highq) ? -bc1_sign : 0.f);
      o_kg[m] = -wk[m + 1] * g[m];
      o_a0[m] = -K0[m] * g[m] + s0 * gb0[m];
      o_b0[m] = -p0[m] * g[m];
      rkg[it][m] = o_kg[m];
      ra1[it][m] = -K1[m] * g[m] + s1 * gb1[m];
      rb1[it][m] = -p1[m] * g[m];
      rd[it][m] = -(p00[m] + p11[m]) * g[m];
      PIDM_OPAQUE_F32(rd[it][m]);     // a ROUNDED product, as the band kernel's round trip through LDS makes it: no fma with the sum below
    }
    *reinterpret_cast<f32x4*>(skg + n) = o_kg;
    *reinterpret_cast<f32x4*>(sa0 + n) = o_a0;
    *reinterpret_cast<f32x4*>(sb0 + n) = o_b0;
  }
  __syncthreads();

  const float dscale = (MODE == DARCY_LOSS) ? 2.f * c_data * darcy_p2w(p2w, tsteps, b) / ((float)B * 2.f * (float)N) : 0.f;
  const QuadT tq1 = quad_T(ax1.c1, j0, P, 3), tq2 = quad_T(ax1.c2, j0, P, 4);
#pragma unroll
  for (int it = 0; it < NIT; ++it) {
    const int i = rt + RS * it, n = i * P + j0;
    const FdTaps5 ti = fd_taps_T(ax0, i, P, 0, P - 1);
    float g00[4] = {0.f, 0.f, 0.f, 0.f}, ga0[4] = {0.f, 0.f, 0.f, 0.f}, gb0[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int k = 0; k < 5; ++k) {
      const int c0 = ti.idx[k] * P + j0;
      const f32x4 vkg = *reinterpret_cast<const f32x4*>(skg + c0), va = *reinterpret_cast<const f32x4*>(sa0 + c0),
                  vb = *reinterpret_cast<const f32x4*>(sb0 + c0);
#pragma unroll
      for (int m = 0; m < 4; ++m) {
        g00[m] = fmaf(ti.w2[k], vkg[m], g00[m]);
        ga0[m] = fmaf(ti.w1[k], va[m], ga0[m]);
        gb0[m] = fmaf(ti.w1[k], vb[m], gb0[m]);
      }
    }
    // row-direction gathers: the row's own values by 16-lane shuffles (first / last element of the row: lanes 0 / 15 of the group)
    float w[6], g11[4], ga1[4], gb1[4];
    quad_window_regs(rkg[it], lowq, highq, w);
    quad_gather(tq2, __shfl(rkg[it][0], 0, 16), __shfl(rkg[it][3], 15, 16), w, g11);
    quad_window_regs(ra1[it], lowq, highq, w);
    quad_gather(tq1, __shfl(ra1[it][0], 0, 16), __shfl(ra1[it][3], 15, 16), w, ga1);
    quad_window_regs(rb1[it], lowq, highq, w);
    quad_gather(tq1, __shfl(rb1[it][0], 0, 16), __shfl(rb1[it][3], 15, 16), w, gb1);
    f32x4 gp, gK;
#pragma unroll
    for (int m = 0; m < 4; ++m) {
      gp[m] = ((g00[m] + g11[m]) + ga0[m]) + ga1[m];
      gK[m] = (rd[it][m] + gb0[m]) + gb1[m];
    }
    if (MODE == DARCY_LOSS) {
#pragma unroll
      for (int m = 0; m < 4; ++m) {
        gp[m] += dscale * t0[it][m];
        gK[m] += dscale * t1[it][m];
      }
    }
    *reinterpret_cast<f32x4*>(grad_pred + (size_t)b * 2 * N + n) = gp;
    *reinterpret_cast<f32x4*>(grad_pred + (size_t)b * 2 * N + N + n) = gK;
  }

  if (MODE == DARCY_LOSS) {
    double (*red)[NT / 64] = reinterpret_cast<double (*)[NT / 64]>(sp);     // p is dead since the second barrier
    double v[3] = {acc_data, acc_r2, acc_rabs};
    for (int qq = 0; qq < 3; ++qq) {
      double x = v[qq];
      for (int off = 32; off > 0; off >>= 1) x += __shfl_down(x, off);
      if ((tid & 63) == 0) red[qq][tid >> 6] = x;
    }
    __syncthreads();
    if (tid < 3) {
      double sv = 0.0;
      for (int w = 0; w < NT / 64; ++w) sv += red[tid][w];     // fixed order
      partial[(size_t)blockIdx.x * 4 + tid] = sv;
    }
  }
}

// Cycle stamps of workgroup 0's first wave (tools/darcy_trace.py; compiled in only with -DPIDM_DARCY_TRACE_BUILD=1: the product
// kernel carries none): per sample, [0] past the top barrier, [1] end of the first pass, [2] past the second barrier, [3] end.
#ifndef PIDM_DARCY_TRACE_BUILD
#define PIDM_DARCY_TRACE_BUILD 0
#endif
#if PIDM_DARCY_TRACE_BUILD
__device__ unsigned long long g_darcy_trace[4 * 64];
#define PIDM_DARCY_STAMP(slot_) \
  do { if (blockIdx.x == 0 && threadIdx.x == 0 && n_done < 64) g_darcy_trace[4 * n_done + (slot_)] = clock64(); } while (0)
#else
#define PIDM_DARCY_STAMP(slot_) ((void)0)
#endif
// ---------------------------------------------------------------------------------------------------------------------------
// The fused loss at large batches, streaming: darcy_full_kernel<DARCY_LOSS> as a PERSISTENT workgroup per CU that walks samples
// b, b + gridDim.x, ... and has the NEXT sample's inputs copied global -> LDS (global_load_lds, no registers) while it computes the
// current one.  One workgroup of 8 waves per CU runs load -> barrier -> stencils -> barrier -> adjoint stencils strictly in turn
// (PMC, batch 4096: vector ALU busy 37 % of the kernel, memory about as much, nothing overlapped); here the loads of sample s + 1
// are in flight during passes 1 and 2 of sample s and the stores drain behind the next sample's arithmetic.
// LDS: p double-buffered 2 x 16 KB, K 16 KB, target 32 KB, the three column-stencil intermediates 48 KB, f_s 16 KB, 192 B of loss sums.
// The copies are issued as instructions hipcc does not track (pidm_glds_b128_untracked; it would answer the first LDS read behind
// a tracked copy with vmcnt(0)); nothing else LOADS from global memory inside the loop (f_s sits in LDS, the per-sample weights are
// scalar loads), so the only vector-memory wait is the explicit one in front of the top barrier.
// Arithmetic, order and results: exactly darcy_full_kernel's.
// ---------------------------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(512, 2) darcy_stream_kernel(const float* __restrict__ x0, const float* __restrict__ pred,
                                                              const float* __restrict__ f_s, const float* __restrict__ p2w,
                                                              const float* __restrict__ inv_var, const long long* __restrict__ tsteps,
                                                              float c_data, float c_res, float bc1_sign, FdAxis ax0, FdAxis ax1,
                                                              float* __restrict__ residual, float* __restrict__ grad_pred,
                                                              double* __restrict__ partial, int B) {
  constexpr int P = 64, N = P * P, NT = 512, RS = NT / 16, NIT = P / RS;
  HIP_DYNAMIC_SHARED(float, smem)
  float* spb = smem;                 // [2] p, double-buffered: the next sample's lands while this one's column stencils read
  float* sKb = smem + 2 * N;         // K: read in the first pass only, the next sample's is copied behind the second barrier
  float* tg = smem + 3 * N;          // target [p, K]: likewise
  float* skg = smem + 5 * N;
  float* sa0 = smem + 6 * N;
  float* sb0 = smem + 7 * N;
  float* sfs = smem + 8 * N;         // f_s, the same for every sample (in registers it cost the kernel its last ones: spills)
  double (*red)[NT / 64] = reinterpret_cast<double (*)[NT / 64]>(smem + 9 * N);
  // The stencil taps depend on the row (column stencils) and on the quad (row stencils) only.  Per thread and sample they were ~250
  // vector instructions of selects; kept in registers across the sample loop they were ~100 registers.  They are computed once per
  // workgroup into LDS tables (64 rows x (12 + 16) dwords, 16 quads x 40 dwords = 9.5 KB) and read back per sample: the 16 lanes of
  // a row read one address.
  FdTaps4* tab4 = reinterpret_cast<FdTaps4*>(smem + 9 * N + 48);
  int* tab5 = reinterpret_cast<int*>(smem + 9 * N + 48 + 64 * 12);              // FdTaps5 in slots of 16 dwords
  QuadT* tabq = reinterpret_cast<QuadT*>(smem + 9 * N + 48 + 64 * 12 + 64 * 16);   // [16][2]: c1 (3 taps), c2 (4 taps)
  const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
  const int q0 = tid & 15, rt0 = tid >> 4;
  // one 16 KB field as 16 pieces of 1 KB: wave w copies pieces w and w + 8
  auto fetch = [&](const float* g, float* l) {
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const int piece = wave + 8 * k;
      pidm_glds_b128_untracked(g + piece * 256 + lane * 4, l + piece * 256);
    }
  };
  if (tid < P) {
    tab4[tid] = fd_taps(ax0, tid, P, 0, P - 1);
    *reinterpret_cast<FdTaps5*>(tab5 + tid * 16) = fd_taps_T(ax0, tid, P, 0, P - 1);
  } else if (tid < P + 16) {
    tabq[2 * (tid - P)] = quad_T(ax1.c1, 4 * (tid - P), P, 3);
    tabq[2 * (tid - P) + 1] = quad_T(ax1.c2, 4 * (tid - P), P, 4);
  }
  int b = blockIdx.x, b_prev = -1, cur = 0;
  [[maybe_unused]] int n_done = 0;
  if (b < B) {
    fetch(f_s, sfs);
    fetch(pred + (size_t)b * 2 * N, spb);
    fetch(pred + (size_t)b * 2 * N + N, sKb);
    fetch(x0 + (size_t)b * 2 * N, tg);
    fetch(x0 + (size_t)b * 2 * N + N, tg + N);
  }
  for (; b < B; b_prev = b, b += gridDim.x, cur ^= 1) {
    const float* sp = spb + cur * N;
    const float* sK = sKb;
    const int rt = rt0, q = q0, j0 = 4 * q;
    const bool lowq = q == 0, highq = q == 15;
    PIDM_WAIT_VMEM_LEAVE(0);     // this sample's copies have landed (leaving the last gradient stores in flight: no gain)
    __syncthreads();
    PIDM_DARCY_STAMP(0);
    if (b_prev >= 0 && tid < 3) {
      double sv = 0.0;
      for (int w = 0; w < NT / 64; ++w) sv += red[tid][w];     // fixed order
      partial[(size_t)b_prev * 4 + tid] = sv;
    }
    if (b + (int)gridDim.x < B) fetch(pred + (size_t)(b + gridDim.x) * 2 * N, spb + (cur ^ 1) * N);

    f32x4 vp[NIT], vK[NIT], t0[NIT], t1[NIT];
    double acc_data = 0.0, acc_r2 = 0.0, acc_rabs = 0.0;
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int n = (rt + RS * it) * P + j0;
      vp[it] = *reinterpret_cast<const f32x4*>(sp + n);
      vK[it] = *reinterpret_cast<const f32x4*>(sK + n);
      t0[it] = *reinterpret_cast<const f32x4*>(tg + n);
      t1[it] = *reinterpret_cast<const f32x4*>(tg + N + n);
#pragma unroll
      for (int m = 0; m < 4; ++m) {
        t0[it][m] = vp[it][m] - t0[it][m];
        t1[it][m] = vK[it][m] - t1[it][m];
        acc_data += (double)(t0[it][m] * t0[it][m]) + (double)(t1[it][m] * t1[it][m]);
      }
    }
    const float gscale = c_res * darcy_inv_var(inv_var, tsteps, b) / ((float)B * (float)N * 3.0f);
    float rkg[NIT][4], ra1[NIT][4], rb1[NIT][4], rd[NIT][4];
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int i = rt + RS * it, n = i * P + j0;
      const FdTaps4 ti = tab4[i];
      float p0[4] = {0.f, 0.f, 0.f, 0.f}, p00[4] = {0.f, 0.f, 0.f, 0.f}, K0[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const f32x4 pa = *reinterpret_cast<const f32x4*>(sp + ti.idx[k] * P + j0);
#pragma unroll
        for (int m = 0; m < 4; ++m) {
          p0[m] = fmaf(ti.w1[k], pa[m], p0[m]);
          p00[m] = fmaf(ti.w2[k], pa[m], p00[m]);
        }
        if (k < 3) {
          const f32x4 ka = *reinterpret_cast<const f32x4*>(sK + ti.idx[k] * P + j0);
#pragma unroll
          for (int m = 0; m < 4; ++m) K0[m] = fmaf(ti.w1[k], ka[m], K0[m]);
        }
      }
      float wp[6], wk[6], p1[4], p11[4], K1[4];
      const float pv[4] = {vp[it][0], vp[it][1], vp[it][2], vp[it][3]}, kv[4] = {vK[it][0], vK[it][1], vK[it][2], vK[it][3]};
      quad_window_regs(pv, lowq, highq, wp);
      quad_window_regs(kv, lowq, highq, wk);
      quad_d1(ax1, wp, lowq, highq, p1);
      quad_d2(ax1, wp, lowq, highq, p11);
      quad_d1(ax1, wk, lowq, highq, K1);
      const f32x4 fs4 = *reinterpret_cast<const f32x4*>(sfs + n);
      const float s0 = (i == 0) ? -1.f : ((i == P - 1) ? 1.f : 0.f);
      float eq[4], bc0[4], bc1[4];
#pragma unroll
      for (int m = 0; m < 4; ++m) {
        const float Kv = wk[m + 1];
        const float vj00 = -Kv * p00[m] - K0[m] * p0[m];
        const float vj11 = -Kv * p11[m] - K1[m] * p1[m];
        eq[m] = vj00 + vj11 - fs4[m];
        const float s1 = (m == 0 && lowq) ? bc1_sign : ((m == 3 && highq) ? -bc1_sign : 0.f);
        bc0[m] = s0 * p0[m];
        bc1[m] = s1 * p1[m];
      }
      f32x4* r = reinterpret_cast<f32x4*>(residual + ((size_t)b * N + n) * 3);
      r[0] = f32x4{eq[0], bc0[0], bc1[0], eq[1]};
      r[1] = f32x4{bc0[1], bc1[1], eq[2], bc0[2]};
      r[2] = f32x4{bc1[2], eq[3], bc0[3], bc1[3]};
      float g[4], gb0[4], gb1[4];
#pragma unroll
      for (int m = 0; m < 4; ++m) {
        // (eq^2 + bc0^2) + bc1^2 in double, as the band kernel: bc0 is an exact zero off the first / last row and bc1 off the
        // first / last column, and adding those zeros changes no bit - they are skipped
        double r2 = (double)(eq[m] * eq[m]), ra = (double)fabsf(eq[m]);
        if (s0 != 0.f) {
          r2 += (double)(bc0[m] * bc0[m]);
          ra += (double)fabsf(bc0[m]);
        }
        if (m == 0 || m == 3) {
          r2 += (double)(bc1[m] * bc1[m]);
          ra += (double)fabsf(bc1[m]);
        }
        acc_r2 += r2;
        acc_rabs += ra;
        g[m] = gscale * eq[m];
        gb0[m] = gscale * bc0[m];
        gb1[m] = gscale * bc1[m];
      }
      f32x4 o_kg, o_a0, o_b0;
#pragma unroll
      for (int m = 0; m < 4; ++m) {
        const float s1 = (m == 0 && lowq) ? bc1_sign : ((m == 3 && highq) ? -bc1_sign : 0.f);
        o_kg[m] = -wk[m + 1] * g[m];
        o_a0[m] = -K0[m] * g[m] + s0 * gb0[m];
        o_b0[m] = -p0[m] * g[m];
        rkg[it][m] = o_kg[m];
        ra1[it][m] = -K1[m] * g[m] + s1 * gb1[m];
        rb1[it][m] = -p1[m] * g[m];
        rd[it][m] = -(p00[m] + p11[m]) * g[m];
        PIDM_OPAQUE_F32(rd[it][m]);     // a ROUNDED product (see darcy_full_kernel)
      }
      *reinterpret_cast<f32x4*>(skg + n) = o_kg;
      *reinterpret_cast<f32x4*>(sa0 + n) = o_a0;
      *reinterpret_cast<f32x4*>(sb0 + n) = o_b0;
    }
    PIDM_DARCY_STAMP(1);
    __syncthreads();
    PIDM_DARCY_STAMP(2);
    // every thread is past its reads of K and of the target: the next sample's may land
    if (b + (int)gridDim.x < B) {
      fetch(pred + (size_t)(b + gridDim.x) * 2 * N + N, sKb);
      fetch(x0 + (size_t)(b + gridDim.x) * 2 * N, tg);
      fetch(x0 + (size_t)(b + gridDim.x) * 2 * N + N, tg + N);
    }

    const float dscale = 2.f * c_data * darcy_p2w(p2w, tsteps, b) / ((float)B * 2.f * (float)N);
    const QuadT tq1 = tabq[2 * q], tq2 = tabq[2 * q + 1];
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int i = rt + RS * it, n = i * P + j0;
      const FdTaps5 ti = *reinterpret_cast<const FdTaps5*>(tab5 + i * 16);
      float g00[4] = {0.f, 0.f, 0.f, 0.f}, ga0[4] = {0.f, 0.f, 0.f, 0.f}, gb0[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int k = 0; k < 5; ++k) {
        const int c0 = ti.idx[k] * P + j0;
        const f32x4 vkg = *reinterpret_cast<const f32x4*>(skg + c0), va = *reinterpret_cast<const f32x4*>(sa0 + c0),
                    vb = *reinterpret_cast<const f32x4*>(sb0 + c0);
#pragma unroll
        for (int m = 0; m < 4; ++m) {
          g00[m] = fmaf(ti.w2[k], vkg[m], g00[m]);
          ga0[m] = fmaf(ti.w1[k], va[m], ga0[m]);
          gb0[m] = fmaf(ti.w1[k], vb[m], gb0[m]);
        }
      }
      float w[6], g11[4], ga1[4], gb1[4];
      quad_window_regs(rkg[it], lowq, highq, w);
      quad_gather(tq2, __shfl(rkg[it][0], 0, 16), __shfl(rkg[it][3], 15, 16), w, g11);
      quad_window_regs(ra1[it], lowq, highq, w);
      quad_gather(tq1, __shfl(ra1[it][0], 0, 16), __shfl(ra1[it][3], 15, 16), w, ga1);
      quad_window_regs(rb1[it], lowq, highq, w);
      quad_gather(tq1, __shfl(rb1[it][0], 0, 16), __shfl(rb1[it][3], 15, 16), w, gb1);
      f32x4 gp, gK;
#pragma unroll
      for (int m = 0; m < 4; ++m) {
        gp[m] = ((g00[m] + g11[m]) + ga0[m]) + ga1[m];
        gK[m] = (rd[it][m] + gb0[m]) + gb1[m];
        gp[m] += dscale * t0[it][m];
        gK[m] += dscale * t1[it][m];
      }
      *reinterpret_cast<f32x4*>(grad_pred + (size_t)b * 2 * N + n) = gp;
      *reinterpret_cast<f32x4*>(grad_pred + (size_t)b * 2 * N + N + n) = gK;
    }
    // loss sums: per wave now, per sample behind the next barrier (the top of the next sample, or the one below)
    double v[3] = {acc_data, acc_r2, acc_rabs};
    for (int qq = 0; qq < 3; ++qq) {
      double x = v[qq];
      for (int off = 32; off > 0; off >>= 1) x += __shfl_down(x, off);
      if (lane == 0) red[qq][wave] = x;
    }
    PIDM_DARCY_STAMP(3);
    ++n_done;
  }
  __syncthreads();
  if (b_prev >= 0 && tid < 3) {
    double sv = 0.0;
    for (int w = 0; w < NT / 64; ++w) sv += red[tid][w];
    partial[(size_t)b_prev * 4 + tid] = sv;
  }
}

// out[0] = loss, out[1] = c_data*data_loss, out[2] = mean|r|, out[3] = 0.  One workgroup; fixed summation order (thread tid owns
// samples tid, tid + 256, ...; bands in order; then a shuffle tree and four wave sums): run-to-run deterministic.
__global__ void __launch_bounds__(256) darcy_loss_finalize(const double* __restrict__ partial, const float* __restrict__ p2w,
                                                           const float* __restrict__ inv_var, const long long* __restrict__ tsteps,
                                                           float c_data, float c_res, int B, int N, int nb, float* __restrict__ out) {
  __shared__ double red[3][4];
  const int tid = threadIdx.x;
  double v[3] = {0.0, 0.0, 0.0};
  for (int b = tid; b < B; b += 256) {
    double d = 0.0, r = 0.0, a = 0.0;
    for (int k = 0; k < nb; ++k) {
      const double* pp = partial + ((size_t)b * nb + k) * 4;
      d += pp[0];
      r += pp[1];
      a += pp[2];
    }
    v[0] += d / (2.0 * N) * (double)darcy_p2w(p2w, tsteps, b);
    v[1] += r * (double)darcy_inv_var(inv_var, tsteps, b);
    v[2] += a;
  }
  for (int q = 0; q < 3; ++q) {
    double x = v[q];
    for (int off = 32; off > 0; off >>= 1) x += __shfl_down(x, off);
    if ((tid & 63) == 0) red[q][tid >> 6] = x;
  }
  __syncthreads();
  if (tid == 0) {
    double data = red[0][0] + red[0][1] + red[0][2] + red[0][3];
    double res = red[1][0] + red[1][1] + red[1][2] + red[1][3];
    const double rabs = red[2][0] + red[2][1] + red[2][2] + red[2][3];
    data = data / B * c_data;
    res = 0.5 * c_res * res / ((double)B * N * 3.0);
    out[0] = (float)(data + res);
    out[1] = (float)data;
    out[2] = (float)(rabs / ((double)B * N * 3.0));
    out[3] = 0.f;
  }
}

static FdAxis make_axis(double inv_h) {
  // textbook 2nd-order-accurate coefficients (what findiff.FinDiff(axis,h,order,acc=2) produces)
  static const double c1[3][3] = {{-1.5, 2.0, -0.5}, {-0.5, 0.0, 0.5}, {1.5, -2.0, 0.5}};
  static const double c2[3][4] = {{2.0, -5.0, 4.0, -1.0}, {1.0, -2.0, 1.0, 0.0}, {2.0, -5.0, 4.0, -1.0}};
  FdAxis a;
  for (int c = 0; c < 3; ++c)
    for (int k = 0; k < 4; ++k) {
      a.c1[c][k] = (k < 3) ? (float)(c1[c][k] * inv_h) : 0.f;
      a.c2[c][k] = (float)(c2[c][k] * inv_h * inv_h);
    }
  return a;
}

// max over ALL entries of the Jacobian d residual[n,k] / d p[m] of one sample (signed max, zeros included) - what the
// reference obtains from a dense vmap(jacfwd) Jacobian of 4096x3 x 4096x2 entries (400 MB per sample,
// src/residuals_darcy.py:217-231).  Analytic from the stencil rows: for pixel (i,j)
//   d eq / d p(k,j) = -K D00[i][k] - K0 D0[i][k]      (k in the column stencil of row i)
//   d eq / d p(i,l) = -K D11[j][l] - K1 D1[j][l]      (l in the row stencil of column j); the two meet at (i,j)
//   d bc0 / d p(k,j) = s0(i) D0[i][k],  d bc1 / d p(i,l) = s1(j) D1[j][l]
__global__ void __launch_bounds__(256) darcy_jacmax_kernel(const float* __restrict__ pred, float bc1_sign, FdAxis ax0, FdAxis ax1,
                                                           float* __restrict__ out, int P) {
  HIP_DYNAMIC_SHARED(float, smem)
  __shared__ float red[4];
  const int N = P * P;
  float* sK = smem;
  const int b = blockIdx.x, tid = threadIdx.x;
  for (int n = tid; n < N; n += 256) sK[n] = pred[(size_t)b * 2 * N + N + n];
  __syncthreads();
  float mx = 0.f;   // the dense Jacobian contains structural zeros
  for (int n = tid; n < N; n += 256) {
    const int i = n / P, j = n - i * P;
    const float Kv = sK[n];
    const float K0 = fd_apply(ax0.c1, sK + j, i, P, P, 3);
    const float K1 = fd_apply(ax1.c1, sK + i * P, j, P, 1, 3);
    const float s0 = (i == 0) ? -1.f : ((i == P - 1) ? 1.f : 0.f);
    const float s1 = (j == 0) ? bc1_sign : ((j == P - 1) ? -bc1_sign : 0.f);
    const int klo = (i == 0) ? 0 : ((i == P - 1) ? P - 4 : i - 1), khi = (i == 0) ? 3 : ((i == P - 1) ? P - 1 : i + 1);
    const int llo = (j == 0) ? 0 : ((j == P - 1) ? P - 4 : j - 1), lhi = (j == 0) ? 3 : ((j == P - 1) ? P - 1 : j + 1);
    const float e1_c = -Kv * fd_coef(ax1.c2, j, j, P, 4) - K1 * fd_coef(ax1.c1, j, j, P, 3);
    for (int k = klo; k <= khi; ++k) {
      const float d0 = fd_coef(ax0.c1, i, k, P, 3);
      float e = -Kv * fd_coef(ax0.c2, i, k, P, 4) - K0 * d0;
      if (k == i) e += e1_c;
      mx = fmaxf(mx, e);
      mx = fmaxf(mx, s0 * d0);
    }
    for (int l = llo; l <= lhi; ++l) {
      const float d1 = fd_coef(ax1.c1, j, l, P, 3);
      if (l != j) mx = fmaxf(mx, -Kv * fd_coef(ax1.c2, j, l, P, 4) - K1 * d1);
      mx = fmaxf(mx, s1 * d1);
    }
  }
  for (int off = 32; off > 0; off >>= 1) mx = fmaxf(mx, __shfl_down(mx, off));
  if ((tid & 63) == 0) red[tid >> 6] = mx;
  __syncthreads();
  if (tid == 0) out[b] = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
}

template <int MODE>
static int launch_darcy(const float* x0, const float* pred, const float* f_s, const float* grad_res, const float* p2w,
                        const float* inv_var, const long long* tsteps, float c_data, float c_res, float inv_h0, float inv_h1,
                        float* residual, float* grad_pred, double* partial, int B, int P, hipStream_t st) {
  if (B <= 0 || P < 5) return fail("darcy: need B>0 and P>=5 (got B=%d P=%d)", B, P);
  const DarcyBands bd = darcy_bands(B, P, MODE == DARCY_RES_ONLY);
  const int rows = darcy_lds_rows(P, bd.R);
  size_t lds = (size_t)(MODE == DARCY_RES_ONLY ? 2 : 8) * rows * P * sizeof(float);
  if (lds > 160 * 1024 - 256) return fail("darcy: P=%d does not fit the 160 KiB LDS", P);
  static bool attr_done = false;
  if (!attr_done) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&darcy_kernel<MODE>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 256);
    attr_done = true;
  }
  const int QP = P >> 2;
  const char* qe = knob("PIDM_DARCY_QUAD");     // 0: the one-pixel-per-thread kernel everywhere (A/B measurements, tests)
  const bool quad = (P & 3) == 0 && P >= 8 && QP <= 256 && (QP & (QP - 1)) == 0 && !(qe && !atoi(qe)) &&
                    ((reinterpret_cast<size_t>(pred) | reinterpret_cast<size_t>(x0) | reinterpret_cast<size_t>(f_s) |
                      reinterpret_cast<size_t>(residual) | reinterpret_cast<size_t>(grad_pred) | reinterpret_cast<size_t>(grad_res)) & 15) == 0;
  if (quad && MODE == DARCY_LOSS && darcy_full_on(B, P) && darcy_stream_on()) {
    static bool attr_s = false;
    const size_t lds_s = (size_t)9 * 64 * 64 * sizeof(float) + 3 * 8 * sizeof(double) + (64 * 12 + 64 * 16 + 16 * 40) * sizeof(float);     // + tap tables
    if (!attr_s) {
      if (hipFuncSetAttribute(reinterpret_cast<const void*>(&darcy_stream_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_s) != hipSuccess) {
        (void)hipGetLastError();
        return fail("darcy_stream_kernel: the device refuses %zu bytes of dynamic LDS", lds_s);
      }
      attr_s = true;
    }
    const char* e = knob("PIDM_DARCY_STREAM_WGS");     // persistent workgroups (one per CU)
    int wgs = e ? atoi(e) : 256;
    if (wgs < 1) wgs = 1;
    if (wgs > B) wgs = B;
    hipLaunchKernelGGL(darcy_stream_kernel, dim3((unsigned)wgs), dim3(512), lds_s, st, x0, pred, f_s, p2w, inv_var, tsteps, c_data, c_res,
                       (inv_h1 < 0.f) ? 1.f : -1.f, make_axis(inv_h0), make_axis(inv_h1), residual, grad_pred, partial, B);
    PIDM_CHECK_LAUNCH("darcy_stream_kernel");
    return 0;
  }
  if (quad && MODE != DARCY_RES_ONLY && darcy_full_on(B, P)) {
    static bool attr_f = false;
    const size_t lds_f = (size_t)5 * 64 * 64 * sizeof(float);
    if (!attr_f) {
      if (hipFuncSetAttribute(reinterpret_cast<const void*>(&darcy_full_kernel<MODE, 512>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_f) != hipSuccess) {
        (void)hipGetLastError();
        return fail("darcy_full_kernel: the device refuses %zu bytes of dynamic LDS", lds_f);
      }
      attr_f = true;
    }
    hipLaunchKernelGGL(HIP_KERNEL_NAME(darcy_full_kernel<MODE, 512>), dim3((unsigned)B), dim3(512), lds_f, st, x0, pred, f_s, grad_res, p2w, inv_var,
                       tsteps, c_data, c_res, (inv_h1 < 0.f) ? 1.f : -1.f, make_axis(inv_h0), make_axis(inv_h1), residual, grad_pred, partial, B);
    PIDM_CHECK_LAUNCH("darcy_full_kernel");
    return 0;
  }
  if (quad) {
    static bool attr_q = false;
    if (!attr_q) {
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&darcy_quad_kernel<MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 256);
      attr_q = true;
    }
    // threads = quads per row x rows of the widest pass (band + 2 halo rows each side) where that is at most 256 (small bands:
    // batch 64 runs 4-row bands with 128 threads, 17.0 -> 13.1 us), else 256: 320-thread blocks for 16-row bands measured 421 us
    // at batch 4096 against 258 with 256 threads (two instead of three workgroups per CU)
    int nt = QP * ((bd.R + 4 < P) ? bd.R + 4 : P);
    nt = (nt + 63) / 64 * 64;
    if (nt > 256) nt = 256;
    if (nt < QP) nt = QP;
    if (nt < 64) nt = 64;
    hipLaunchKernelGGL(HIP_KERNEL_NAME(darcy_quad_kernel<MODE>), dim3((unsigned)B * bd.nb), dim3(nt), lds, st, x0, pred, f_s, grad_res, p2w,
                       inv_var, tsteps, c_data, c_res, (inv_h1 < 0.f) ? 1.f : -1.f, make_axis(inv_h0), make_axis(inv_h1), residual,
                       grad_pred, partial, B, P, bd.R, bd.nb, rows);
    PIDM_CHECK_LAUNCH("darcy_quad_kernel");
    return 0;
  }
  hipLaunchKernelGGL(HIP_KERNEL_NAME(darcy_kernel<MODE>), dim3((unsigned)B * bd.nb), dim3(256), lds, st, x0, pred, f_s, grad_res, p2w,
                     inv_var, tsteps, c_data, c_res, (inv_h1 < 0.f) ? 1.f : -1.f, make_axis(inv_h0), make_axis(inv_h1), residual,
                     grad_pred, partial, B, P, bd.R, bd.nb, rows);
  PIDM_CHECK_LAUNCH("darcy_kernel");
  return 0;
}

static int darcy_loss_impl(const float* x0, const float* x0_pred, const float* f_s, const float* p2w, const float* inv_var,
                           const long long* tsteps, float c_data, float c_residual, float inv_h0, float inv_h1, float* residual,
                           float* grad_x0_pred, float* out_scalars, void* workspace, int B, int P, hipStream_t st) {
  double* partial = reinterpret_cast<double*>(workspace);
  int rc = launch_darcy<DARCY_LOSS>(x0, x0_pred, f_s, nullptr, p2w, inv_var, tsteps, c_data, c_residual, inv_h0, inv_h1, residual,
                                    grad_x0_pred, partial, B, P, st);
  if (rc) return rc;
  hipLaunchKernelGGL(darcy_loss_finalize, dim3(1), dim3(256), 0, st, partial, p2w, inv_var, tsteps, c_data, c_residual, B, P * P,
                     darcy_bands(B, P).nb, out_scalars);
  PIDM_CHECK_LAUNCH("darcy_loss_finalize");
  return 0;
}

}  // namespace pidm

using namespace pidm;

extern "C" int pidm_darcy_residual_fwd(const float* x0, const float* f_s, float inv_h0, float inv_h1, float* residual,
                                       int B, int P, void* stream) {
  return launch_darcy<DARCY_RES_ONLY>(nullptr, x0, f_s, nullptr, nullptr, nullptr, nullptr, 0.f, 0.f, inv_h0, inv_h1, residual,
                                      nullptr, nullptr, B, P, as_stream(stream));
}

extern "C" int pidm_darcy_residual_bwd(const float* x0, const float* grad_res, float inv_h0, float inv_h1,
                                       float* grad_x0, int B, int P, void* stream) {
  return launch_darcy<DARCY_BWD>(nullptr, x0, nullptr, grad_res, nullptr, nullptr, nullptr, 0.f, 0.f, inv_h0, inv_h1, nullptr,
                                 grad_x0, nullptr, B, P, as_stream(stream));
}

extern "C" size_t pidm_darcy_loss_ws(int B, int P) {
  if (B <= 0 || P < 5) return 0;
  return (size_t)B * darcy_bands(B, P).nb * 4 * sizeof(double);
}

extern "C" int pidm_darcy_loss_fwd_bwd(const float* x0, const float* x0_pred, const float* f_s, const float* p2w,
                                       const float* inv_var, float c_data, float c_residual, float inv_h0,
                                       float inv_h1, float* residual, float* grad_x0_pred, float* out_scalars,
                                       void* workspace, int B, int P, void* stream) {
  return darcy_loss_impl(x0, x0_pred, f_s, p2w, inv_var, nullptr, c_data, c_residual, inv_h0, inv_h1, residual, grad_x0_pred,
                         out_scalars, workspace, B, P, as_stream(stream));
}

extern "C" int pidm_darcy_loss_fwd_bwd_t(const float* x0, const float* x0_pred, const float* f_s, const int64_t* t,
                                         const float* p2w_table, const float* var_table, float c_data, float c_residual,
                                         float inv_h0, float inv_h1, float* residual, float* grad_x0_pred, float* out_scalars,
                                         void* workspace, int B, int P, void* stream) {
  if (!t || !p2w_table || !var_table) return fail("darcy_loss_fwd_bwd_t: null time steps / tables");
  return darcy_loss_impl(x0, x0_pred, f_s, p2w_table, var_table, reinterpret_cast<const long long*>(t), c_data, c_residual, inv_h0,
                         inv_h1, residual, grad_x0_pred, out_scalars, workspace, B, P, as_stream(stream));
}

extern "C" int pidm_darcy_jacobian_max(const float* x0, float inv_h0, float inv_h1, float* max_dr_dp, int B, int P, void* stream) {
  if (!x0 || !max_dr_dp) return fail("darcy_jacobian_max: null buffer");
  if (B <= 0 || P < 5) return fail("darcy: need B>0 and P>=5 (got B=%d P=%d)", B, P);
  const size_t lds = (size_t)P * P * sizeof(float);
  if (lds > 160 * 1024 - 256) return fail("darcy: P=%d does not fit the 160 KiB LDS", P);
  static bool attr_done = false;
  if (!attr_done) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&darcy_jacmax_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 256);
    attr_done = true;
  }
  hipLaunchKernelGGL(darcy_jacmax_kernel, dim3(B), dim3(256), lds, as_stream(stream), x0, (inv_h1 < 0.f) ? 1.f : -1.f,
                     make_axis(inv_h0), make_axis(inv_h1), max_dr_dp, P);
  PIDM_CHECK_LAUNCH("darcy_jacmax_kernel");
  return 0;
}

#if PIDM_DARCY_TRACE_BUILD
extern "C" int pidm_debug_darcy_trace(unsigned long long* out256) {
  return hipMemcpyFromSymbol(out256, HIP_SYMBOL(pidm::g_darcy_trace), sizeof(unsigned long long) * 256) == hipSuccess ? 0 : -1;
}
#endif
