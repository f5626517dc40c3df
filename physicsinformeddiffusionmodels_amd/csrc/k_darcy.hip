// Darcy PDE residual, its adjoint, and the fused PIDM loss (forward + d loss / d x0_pred) for gfx950.
//
// Replaces (reference paths): ResidualsDarcy.compute_residual src/residuals_darcy.py:137-183, the 54
// depthwise convolutions + 60 slice scatters of StencilGradientComputation src/grad_utils.py:64-146 and
// the loss algebra of DenoisingDiffusion.model_estimation_loss src/denoising_utils.py:666-692.
//
// One workgroup per sample.  Both fields of the sample (p, K: 2*P*P floats) are staged once in LDS, every
// derivative is a 3/4-tap read of LDS with the one-sided boundary rows selected per pixel, and the adjoint
// is a GATHER over the transposed stencil (no atomics, run-to-run deterministic).  HBM traffic is the
// compulsory 32 KB read + 48 KB residual write + 32 KB gradient write per 64x64 sample.
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

// dynamic LDS: MODE 0: 2 fields; MODE 1/2: 7 fields of P*P floats
template <int MODE>
__global__ void __launch_bounds__(256) darcy_kernel(const float* __restrict__ x0,       // target (MODE 2) [B,2,P,P]
                                                    const float* __restrict__ pred,     // x0_pred [B,2,P,P]
                                                    const float* __restrict__ f_s,      // [P*P]
                                                    const float* __restrict__ grad_res, // MODE 1: [B,P*P,3]
                                                    const float* __restrict__ p2w, const float* __restrict__ inv_var,
                                                    float c_data, float c_res, float bc1_sign, FdAxis ax0, FdAxis ax1,
                                                    float* __restrict__ residual, float* __restrict__ grad_pred,
                                                    double* __restrict__ partial,       // MODE 2: [B][4]
                                                    int B, int P) {
  HIP_DYNAMIC_SHARED(float, smem)
  const int N = P * P;
  float* sp = smem;
  float* sK = smem + N;
  float* sg = smem + 2 * N;
  float* sa0 = smem + 3 * N;
  float* sa1 = smem + 4 * N;
  float* sb0 = smem + 5 * N;
  float* sb1 = smem + 6 * N;
  const int b = blockIdx.x;
  const int tid = threadIdx.x;
  const float* pb = pred + (size_t)b * 2 * N;

  double acc_data = 0.0, acc_r2 = 0.0, acc_rabs = 0.0;
  for (int n = tid; n < N; n += 256) {
    float vp = pb[n], vK = pb[N + n];
    sp[n] = vp;
    sK[n] = vK;
    if (MODE == DARCY_LOSS) {
      float d0 = x0[(size_t)b * 2 * N + n] - vp, d1 = x0[(size_t)b * 2 * N + N + n] - vK;
      acc_data += (double)(d0 * d0) + (double)(d1 * d1);
    }
  }
  __syncthreads();

  const float gscale = (MODE == DARCY_LOSS) ? c_res * inv_var[b] / ((float)B * (float)N * 3.0f) : 0.f;
  for (int n = tid; n < N; n += 256) {
    const int i = n / P, j = n - i * P;
    const float* colp = sp + j;       // walk axis 0 with stride P
    const float* rowp = sp + i * P;   // walk axis 1 with stride 1
    const float* colK = sK + j;
    const float* rowK = sK + i * P;
    float p0 = fd_apply(ax0.c1, colp, i, P, P, 3);
    float p1 = fd_apply(ax1.c1, rowp, j, P, 1, 3);
    float p00 = fd_apply(ax0.c2, colp, i, P, P, 4);
    float p11 = fd_apply(ax1.c2, rowp, j, P, 1, 4);
    float K0 = fd_apply(ax0.c1, colK, i, P, P, 3);
    float K1 = fd_apply(ax1.c1, rowK, j, P, 1, 3);
    float Kv = sK[n];
    // reference op order: vj00 = -K*p00 - K0*p0 ; vj11 = -K*p11 - K1*p1 ; eq = vj00 + vj11 - f_s
    float vj00 = -Kv * p00 - K0 * p0;
    float vj11 = -Kv * p11 - K1 * p1;
    float eq = vj00 + vj11 - f_s[n];
    // boundary rows: bc0 = -p0 (row 0), +p0 (row P-1); bc1 = +p1 (col 0), -p1 (col P-1) when reverse_d1
    // (d1 < 0, bc1_sign = +1), opposite signs otherwise (src/residuals_darcy.py:173-180)
    float s0 = (i == 0) ? -1.f : ((i == P - 1) ? 1.f : 0.f);
    float s1 = (j == 0) ? bc1_sign : ((j == P - 1) ? -bc1_sign : 0.f);
    float bc0 = s0 * p0, bc1 = s1 * p1;
    if (MODE != DARCY_BWD) {
      float* r = residual + ((size_t)b * N + n) * 3;
      r[0] = eq;
      r[1] = bc0;
      r[2] = bc1;
    }
    if (MODE == DARCY_RES_ONLY) continue;
    float g, gb0, gb1;
    if (MODE == DARCY_BWD) {
      const float* gr = grad_res + ((size_t)b * N + n) * 3;
      g = gr[0];
      gb0 = gr[1];
      gb1 = gr[2];
    } else {
      acc_r2 += (double)(eq * eq) + (double)(bc0 * bc0) + (double)(bc1 * bc1);
      acc_rabs += (double)fabsf(eq) + (double)fabsf(bc0) + (double)fabsf(bc1);
      g = gscale * eq;
      gb0 = gscale * bc0;
      gb1 = gscale * bc1;
    }
    sg[n] = g;
    sa0[n] = -K0 * g + s0 * gb0;   // coefficient on p0
    sa1[n] = -K1 * g + s1 * gb1;   // coefficient on p1
    sb0[n] = -p0 * g;              // coefficient on K0
    sb1[n] = -p1 * g;              // coefficient on K1
    // direct dependence of eq on K at the same pixel: -(p00 + p11) * g ; stash in the output buffer
    grad_pred[(size_t)b * 2 * N + N + n] = -(p00 + p11) * g;
  }
  if (MODE == DARCY_RES_ONLY) return;
  __syncthreads();

  const float dscale = (MODE == DARCY_LOSS) ? 2.f * c_data * p2w[b] / ((float)B * 2.f * (float)N) : 0.f;
  for (int n = tid; n < N; n += 256) {
    const int i = n / P, j = n - i * P;
    // grad wrt p: D00^T(-K g) + D11^T(-K g) + D0^T a0 + D1^T a1
    float gp = fd_apply_T(ax0.c2, [&](int ii) { return -sK[ii * P + j] * sg[ii * P + j]; }, i, P, 4);
    gp += fd_apply_T(ax1.c2, [&](int jj) { return -sK[i * P + jj] * sg[i * P + jj]; }, j, P, 4);
    gp += fd_apply_T(ax0.c1, [&](int ii) { return sa0[ii * P + j]; }, i, P, 3);
    gp += fd_apply_T(ax1.c1, [&](int jj) { return sa1[i * P + jj]; }, j, P, 3);
    // grad wrt K: direct + D0^T b0 + D1^T b1
    float gK = grad_pred[(size_t)b * 2 * N + N + n];
    gK += fd_apply_T(ax0.c1, [&](int ii) { return sb0[ii * P + j]; }, i, P, 3);
    gK += fd_apply_T(ax1.c1, [&](int jj) { return sb1[i * P + jj]; }, j, P, 3);
    if (MODE == DARCY_LOSS) {
      gp += dscale * (sp[n] - x0[(size_t)b * 2 * N + n]);
      gK += dscale * (sK[n] - x0[(size_t)b * 2 * N + N + n]);
    }
    grad_pred[(size_t)b * 2 * N + n] = gp;
    grad_pred[(size_t)b * 2 * N + N + n] = gK;
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
      partial[(size_t)b * 4 + tid] = s;
    }
  }
}

// out[0] = loss, out[1] = c_data*data_loss, out[2] = mean|r|, out[3] = 0
__global__ void darcy_loss_finalize(const double* __restrict__ partial, const float* __restrict__ p2w,
                                    const float* __restrict__ inv_var, float c_data, float c_res, int B, int N,
                                    float* __restrict__ out) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  double data = 0.0, res = 0.0, rabs = 0.0;
  for (int b = 0; b < B; ++b) {
    data += partial[b * 4 + 0] / (2.0 * N) * (double)p2w[b];
    res += partial[b * 4 + 1] * (double)inv_var[b];
    rabs += partial[b * 4 + 2];
  }
  data = data / B * c_data;
  res = 0.5 * c_res * res / ((double)B * N * 3.0);
  out[0] = (float)(data + res);
  out[1] = (float)data;
  out[2] = (float)(rabs / ((double)B * N * 3.0));
  out[3] = 0.f;
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
                        const float* inv_var, float c_data, float c_res, float inv_h0, float inv_h1, float* residual,
                        float* grad_pred, double* partial, int B, int P, hipStream_t st) {
  if (B <= 0 || P < 5) return fail("darcy: need B>0 and P>=5 (got B=%d P=%d)", B, P);
  size_t lds = (size_t)(MODE == DARCY_RES_ONLY ? 2 : 7) * P * P * sizeof(float);
  if (lds > 160 * 1024 - 256) return fail("darcy: P=%d does not fit the 160 KiB LDS", P);
  static bool attr_done = false;
  if (!attr_done) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&darcy_kernel<MODE>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 256);
    attr_done = true;
  }
  hipLaunchKernelGGL(HIP_KERNEL_NAME(darcy_kernel<MODE>), dim3(B), dim3(256), lds, st, x0, pred, f_s, grad_res, p2w,
                     inv_var, c_data, c_res, (inv_h1 < 0.f) ? 1.f : -1.f, make_axis(inv_h0), make_axis(inv_h1), residual, grad_pred, partial, B, P);
  PIDM_CHECK_LAUNCH("darcy_kernel");
  return 0;
}

}  // namespace pidm

using namespace pidm;

extern "C" int pidm_darcy_residual_fwd(const float* x0, const float* f_s, float inv_h0, float inv_h1, float* residual,
                                       int B, int P, void* stream) {
  return launch_darcy<DARCY_RES_ONLY>(nullptr, x0, f_s, nullptr, nullptr, nullptr, 0.f, 0.f, inv_h0, inv_h1, residual,
                                      nullptr, nullptr, B, P, as_stream(stream));
}

extern "C" int pidm_darcy_residual_bwd(const float* x0, const float* grad_res, float inv_h0, float inv_h1,
                                       float* grad_x0, int B, int P, void* stream) {
  return launch_darcy<DARCY_BWD>(nullptr, x0, nullptr, grad_res, nullptr, nullptr, 0.f, 0.f, inv_h0, inv_h1, nullptr,
                                 grad_x0, nullptr, B, P, as_stream(stream));
}

extern "C" size_t pidm_darcy_loss_ws(int B, int P) {
  (void)P;
  return (size_t)B * 4 * sizeof(double);
}

extern "C" int pidm_darcy_loss_fwd_bwd(const float* x0, const float* x0_pred, const float* f_s, const float* p2w,
                                       const float* inv_var, float c_data, float c_residual, float inv_h0,
                                       float inv_h1, float* residual, float* grad_x0_pred, float* out_scalars,
                                       void* workspace, int B, int P, void* stream) {
  double* partial = reinterpret_cast<double*>(workspace);
  int rc = launch_darcy<DARCY_LOSS>(x0, x0_pred, f_s, nullptr, p2w, inv_var, c_data, c_residual, inv_h0, inv_h1,
                                    residual, grad_x0_pred, partial, B, P, as_stream(stream));
  if (rc) return rc;
  hipLaunchKernelGGL(darcy_loss_finalize, dim3(1), dim3(64), 0, as_stream(stream), partial, p2w, inv_var, c_data,
                     c_residual, B, P * P, out_scalars);
  PIDM_CHECK_LAUNCH("darcy_loss_finalize");
  return 0;
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
