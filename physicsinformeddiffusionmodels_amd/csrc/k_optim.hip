// Fused global-norm gradient clip + Adam update over FLAT fp32 buffers (main.py:165-166
// `clip_grad_norm_(model.parameters(), 1.)` + `torch.optim.Adam.step()`).  The engine already writes every gradient into
// one flat buffer; with the parameters and both moments flat as well, the whole optimizer is 2 launches that stream
// 7 x 4 B per parameter (HBM-bound: read g,p,m,v - write p,m,v) instead of ~25 multi-tensor launches.
// Deterministic: fixed grid, fixed summation order, no atomics.
#include <hip/hip_runtime.h>
#include <math.h>

#include "pidm_common.h"

namespace pidm {

static constexpr int kNormBlocks = 512;

__global__ void __launch_bounds__(256) sqsum_partial_kernel(const float* __restrict__ g, size_t n, double* __restrict__ partial) {
  __shared__ double red[256];
  const int tid = threadIdx.x;
  const size_t n4 = n / 4;
  const f32x4* g4 = reinterpret_cast<const f32x4*>(g);
  float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;   // 4 independent fp32 chains per thread (<= ~20 terms each at 9M params)
  for (size_t i = (size_t)blockIdx.x * 256 + tid; i < n4; i += (size_t)gridDim.x * 256) {
    const f32x4 v = g4[i];
    a0 += v[0] * v[0]; a1 += v[1] * v[1]; a2 += v[2] * v[2]; a3 += v[3] * v[3];
  }
  double s = ((double)a0 + (double)a1) + ((double)a2 + (double)a3);
  if (blockIdx.x == 0 && tid < (int)(n - 4 * n4)) { const float t = g[4 * n4 + tid]; s += (double)t * t; }
  red[tid] = s;
  __syncthreads();
  for (int w = 128; w > 0; w >>= 1) {
    if (tid < w) red[tid] += red[tid + w];
    __syncthreads();
  }
  if (tid == 0) partial[blockIdx.x] = red[0];
}

// reference EMA.update (src/denoising_utils.py:176): (1. - mu) * param + mu * shadow - two fp32 products and one sum, each
// rounded (no contraction into an fma), so the shadow matches the reference bit for bit
__device__ __forceinline__ float ema_mix(float p, float s, float one_minus_mu, float mu) {
  // HIP's __fmul_rn / __fadd_rn are plain operators and -ffp-contract=fast fuses across them (and across a contract(off)
  // pragma: the back end fuses on its own): the products are made opaque so that each is rounded before the sum
  float a = one_minus_mu * p;
  float b = mu * s;
  PIDM_OPAQUE_F32(a);
  PIDM_OPAQUE_F32(b);
  return a + b;
}

// p, m, v updated in place.  step_size = lr / (1 - beta1^t), bc2_sqrt = sqrt(1 - beta2^t) (computed on the host in double,
// as torch.optim.Adam does); arithmetic order follows torch's foreach implementation:
//   m = lerp(m, g, 1-beta1);  v = v*beta2 + (1-beta2)*g*g;  p += -step_size * m / (sqrt(v)/bc2_sqrt + eps)
// EMA: the parameter average of main.py:178-179 rides along (one extra read + write of the shadow)
template <bool EMA>
__global__ void __launch_bounds__(256) clip_adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                        float* __restrict__ v, size_t n, float step_size, float w1, float beta2, float w2,
                                                        float eps, float bc2_sqrt, float max_norm,
                                                        const double* __restrict__ partial, float* __restrict__ norm_out,
                                                        float* __restrict__ shadow, float ema_w, float ema_mu) {
  __shared__ double red[256];
  __shared__ float coef_s;
  const int tid = threadIdx.x;
  float coef = 1.f;
  if (partial) {
    red[tid] = partial[tid] + partial[tid + 256];
    __syncthreads();
    for (int w = 128; w > 0; w >>= 1) {
      if (tid < w) red[tid] += red[tid + w];
      __syncthreads();
    }
    if (tid == 0) {
      const float nrm = (float)sqrt(red[0]);
      if (blockIdx.x == 0 && norm_out) *norm_out = nrm;
      float c = max_norm / (nrm + 1e-6f);            // torch.nn.utils.clip_grad_norm_: clamp(max_norm/(norm+1e-6), max=1)
      coef_s = (max_norm > 0.f && c < 1.f) ? c : 1.f;
    }
    __syncthreads();
    coef = coef_s;
  }
  const size_t n4 = n / 4;
  f32x4* p4 = reinterpret_cast<f32x4*>(p);
  f32x4* m4 = reinterpret_cast<f32x4*>(m);
  f32x4* v4 = reinterpret_cast<f32x4*>(v);
  const f32x4* g4 = reinterpret_cast<const f32x4*>(g);
  f32x4* s4 = reinterpret_cast<f32x4*>(shadow);
  for (size_t i = (size_t)blockIdx.x * 256 + tid; i < n4; i += (size_t)gridDim.x * 256) {
    f32x4 pp = p4[i], mm = m4[i], vv = v4[i];
    const f32x4 gg = g4[i];
    f32x4 ss;
    if (EMA) ss = s4[i];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float gj = gg[j] * coef;
      mm[j] = mm[j] + w1 * (gj - mm[j]);
      vv[j] = vv[j] * beta2 + w2 * gj * gj;
      const float denom = sqrtf(vv[j]) / bc2_sqrt + eps;
      pp[j] = pp[j] - step_size * (mm[j] / denom);
      if (EMA) ss[j] = ema_mix(pp[j], ss[j], ema_w, ema_mu);
    }
    p4[i] = pp; m4[i] = mm; v4[i] = vv;
    if (EMA) s4[i] = ss;
  }
  if (blockIdx.x == 0 && tid < (int)(n - 4 * n4)) {
    const size_t i = 4 * n4 + tid;
    const float gj = g[i] * coef;
    const float mj = m[i] + w1 * (gj - m[i]);
    const float vj = v[i] * beta2 + w2 * gj * gj;
    m[i] = mj; v[i] = vj;
    const float pn = p[i] - step_size * (mj / (sqrtf(vj) / bc2_sqrt + eps));
    p[i] = pn;
    if (EMA) shadow[i] = ema_mix(pn, shadow[i], ema_w, ema_mu);
  }
}

__global__ void __launch_bounds__(256) ema_update_kernel(float* __restrict__ shadow, const float* __restrict__ p, size_t n, float ema_w,
                                                         float ema_mu) {
  const int tid = threadIdx.x;
  const size_t n4 = n / 4;
  f32x4* s4 = reinterpret_cast<f32x4*>(shadow);
  const f32x4* p4 = reinterpret_cast<const f32x4*>(p);
  for (size_t i = (size_t)blockIdx.x * 256 + tid; i < n4; i += (size_t)gridDim.x * 256) {
    f32x4 ss = s4[i];
    const f32x4 pp = p4[i];
#pragma unroll
    for (int j = 0; j < 4; ++j) ss[j] = ema_mix(pp[j], ss[j], ema_w, ema_mu);
    s4[i] = ss;
  }
  if (blockIdx.x == 0 && tid < (int)(n - 4 * n4)) {
    const size_t i = 4 * n4 + tid;
    shadow[i] = ema_mix(p[i], shadow[i], ema_w, ema_mu);
  }
}

}  // namespace pidm

using namespace pidm;

extern "C" size_t pidm_clip_adam_ws_bytes(void) { return (size_t)kNormBlocks * sizeof(double) + 256; }

static int clip_adam_launch(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, float* ema_shadow, size_t n, double lr,
                            double beta1, double beta2, double eps, long long step, double max_norm, double ema_mu,
                            float* total_norm_out, void* workspace, void* stream) {
  if (!param || !grad || !exp_avg || !exp_avg_sq || !workspace) return fail("clip_adam: null buffer");
  if (step < 1) return fail("clip_adam: step must be >= 1 (got %lld)", step);
  if ((reinterpret_cast<size_t>(param) | reinterpret_cast<size_t>(grad) | reinterpret_cast<size_t>(exp_avg) |
       reinterpret_cast<size_t>(exp_avg_sq) | reinterpret_cast<size_t>(ema_shadow)) & 15)
    return fail("clip_adam: buffers must be 16-byte aligned");
  hipStream_t st = as_stream(stream);
  double* partial = reinterpret_cast<double*>((reinterpret_cast<size_t>(workspace) + 255) & ~(size_t)255);
  const bool need_norm = max_norm > 0.0 || total_norm_out;
  if (need_norm) {
    hipLaunchKernelGGL(sqsum_partial_kernel, dim3(kNormBlocks), dim3(256), 0, st, grad, n, partial);
    PIDM_CHECK_LAUNCH("sqsum_partial_kernel");
  }
  const double bc1 = 1.0 - pow(beta1, (double)step), bc2 = 1.0 - pow(beta2, (double)step);
  const float step_size = (float)(lr / bc1), bc2_sqrt = (float)sqrt(bc2);
  size_t blocks = (n / 4 + 255) / 256;
  if (blocks > 2048) blocks = 2048;
  if (blocks < 1) blocks = 1;
  // (1. - mu) is formed in double and rounded once, as python does before torch multiplies the fp32 tensor by the scalar
  const float ema_w = (float)(1.0 - ema_mu), mu_f = (float)ema_mu;
  if (ema_shadow) {
    hipLaunchKernelGGL(HIP_KERNEL_NAME(clip_adam_kernel<true>), dim3((unsigned)blocks), dim3(256), 0, st, param, grad, exp_avg, exp_avg_sq, n,
                       step_size, (float)(1.0 - beta1), (float)beta2, (float)(1.0 - beta2), (float)eps, bc2_sqrt, (float)max_norm,
                       need_norm ? partial : nullptr, total_norm_out, ema_shadow, ema_w, mu_f);
  } else {
    hipLaunchKernelGGL(HIP_KERNEL_NAME(clip_adam_kernel<false>), dim3((unsigned)blocks), dim3(256), 0, st, param, grad, exp_avg, exp_avg_sq, n,
                       step_size, (float)(1.0 - beta1), (float)beta2, (float)(1.0 - beta2), (float)eps, bc2_sqrt, (float)max_norm,
                       need_norm ? partial : nullptr, total_norm_out, nullptr, 0.f, 0.f);
  }
  PIDM_CHECK_LAUNCH("clip_adam_kernel");
  return 0;
}

extern "C" int pidm_clip_adam_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, size_t n, double lr,
                                   double beta1, double beta2, double eps, long long step, double max_norm, float* total_norm_out,
                                   void* workspace, void* stream) {
  return clip_adam_launch(param, grad, exp_avg, exp_avg_sq, nullptr, n, lr, beta1, beta2, eps, step, max_norm, 0.0, total_norm_out,
                          workspace, stream);
}

extern "C" int pidm_clip_adam_ema_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, float* ema_shadow, size_t n,
                                       double lr, double beta1, double beta2, double eps, long long step, double max_norm,
                                       double ema_mu, float* total_norm_out, void* workspace, void* stream) {
  if (!ema_shadow) return fail("clip_adam_ema: null shadow buffer");
  return clip_adam_launch(param, grad, exp_avg, exp_avg_sq, ema_shadow, n, lr, beta1, beta2, eps, step, max_norm, ema_mu, total_norm_out,
                          workspace, stream);
}

extern "C" int pidm_ema_update(float* ema_shadow, const float* param, size_t n, double ema_mu, void* stream) {
  if (!ema_shadow || !param) return fail("ema_update: null buffer");
  if ((reinterpret_cast<size_t>(ema_shadow) | reinterpret_cast<size_t>(param)) & 15) return fail("ema_update: buffers must be 16-byte aligned");
  size_t blocks = (n / 4 + 255) / 256;
  if (blocks > 2048) blocks = 2048;
  if (blocks < 1) blocks = 1;
  hipLaunchKernelGGL(ema_update_kernel, dim3((unsigned)blocks), dim3(256), 0, as_stream(stream), ema_shadow, param, n,
                     (float)(1.0 - ema_mu), (float)ema_mu);
  PIDM_CHECK_LAUNCH("ema_update_kernel");
  return 0;
}
