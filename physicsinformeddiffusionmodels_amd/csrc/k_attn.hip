// Linear attention (every UNet level) and the 64-token softmax attention of the bottleneck, fwd + bwd.
//
// Replaces SpatialLinearAttention.forward src/unet_model.py:281-299 and Attention.forward :341-367 (+ their
// autograd).  qkv comes from the 1x1 to_qkv GEMM as [B][N][3*HD] channels-last (q | k | v, head-major), HD =
// heads*32.  The contractions run on the fp32 matrix cores (32x32x2) straight from global memory: a
// 32-wide head slice of one pixel is 128 contiguous bytes, so every lane-half fetch is a coalesced line.
//   context[d][e] = 1/N sum_n softmax_n(k)[n][d] v[n][e]        (K = N pixels, split over 4 waves)
//   out[n][e]     = sum_d context[d][e] * softmax_d(q)[n][d]*scale   (K = 32)
// q/k/v are never re-materialised after softmax: the softmax is applied on the fly from per-column (k) and
// per-pixel (q) statistics.
//
// Kernel index
//   la_kstats / la_kstats_final        k-softmax column statistics over the pixels (two stages)
//   la_nreduce<MODE> / _final          32x32 pixel reductions per (image, head): context (MODE 0), dctx + rowdot (MODE 1)
//   la_out<ALIGNED>                    out = context^T softmax_d(q)                          (separate-projection form)
//   la_bwd_pix / la_bwd_pix_mfma       dq, dk, dv from dA                                    (separate-projection form)
//   la_out_proj<CT>, la_wt             y = to_out(out) + bias + residual, out never stored   (fused form, forward)
//   la_g<NT> / la_g_final<NT>          G = qs^T dY -> dctx = G W, rowdot, dW_out share       (fused form, backward)
//   la_bwd_dq, la_bwd_dkdv             dq from (q, dY); dk, dv from (k, v); transposed MFMA chaining, full-line I/O
//   mid_attn<BWD>                      bottleneck softmax attention (<= 64 tokens)
// Launchers: launch_la_forward / _backward (separate), launch_la_forward_fused / _backward_fused (+ la_fused_ok, la_fused_pays).
#include "pidm_launch.h"

namespace pidm {

static const int DH = 32;  // dim_head (the 32x32 MFMA tile)

// softmax exponentials: arguments are <= 0 after the max subtraction, so the hardware exp2 of x*log2(e) (2 instructions, relative
// error <= |x| * 1e-7 on a term of size e^x) replaces the ~15-instruction libm expf; the per-pixel kernels issue 16-48 of these per
// lane and head, which was on a par with their matrix-core time.
__device__ __forceinline__ float fexp(float x) { return __expf(x); }

// ---- k softmax statistics over pixels: kstat[b][j] = (max_n, 1/sum_n exp(k - max)) -----------------
// stage 1: online (max, sum) over one segment of pixels per block; stage 2 merges the segments.
// kstat != null (one segment: every level of the Darcy and mechanics models): the statistics are final - (max, 1 / sum) go straight
// to kstat and la_kstats_final_kernel is not launched (the same values: it would multiply the sum by exp(0) = 1)
__global__ void __launch_bounds__(256) la_kstats_kernel(const float* __restrict__ qkv, int N, int HD, int nseg,
                                                        float* __restrict__ kpart, float* __restrict__ kstat) {
  __shared__ float sm[4][64], ssum[4][64];
  const int b = blockIdx.y, seg = blockIdx.z, tid = threadIdx.x, cl = tid & 63, rl = tid >> 6;
  const int j = blockIdx.x * 64 + cl;
  const int per = (N + nseg - 1) / nseg;
  const int n_lo = seg * per, n_hi = (n_lo + per < N) ? n_lo + per : N;
  float m = -3.0e38f, s = 0.f;
  if (j < HD) {
    const float* base = qkv + (size_t)b * N * 3 * HD + HD + j;
    // 8 independent loads in flight per thread, then ONE rescale per group (max of the group first): the dependent
    // load -> exp -> exp chain of a per-element online softmax left this streaming kernel latency bound
    int n = n_lo + rl;
    for (; n + 28 < n_hi; n += 32) {
      float v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = base[(size_t)(n + 4 * u) * 3 * HD];
      float gm = v[0];
#pragma unroll
      for (int u = 1; u < 8; ++u) gm = fmaxf(gm, v[u]);
      const float mn = fmaxf(m, gm);
      float gs = 0.f;
#pragma unroll
      for (int u = 0; u < 8; ++u) gs += fexp(v[u] - mn);
      s = s * fexp(m - mn) + gs;
      m = mn;
    }
    for (; n < n_hi; n += 4) {
      const float v = base[(size_t)n * 3 * HD];
      const float mn = fmaxf(m, v);
      s = s * fexp(m - mn) + fexp(v - mn);
      m = mn;
    }
  }
  sm[rl][cl] = m;
  ssum[rl][cl] = s;
  __syncthreads();
  if (rl == 0 && j < HD) {
    float M = fmaxf(fmaxf(sm[0][cl], sm[1][cl]), fmaxf(sm[2][cl], sm[3][cl]));
    float S = 0.f;
    for (int r = 0; r < 4; ++r) S += ssum[r][cl] * fexp(sm[r][cl] - M);
    if (kstat) {
      kstat[((size_t)b * HD + j) * 2] = M;
      kstat[((size_t)b * HD + j) * 2 + 1] = 1.f / S;
    } else {
      kpart[(((size_t)b * nseg + seg) * HD + j) * 2] = M;
      kpart[(((size_t)b * nseg + seg) * HD + j) * 2 + 1] = S;
    }
  }
}
__global__ void la_kstats_final_kernel(const float* __restrict__ kpart, int B, int HD, int nseg, float* __restrict__ kstat) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= B * HD) return;
  const int b = i / HD, j = i % HD;
  float M = -3.0e38f;
  for (int sg = 0; sg < nseg; ++sg) M = fmaxf(M, kpart[(((size_t)b * nseg + sg) * HD + j) * 2]);
  float S = 0.f;
  for (int sg = 0; sg < nseg; ++sg) {
    const float* p = kpart + (((size_t)b * nseg + sg) * HD + j) * 2;
    S += p[1] * fexp(p[0] - M);
  }
  kstat[(size_t)i * 2] = M;
  kstat[(size_t)i * 2 + 1] = 1.f / S;
}

// ---- D[b][h][i][j] = alpha * sum_n A(n,i) * Bm(n,j)  (32x32 per (b,h), K = N) ----------------------
// MODE 0 (forward context): A = softmax_n(k)[n][h*32+i] from kstat,  Bm = v[n][h*32+j]
// MODE 1 (backward dctx):   A = softmax_d(q)[n][h*32+i]*scale from qstat, Bm = dA[n][h*32+j]; also writes
//                           rowdot[b][h][i] = sum_j D[i][j] * ctx[b][h][i][j]
template <int MODE>
__global__ void __launch_bounds__(256) la_nreduce_kernel(const float* __restrict__ qkv, const float* __restrict__ stat,
                                                         const float* __restrict__ dA, const float* __restrict__ ctx,
                                                         float* __restrict__ Dpart, int N, int heads, float scale,
                                                         float* __restrict__ Dfinal, float* __restrict__ rowdot, float alpha) {
  __shared__ float red[4][1024];
  const int bh = blockIdx.x, b = bh / heads, h = bh % heads;
  const int ns = blockIdx.y, NS = gridDim.y;
  const int HD = heads * DH;
  const int tid = threadIdx.x, lane = tid & 63, half = lane >> 5, l31 = lane & 31;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // scalar pixel ranges / row pointers (see la_g_kernel)
  f32x16 acc;
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  const float* q0 = qkv + (size_t)b * N * 3 * HD;
  float cm = 0.f, cis = 0.f;
  if (MODE == 0) {
    cm = stat[((size_t)b * HD + h * DH + l31) * 2];
    cis = stat[((size_t)b * HD + h * DH + l31) * 2 + 1];
  }
  const int per_blk = (N + NS - 1) / NS;
  const int blk_lo = ns * per_blk, blk_hi = (blk_lo + per_blk < N) ? blk_lo + per_blk : N;
  const int per = (blk_hi - blk_lo + 3) / 4;
  const int n_lo = blk_lo + wave * per, n_hi = (n_lo + per < blk_hi) ? n_lo + per : blk_hi;
  int n0 = n_lo;
  {
    // UNR independent pixel pairs in flight per wave (wave-uniform bounds): one load -> exp -> MFMA round trip per pair
    // left this stream latency bound
    constexpr int UNR = 4;
    const int la = half * 3 * HD + (MODE == 0 ? HD : 0) + h * DH + l31;       // A source: k (MODE 0) or q (MODE 1)
    const int lb = (MODE == 0) ? half * 3 * HD + 2 * HD + h * DH + l31 : half * HD + h * DH + l31;
    const int ls = (half * heads + h) * 2;
    for (; n0 + 2 * UNR <= n_hi; n0 += 2 * UNR) {
      const float* qrow = q0 + (size_t)n0 * 3 * HD;
      const float* brow = (MODE == 0) ? qrow : dA + ((size_t)b * N + n0) * HD;
      const float* srow = (MODE == 0) ? nullptr : stat + ((size_t)b * N + n0) * heads * 2;
      float av[UNR], bvv[UNR];
      float2 qs[UNR];
#pragma unroll
      for (int u = 0; u < UNR; ++u) {
        av[u] = (qrow + (size_t)(2 * u) * 3 * HD)[la];
        bvv[u] = (brow + (size_t)(2 * u) * (MODE == 0 ? 3 * HD : HD))[lb];
        if (MODE == 1) qs[u] = *reinterpret_cast<const float2*>(srow + (size_t)(2 * u) * heads * 2 + ls);
      }
#pragma unroll
      for (int u = 0; u < UNR; ++u) {
        const float a = (MODE == 0) ? fexp(av[u] - cm) * cis : fexp(av[u] - qs[u].x) * qs[u].y * scale;
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, bvv[u], acc, 0, 0, 0);
      }
    }
  }
  for (; n0 < n_hi; n0 += 2) {
    const int n = n0 + half;
    float a = 0.f, bv = 0.f;
    if (n < n_hi) {
      const float* row = q0 + (size_t)n * 3 * HD;
      if (MODE == 0) {
        a = fexp(row[HD + h * DH + l31] - cm) * cis;
        bv = row[2 * HD + h * DH + l31];
      } else {
        const float* qs = stat + (((size_t)b * N + n) * heads + h) * 2;
        a = fexp(row[h * DH + l31] - qs[0]) * qs[1] * scale;
        bv = dA[((size_t)b * N + n) * HD + h * DH + l31];
      }
    }
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, bv, acc, 0, 0, 0);
  }
  for (int r = 0; r < 16; ++r) {
    const int row = (r & 3) + 8 * (r >> 2) + 4 * half;
    red[wave][row * 32 + l31] = acc[r];
  }
  __syncthreads();
  if (Dfinal) {
    // one pixel range per (image, head) (gridDim.y == 1: images of <= 512 pixels): this block's sum IS the result - what
    // la_nreduce_final_kernel would make of the single partial (alpha * (0 + sum); MODE 1: the row dots in the same order)
    float v[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int e = tid + 256 * k;
      v[k] = ((red[0][e] + red[1][e]) + (red[2][e] + red[3][e])) * alpha;
      Dfinal[(size_t)bh * 1024 + e] = v[k];
    }
    if (MODE == 1) {
      __syncthreads();
#pragma unroll
      for (int k = 0; k < 4; ++k) red[0][tid + 256 * k] = v[k] * ctx[(size_t)bh * 1024 + tid + 256 * k];
      __syncthreads();
      if (tid < 32) {
        float sacc = 0.f;
        for (int j = 0; j < 32; ++j) sacc += red[0][tid * 32 + j];
        rowdot[(size_t)bh * 32 + tid] = sacc;
      }
    }
    return;
  }
  float* out = Dpart + ((size_t)bh * NS + ns) * 1024;
  for (int e = tid; e < 1024; e += 256) out[e] = (red[0][e] + red[1][e]) + (red[2][e] + red[3][e]);
}

// D[bh] = alpha * sum_ns Dpart[bh][ns]; MODE 1 additionally rowdot[bh][i] = sum_j D[i][j] * ctx[bh][i][j]
template <int MODE>
__global__ void __launch_bounds__(256) la_nreduce_final_kernel(const float* __restrict__ Dpart, const float* __restrict__ ctx,
                                                               float* __restrict__ D, float* __restrict__ rowdot, int NS,
                                                               float alpha) {
  __shared__ float prod[1024];
  const int bh = blockIdx.x, tid = threadIdx.x;
  for (int e = tid; e < 1024; e += 256) {
    float v = 0.f;
    for (int k = 0; k < NS; ++k) v += Dpart[((size_t)bh * NS + k) * 1024 + e];
    v *= alpha;
    D[(size_t)bh * 1024 + e] = v;
    if (MODE == 1) prod[e] = v * ctx[(size_t)bh * 1024 + e];
  }
  if (MODE == 1) {
    __syncthreads();
    if (tid < 32) {
      float sacc = 0.f;
      for (int j = 0; j < 32; ++j) sacc += prod[tid * 32 + j];
      rowdot[(size_t)bh * 32 + tid] = sacc;
    }
  }
}

// ---- out[n][h*32+e] = sum_d ctx[d][e] * softmax_d(q[n][h*32+:])[d] * scale ; qstat = (max, 1/sum) -------
// one wave = 32 pixels, loops over heads.  lane (pixel l31, half) owns d in [16*half, 16*half+16).
// ALIGNED (N % 32 == 0): the wave's 32 pixels share one image -> matrix-core path.  A compile-time switch: as a runtime
// branch the register-hungry fallback (a 32-float row per lane) set the allocation of the hot path too (236 VGPRs).
template <bool ALIGNED>
__global__ void __launch_bounds__(256) la_out_kernel(const float* __restrict__ qkv, const float* __restrict__ ctx,
                                                     float* __restrict__ out, float* __restrict__ qstat, int N, int heads,
                                                     size_t npix, float scale) {
  const int HD = heads * DH;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, half = lane >> 5, l31 = lane & 31;
  const size_t p = ((size_t)blockIdx.x * 4 + wave) * 32 + l31;   // global pixel index (b*N + n)
  const bool valid = p < npix;
  const size_t b = valid ? p / N : 0;
  for (int h = blockIdx.y; h < heads; h += gridDim.y) {     // gridDim.y = heads where the pixel grid alone leaves CUs idle
    float qv[16];
    float m = -3.0e38f;
    if (valid) {
      const float* qp = qkv + p * 3 * HD + h * DH + 16 * half;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const float4 v = *reinterpret_cast<const float4*>(qp + 4 * k);
        qv[4 * k] = v.x; qv[4 * k + 1] = v.y; qv[4 * k + 2] = v.z; qv[4 * k + 3] = v.w;
      }
#pragma unroll
      for (int k = 0; k < 16; ++k) m = fmaxf(m, qv[k]);
    } else {
#pragma unroll
      for (int k = 0; k < 16; ++k) qv[k] = 0.f;
      m = 0.f;
    }
    m = fmaxf(m, pidm_other_half(m));
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < 16; ++k) {
      qv[k] = fexp(qv[k] - m);
      s += qv[k];
    }
    s += pidm_other_half(s);
    const float inv = 1.f / s;
    if (valid && half == 0) {
      qstat[(p * heads + h) * 2] = m;
      qstat[(p * heads + h) * 2 + 1] = inv;
    }
    const float* cb = ctx + (b * heads + h) * 1024;   // [d][e]; rows of *other* batches are never mixed:
    // a wave's 32 pixels may straddle two images only when N < 32; handle by per-lane b below
    f32x16 acc;
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    if constexpr (ALIGNED) {
#pragma unroll
      for (int s2 = 0; s2 < 16; ++s2) {
        const float a = qv[s2] * inv * scale;
        const float bv = cb[(16 * half + s2) * 32 + l31];
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, bv, acc, 0, 0, 0);
      }
      // all 32 rows belong to image b of lane... rows = pixels of this wave: same image since N % 32 == 0
      const size_t pw = ((size_t)blockIdx.x * 4 + wave) * 32;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = (r & 3) + 8 * (r >> 2) + 4 * half;
        if (pw + row < npix) out[(pw + row) * HD + h * DH + l31] = acc[r];
      }
    } else {
      // small images (N < 32 or ragged): the B operand differs per row -> plain FMA path, lane = pixel
      // each lane-half computes e in [16*half, 16*half+16) for its own pixel from the full q row
      float qfull[32];
#pragma unroll
      for (int k = 0; k < 16; ++k) {
        const float mine = qv[k] * inv * scale;
        const float other = pidm_other_half(mine);
        qfull[k] = half ? other : mine;        // static indices keep the array in registers
        qfull[16 + k] = half ? mine : other;
      }
      if (valid) {
        for (int e = 16 * half; e < 16 * half + 16; ++e) {
          float o = 0.f;
#pragma unroll
          for (int d = 0; d < 32; ++d) o = fmaf(cb[d * 32 + e], qfull[d], o);
          out[p * HD + h * DH + e] = o;
        }
      }
    }
  }
}

// ---- backward, per pixel: dq, dk, dv into dqkv[B][N][3*HD] ------------------------------------------------
// dq' = dA ctx^T ; dq = scale*qs*(dq' - sum_d qs dq')        qs = softmax_d(q)
// dP  = (1/N) v dctx^T ; dk = P*(dP - rowdot)                 P  = softmax_n(k)
// dv  = (1/N) P dctx
// generic (non-MFMA) formulation: one thread per (pixel, head); the three 32x32 mat-vecs read ctx/dctx from LDS.
__global__ void __launch_bounds__(256) la_bwd_pix_kernel(const float* __restrict__ qkv, const float* __restrict__ kstat,
                                                         const float* __restrict__ qstat, const float* __restrict__ ctx,
                                                         const float* __restrict__ dctx, const float* __restrict__ rowdot,
                                                         const float* __restrict__ dA, float* __restrict__ dqkv, int N,
                                                         int heads, float scale) {
  // block = (b, h, 256-pixel slab)
  __shared__ float sc[32][33], sd[32][33], srd[32], skm[32], skis[32];
  const int HD = heads * DH;
  const int bh = blockIdx.y, b = bh / heads, h = bh % heads;
  const int tid = threadIdx.x;
  for (int e = tid; e < 1024; e += 256) {
    sc[e >> 5][e & 31] = ctx[(size_t)bh * 1024 + e];
    sd[e >> 5][e & 31] = dctx[(size_t)bh * 1024 + e];
  }
  if (tid < 32) {
    srd[tid] = rowdot[(size_t)bh * 32 + tid];
    skm[tid] = kstat[((size_t)b * HD + h * DH + tid) * 2];
    skis[tid] = kstat[((size_t)b * HD + h * DH + tid) * 2 + 1];
  }
  __syncthreads();
  const int n = blockIdx.x * 256 + tid;
  if (n >= N) return;
  const size_t p = (size_t)b * N + n;
  const float* row = qkv + p * 3 * HD + h * DH;
  float q[32], da[32], tmp[32];
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    const float4 v = *reinterpret_cast<const float4*>(row + 4 * k);
    q[4 * k] = v.x; q[4 * k + 1] = v.y; q[4 * k + 2] = v.z; q[4 * k + 3] = v.w;
    const float4 w = *reinterpret_cast<const float4*>(dA + p * HD + h * DH + 4 * k);
    da[4 * k] = w.x; da[4 * k + 1] = w.y; da[4 * k + 2] = w.z; da[4 * k + 3] = w.w;
  }
  const float qm = qstat[(p * heads + h) * 2], qis = qstat[(p * heads + h) * 2 + 1];
  // dq
  float dot = 0.f;
#pragma unroll
  for (int d = 0; d < 32; ++d) {
    float dqp = 0.f;
#pragma unroll
    for (int e = 0; e < 32; ++e) dqp = fmaf(sc[d][e], da[e], dqp);
    const float qs = fexp(q[d] - qm) * qis;
    q[d] = qs;
    tmp[d] = dqp;
    dot = fmaf(qs, dqp, dot);
  }
  float* orow = dqkv + p * 3 * HD + h * DH;
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    float4 o;
    o.x = scale * q[4 * k] * (tmp[4 * k] - dot);
    o.y = scale * q[4 * k + 1] * (tmp[4 * k + 1] - dot);
    o.z = scale * q[4 * k + 2] * (tmp[4 * k + 2] - dot);
    o.w = scale * q[4 * k + 3] * (tmp[4 * k + 3] - dot);
    *reinterpret_cast<float4*>(orow + 4 * k) = o;
  }
  // dk: needs v and P
  float v[32], P[32];
  const float invN = 1.f / (float)N;
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    const float4 kk = *reinterpret_cast<const float4*>(row + HD + 4 * k);
    const float4 vv = *reinterpret_cast<const float4*>(row + 2 * HD + 4 * k);
    P[4 * k] = fexp(kk.x - skm[4 * k]) * skis[4 * k];
    P[4 * k + 1] = fexp(kk.y - skm[4 * k + 1]) * skis[4 * k + 1];
    P[4 * k + 2] = fexp(kk.z - skm[4 * k + 2]) * skis[4 * k + 2];
    P[4 * k + 3] = fexp(kk.w - skm[4 * k + 3]) * skis[4 * k + 3];
    v[4 * k] = vv.x; v[4 * k + 1] = vv.y; v[4 * k + 2] = vv.z; v[4 * k + 3] = vv.w;
  }
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    float o[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int d = 4 * k + u;
      float dP = 0.f;
#pragma unroll
      for (int e = 0; e < 32; ++e) dP = fmaf(sd[d][e], v[e], dP);
      o[u] = P[d] * (dP * invN - srd[d]);
    }
    *reinterpret_cast<float4*>(orow + HD + 4 * k) = make_float4(o[0], o[1], o[2], o[3]);
  }
  // dv
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    float o[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int e = 4 * k + u;
      float s = 0.f;
#pragma unroll
      for (int d = 0; d < 32; ++d) s = fmaf(P[d], sd[d][e], s);
      o[u] = s * invN;
    }
    *reinterpret_cast<float4*>(orow + 2 * HD + 4 * k) = make_float4(o[0], o[1], o[2], o[3]);
  }
}

// ---- backward per pixel on the matrix cores (N % 128 == 0): one wave = 32 pixels, loops over heads ----------
// three 32x32x32 GEMMs per (32 pixels, head): dq' = dA ctx^T, dP = v dctx^T, dv = P dctx; the softmax Jacobians are
// applied in the MFMA C layout (row = pixel, col = channel), where the d-reduction of the q-softmax is a 32-lane
// butterfly and the k-softmax term needs only per-column constants.
__global__ void __launch_bounds__(256) la_bwd_pix_mfma_kernel(const float* __restrict__ qkv, const float* __restrict__ kstat,
                                                              const float* __restrict__ qstat, const float* __restrict__ ctx,
                                                              const float* __restrict__ dctx, const float* __restrict__ rowdot,
                                                              const float* __restrict__ dA, float* __restrict__ dqkv, int N,
                                                              int heads, float scale, int ppb) {
  // ppb = pixels per block = 32 x its waves (128 / 64 / 32, chosen by the launcher), N % ppb == 0: one image per block
  __shared__ float sc[32][33], sd[32][33], srd[32], skm[32], skis[32];
  const int HD = heads * DH;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, half = lane >> 5, l31 = lane & 31;
  const size_t pblk = (size_t)blockIdx.x * ppb;          // first pixel (b*N + n) of the block; one image per block
  const int b = (int)(pblk / N);
  const size_t pw = pblk + wave * 32;                     // first pixel of this wave
  const size_t pa = pw + l31;                             // this lane's A-layout pixel
  const float invN = 1.f / (float)N;
  // one head per block (blockIdx.y): with the heads walked one after the other inside a block the launch was a serial chain of
  // 8 x (stage, 48 MFMAs, epilogue) on a grid that left most of the chip idle at batch 64 - ~70 us for the 16x16 AND the 8x8 level
  {
    const int h = blockIdx.y;
    const int bh = b * heads + h;
    for (int e = tid; e < 1024; e += 2 * ppb) {
      sc[e >> 5][e & 31] = ctx[(size_t)bh * 1024 + e];
      sd[e >> 5][e & 31] = dctx[(size_t)bh * 1024 + e];
    }
    if (tid < 32) {
      srd[tid] = rowdot[(size_t)bh * 32 + tid];
      skm[tid] = kstat[((size_t)b * HD + h * DH + tid) * 2];
      skis[tid] = kstat[((size_t)b * HD + h * DH + tid) * 2 + 1];
    }
    __syncthreads();
    // A-layout operands: 16 contiguous channels [16*half, 16*half+16) of this lane's pixel
    const float* rowA = qkv + pa * 3 * HD + h * DH + 16 * half;
    f32x4 a_dA[4], a_v[4], a_k[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      a_dA[j] = *reinterpret_cast<const f32x4*>(dA + pa * HD + h * DH + 16 * half + 4 * j);
      a_k[j] = *reinterpret_cast<const f32x4*>(rowA + HD + 4 * j);
      a_v[j] = *reinterpret_cast<const f32x4*>(rowA + 2 * HD + 4 * j);
    }
    f32x16 acc1, acc2, acc3;
    for (int r = 0; r < 16; ++r) { acc1[r] = 0.f; acc2[r] = 0.f; acc3[r] = 0.f; }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const int s = 4 * j + c, kk = 16 * half + s;
        acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a_dA[j][c], sc[l31][kk], acc1, 0, 0, 0);   // dq'[n][d] += dA[n][e] ctx[d][e]
        acc2 = __builtin_amdgcn_mfma_f32_32x32x2f32(a_v[j][c], sd[l31][kk], acc2, 0, 0, 0);    // dP [n][d] += v[n][e] dctx[d][e]
        const float pA = fexp(a_k[j][c] - skm[kk]) * skis[kk];
        acc3 = __builtin_amdgcn_mfma_f32_32x32x2f32(pA, sd[kk][l31], acc3, 0, 0, 0);           // dv [n][e] += P[n][d] dctx[d][e]
      }
    }
    // epilogues in the C layout: row r -> pixel pw + prow, column l31 -> channel
    const float km = skm[l31], kis = skis[l31], rd = srd[l31];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int prow = (r & 3) + 8 * (r >> 2) + 4 * half;
      const size_t p = pw + prow;
      const float* rowC = qkv + p * 3 * HD + h * DH + l31;
      const float qm = qstat[(p * heads + h) * 2], qis = qstat[(p * heads + h) * 2 + 1];
      const float qs = fexp(rowC[0] - qm) * qis;
      float dot = qs * acc1[r];
#pragma unroll
      for (int off = 1; off < 32; off <<= 1) dot += __shfl_xor(dot, off);
      float* o = dqkv + p * 3 * HD + h * DH + l31;
      o[0] = scale * qs * (acc1[r] - dot);
      const float P = fexp(rowC[HD] - km) * kis;
      o[HD] = P * (acc2[r] * invN - rd);
      o[2 * HD] = acc3[r] * invN;
    }
  }
}

// ---------------------------------------------------------------------------------------------------
// Backward fused with the to_out 1x1 convolution (weights W[Cout][HD], output gradient dY[B][N][Cout]):
// dA = dY W is never written.  With G[b,h][d][c] = sum_n qs[n][(h,d)] dY[n][c]  (qs = softmax_d(q)*scale):
//   dctx[b,h][d][e]   = sum_n qs[n][d] dA[n][e]         = sum_c G[d][c] W[c][(h,e)]
//   dW[c][(h,e)]      = sum_n dY[n][c] out[n][(h,e)]    = sum_b sum_d G[b,h][d][c] ctx[b,h][d][e]
// so one n-reduction (reads q and the Cout-channel dY instead of q and the HD-channel dA) replaces the conv dgrad, the
// dctx reduction and the conv wgrad (which read the materialised attention output); the per-pixel kernel rebuilds its
// dA tile from dY on the matrix cores.
// ---------------------------------------------------------------------------------------------------
// Forward fused with the projection: y[n][c] = sum_(h,e) out[n][(h,e)] W[c][(h,e)] + bias[c] + resid[n][c]; the attention output
// out (HD channels per pixel) is never written.  Same transposed chaining: T^T[e][pixel] = sum_d ctx[d][e] qs[pixel][d] leaves
// the matrix core in the B-operand layout of y^T[c][pixel] += W_h[c][e] T^T[e][pixel]; y^T has one pixel per lane with
// groups of four consecutive channels -> float4 bias / residual / store.  One wave = 32 pixels, N % 128 == 0, Cout = 32*CT.
template <int CT>
__global__ void __launch_bounds__(256) la_out_proj_kernel(const float* __restrict__ qkv, const float* __restrict__ ctx,
                                                          const float* __restrict__ wt, const float* __restrict__ bias,
                                                          const float* __restrict__ resid, float* __restrict__ y,
                                                          float* __restrict__ qstat, int N, int heads, float scale) {
  // wt = W^T [HD][Cout] (la_wt_kernel): the A operand W_h[c = lane][e] is then a coalesced 128-byte row per e, served by L1/L2
  // like the context - no LDS, no barrier, and the next head's q is in flight while this head computes
  constexpr int CO = 32 * CT;
  const int HD = heads * DH;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, half = lane >> 5, l31 = lane & 31;
  const size_t p = ((size_t)blockIdx.x * 4 + wave) * 32 + l31;
  const size_t b = ((size_t)blockIdx.x * 128) / N;
  f32x16 accY[CT];
  for (int t = 0; t < CT; ++t)
    for (int r = 0; r < 16; ++r) accY[t][r] = 0.f;
  const float* qp = qkv + p * 3 * HD + 16 * half;
  f32x4 qn[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) qn[k] = *reinterpret_cast<const f32x4*>(qp + 4 * k);
  // MFMA A operands (context rows, W^T rows) do not depend on the pixels: they are fetched one whole head ahead into registers.
  // Left to the compiler, each dependent MFMA waited on a load issued four MFMAs earlier (L2 latency > 4 x 64 cycles): the
  // chain ran at ~600 cycles per MFMA instead of 64.
  constexpr int WT = (CT <= 2) ? CT : 1;                   // output-channel tiles whose W rows are fetched ahead (register budget)
  float cn[16], wn[16 * WT];
  {
    const float* cb = ctx + (b * heads) * 1024 + (16 * half) * 32 + l31;
    const float* wh = wt + (size_t)(4 * half) * CO + l31;
#pragma unroll
    for (int s2 = 0; s2 < 16; ++s2) cn[s2] = cb[s2 * 32];
#pragma unroll
    for (int t = 0; t < WT; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) wn[16 * t + r] = wh[((r & 3) + 8 * (r >> 2)) * CO + 32 * t];
  }
  for (int h = 0; h < heads; ++h) {
    float qv[16], ca[16], wa[16 * WT];
#pragma unroll
    for (int k = 0; k < 4; ++k) { qv[4 * k] = qn[k][0]; qv[4 * k + 1] = qn[k][1]; qv[4 * k + 2] = qn[k][2]; qv[4 * k + 3] = qn[k][3]; }
#pragma unroll
    for (int i = 0; i < 16; ++i) ca[i] = cn[i];
#pragma unroll
    for (int i = 0; i < 16 * WT; ++i) wa[i] = wn[i];
    if (h + 1 < heads) {
#pragma unroll
      for (int k = 0; k < 4; ++k) qn[k] = *reinterpret_cast<const f32x4*>(qp + (h + 1) * DH + 4 * k);
      const float* cb = ctx + (b * heads + h + 1) * 1024 + (16 * half) * 32 + l31;
      const float* wh = wt + (size_t)((h + 1) * DH + 4 * half) * CO + l31;
#pragma unroll
      for (int s2 = 0; s2 < 16; ++s2) cn[s2] = cb[s2 * 32];
#pragma unroll
      for (int t = 0; t < WT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) wn[16 * t + r] = wh[((r & 3) + 8 * (r >> 2)) * CO + 32 * t];
    }
    float m = qv[0];
#pragma unroll
    for (int k = 1; k < 16; ++k) m = fmaxf(m, qv[k]);
    m = fmaxf(m, pidm_other_half(m));
    float sum = 0.f;
#pragma unroll
    for (int k = 0; k < 16; ++k) {
      qv[k] = fexp(qv[k] - m);
      sum += qv[k];
    }
    sum += pidm_other_half(sum);
    const float inv = 1.f / sum;
    if (half == 0) *reinterpret_cast<float2*>(qstat + (p * heads + h) * 2) = make_float2(m, inv);
    f32x16 accT;
    for (int r = 0; r < 16; ++r) accT[r] = 0.f;
    const float is = inv * scale;
#pragma unroll
    for (int s2 = 0; s2 < 16; ++s2) accT = __builtin_amdgcn_mfma_f32_32x32x2f32(ca[s2], qv[s2] * is, accT, 0, 0, 0);
#pragma unroll
    for (int t = 0; t < WT; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) accY[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(wa[16 * t + r], accT[r], accY[t], 0, 0, 0);
    if (WT < CT) {
      const float* wh = wt + (size_t)(h * DH + 4 * half) * CO + l31;
#pragma unroll
      for (int t = WT; t < CT; ++t) {
        float wl[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) wl[r] = wh[((r & 3) + 8 * (r >> 2)) * CO + 32 * t];
#pragma unroll
        for (int r = 0; r < 16; ++r) accY[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(wl[r], accT[r], accY[t], 0, 0, 0);
      }
    }
  }
#pragma unroll
  for (int t = 0; t < CT; ++t)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int c0 = 32 * t + 8 * j + 4 * half;
      f32x4 o = {accY[t][4 * j], accY[t][4 * j + 1], accY[t][4 * j + 2], accY[t][4 * j + 3]};
      if (bias) o += *reinterpret_cast<const f32x4*>(bias + c0);
      if (resid) o += *reinterpret_cast<const f32x4*>(resid + p * CO + c0);
      *reinterpret_cast<f32x4*>(y + p * CO + c0) = o;
    }
}

// wt[hd][c] = w[c][hd]  (Cout x HD, a few 10 KB)
__global__ void la_wt_kernel(const float* __restrict__ w, float* __restrict__ wt, int Cout, int HD) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < Cout * HD) wt[(size_t)(i % HD) * Cout + i / HD] = w[i];
}

template <int NT>
__global__ void __launch_bounds__(256) la_g_kernel(const float* __restrict__ qkv, const float* __restrict__ qstat,
                                                   const float* __restrict__ dy, int ld_dy, float* __restrict__ Gpart, int N,
                                                   int heads, float scale) {
  __shared__ float red[4][1024];
  const int bh = blockIdx.x, b = bh / heads, h = bh % heads;
  const int ns = blockIdx.y, NS = gridDim.y;
  const int HD = heads * DH;
  const int tid = threadIdx.x, lane = tid & 63, half = lane >> 5, l31 = lane & 31;
  // the wave index as a scalar: pixel ranges, loop bounds and row pointers then live in SGPRs and every load is
  // (uniform row pointer) + (32-bit lane offset) - the 64-bit per-lane address arithmetic was ~20 VALU per MFMA
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  f32x16 acc[NT];
  for (int t = 0; t < NT; ++t)
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
  const int per_blk = (N + NS - 1) / NS;
  const int blk_lo = ns * per_blk, blk_hi = (blk_lo + per_blk < N) ? blk_lo + per_blk : N;
  const int per = (blk_hi - blk_lo + 3) / 4;
  const int n_lo = blk_lo + wave * per, n_hi = (n_lo + per < blk_hi) ? n_lo + per : blk_hi;
  const float* qb = qkv + (size_t)b * N * 3 * HD + h * DH + l31;
  const float* sb = qstat + ((size_t)b * N * heads + h) * 2;
  const float* yb = dy + (size_t)b * N * ld_dy + l31;
  const int lq = half * 3 * HD + h * DH + l31, ls = (half * heads + h) * 2, ly = half * ld_dy + l31;
  constexpr int UNR = (NT <= 2) ? 4 : 2;
  int n0 = n_lo;
  for (; n0 + 2 * UNR <= n_hi; n0 += 2 * UNR) {       // wave-uniform bounds, UNR independent pixel pairs in flight
    float qa[UNR], bv[UNR][NT];
    float2 qs[UNR];
    const float* qrow = qkv + ((size_t)b * N + n0) * 3 * HD;
    const float* srow = qstat + ((size_t)b * N + n0) * heads * 2;
    const float* yrow = dy + ((size_t)b * N + n0) * ld_dy;
#pragma unroll
    for (int u = 0; u < UNR; ++u) {
      qa[u] = (qrow + (size_t)(2 * u) * 3 * HD)[lq];
      qs[u] = *reinterpret_cast<const float2*>(srow + (size_t)(2 * u) * heads * 2 + ls);
#pragma unroll
      for (int t = 0; t < NT; ++t) bv[u][t] = (yrow + (size_t)(2 * u) * ld_dy + 32 * t)[ly];
    }
#pragma unroll
    for (int u = 0; u < UNR; ++u) {
      const float a = fexp(qa[u] - qs[u].x) * qs[u].y * scale;
#pragma unroll
      for (int t = 0; t < NT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, bv[u][t], acc[t], 0, 0, 0);
    }
  }
  for (; n0 < n_hi; n0 += 2) {
    const int n = n0 + half;
    float a = 0.f, bv[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) bv[t] = 0.f;
    if (n < n_hi) {
      const float2 qs = *reinterpret_cast<const float2*>(sb + (size_t)n * heads * 2);
      a = fexp(qb[(size_t)n * 3 * HD] - qs.x) * qs.y * scale;
#pragma unroll
      for (int t = 0; t < NT; ++t) bv[t] = yb[(size_t)n * ld_dy + 32 * t];
    }
#pragma unroll
    for (int t = 0; t < NT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, bv[t], acc[t], 0, 0, 0);
  }
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    if (t) __syncthreads();
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = (r & 3) + 8 * (r >> 2) + 4 * half;
      red[wave][row * 32 + l31] = acc[t][r];
    }
    __syncthreads();
    float* out = Gpart + (((size_t)bh * NS + ns) * NT + t) * 1024;
    for (int e = tid; e < 1024; e += 256) out[e] = (red[0][e] + red[1][e]) + (red[2][e] + red[3][e]);
  }
}

// per (b,h): G = sum of the pixel-split partials (fixed order); dctx = G W_h; rowdot[d] = sum_e dctx[d][e] ctx[d][e];
// this image's share of the to_out weight gradient dWpart[b][c][(h,e)] = sum_d G[d][c] ctx[d][e] (summed over b by the
// caller's deterministic split reduction).  Cout = 32*NT.
template <int NT>
__global__ void __launch_bounds__(256) la_g_final_kernel(const float* __restrict__ Gpart, const float* __restrict__ ctx,
                                                         const float* __restrict__ w_out, float* __restrict__ dctx,
                                                         float* __restrict__ rowdot, float* __restrict__ dwpart, int NS,
                                                         int heads) {
  constexpr int CO = 32 * NT;
  __shared__ float sG[32][CO + 1], sC[32][33], sP[32][33];
  const int bh = blockIdx.x, b = bh / heads, h = bh % heads, tid = threadIdx.x;
  const int HD = heads * DH;
  for (int i = tid; i < NT * 1024; i += 256) {
    const int t = i >> 10, r = i & 1023;
    float v = 0.f;
    for (int k = 0; k < NS; ++k) v += Gpart[(((size_t)bh * NS + k) * NT + t) * 1024 + r];
    sG[r >> 5][32 * t + (r & 31)] = v;
  }
  for (int i = tid; i < 1024; i += 256) sC[i >> 5][i & 31] = ctx[(size_t)bh * 1024 + i];
  __syncthreads();
  const int e = tid & 31, r0 = tid >> 5;
  float acc[4] = {0.f, 0.f, 0.f, 0.f};
  for (int c = 0; c < CO; ++c) {
    const float w = w_out[(size_t)c * HD + h * DH + e];
#pragma unroll
    for (int i = 0; i < 4; ++i) acc[i] = fmaf(sG[r0 + 8 * i][c], w, acc[i]);
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int d = r0 + 8 * i;
    dctx[(size_t)bh * 1024 + d * 32 + e] = acc[i];
    sP[d][e] = acc[i] * sC[d][e];
  }
  for (int c = r0; c < CO; c += 8) {
    float a = 0.f;
#pragma unroll
    for (int d = 0; d < 32; ++d) a = fmaf(sG[d][c], sC[d][e], a);
    dwpart[((size_t)b * CO + c) * HD + h * DH + e] = a;
  }
  __syncthreads();
  if (tid < 32) {
    float sacc = 0.f;
    for (int j = 0; j < 32; ++j) sacc += sP[tid][j];
    rowdot[(size_t)bh * 32 + tid] = sacc;
  }
}

// per pixel (N % 128 == 0), one wave = 32 pixels, loops over heads; two kernels with disjoint inputs (dq: q, dY; dk/dv: k, v), each
// small enough in registers for 3 waves per SIMD.  Everything runs TRANSPOSED (rows = channels, columns = pixels):
//   dA^T = W_h^T dY^T lands in the MFMA C layout, which IS the B-operand layout of dq'^T = ctx dA^T (k = e in the accumulator's own
//   row order) - no shuffle, no LDS round trip;
//   every result has one pixel per lane and groups of four consecutive channels -> q / k / dq / dk / dv move as float4, the
//   softmax-Jacobian dot over d is an in-lane sum plus one cross-half exchange, the k-softmax constants are LDS broadcasts.
// All heads' 32x32 matrices are staged in LDS once: no barrier in the head loop, next head's operands in flight during the MFMAs.
static const int kLaTileLd = 36;                            // row stride of a wave's 32x32 transposition tile (16-byte aligned, conflict-free)
__global__ void __launch_bounds__(256, 3) la_bwd_dq_kernel(const float* __restrict__ qkv, const float* __restrict__ qstat,
                                                           const float* __restrict__ ctx, const float* __restrict__ dy, int ld_dy,
                                                           const float* __restrict__ w_out, int Cout, float* __restrict__ dqkv,
                                                           int N, int heads, float scale) {
  HIP_DYNAMIC_SHARED(float, sm)                              // [heads][32][33] ctx
  const int HD = heads * DH;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, half = lane >> 5, l31 = lane & 31;
  const size_t pblk = (size_t)blockIdx.x * 128;
  const int b = (int)(pblk / N);
  const size_t pa = pblk + wave * 32 + l31;
  const int CC = Cout / 32;
  for (int i = tid; i < heads * 1024; i += 256) sm[(i >> 10) * 1056 + ((i & 1023) >> 5) * 33 + (i & 31)] = ctx[(size_t)b * heads * 1024 + i];
  __syncthreads();
  float* tb = sm + heads * 1056 + wave * (32 * kLaTileLd);
  const int trow = lane >> 3, tcol = 4 * (lane & 7);
  const float* rowA = qkv + pa * 3 * HD;
  f32x4 n_q[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) n_q[j] = *reinterpret_cast<const f32x4*>(rowA + 8 * j + 4 * half);
  float2 n_qst = *reinterpret_cast<const float2*>(qstat + pa * heads * 2);
  f32x4 by0[4];                                             // the dY tile is the same for every head (first 32 channels kept)
#pragma unroll
  for (int j = 0; j < 4; ++j) by0[j] = *reinterpret_cast<const f32x4*>(dy + pa * ld_dy + 16 * half + 4 * j);
  float wn[16];                                             // W rows (MFMA A operands) of the next head, fetched a head ahead
#pragma unroll
  for (int i = 0; i < 16; ++i) wn[i] = w_out[(size_t)(16 * half + i) * HD + l31];
  for (int h = 0; h < heads; ++h) {
    const float* sc = sm + h * 1056;
    f32x4 q4[4];
    float wa[16];
#pragma unroll
    for (int j = 0; j < 4; ++j) q4[j] = n_q[j];
#pragma unroll
    for (int i = 0; i < 16; ++i) wa[i] = wn[i];
    const float2 qst = n_qst;
    if (h + 1 < heads) {
#pragma unroll
      for (int i = 0; i < 16; ++i) wn[i] = w_out[(size_t)(16 * half + i) * HD + (h + 1) * DH + l31];
#pragma unroll
      for (int j = 0; j < 4; ++j) n_q[j] = *reinterpret_cast<const f32x4*>(rowA + (h + 1) * DH + 8 * j + 4 * half);   // q[pixel][d = 8j + 4half + 0..3]
      n_qst = *reinterpret_cast<const float2*>(qstat + (pa * heads + h + 1) * 2);
    }
    // dA^T[e][pixel] = sum_c W[c][(h,e)] dY[pixel][c]      (k = c = 32cc + 16half + s; W rows straight from global: lane = e)
    f32x16 accT;
    for (int r = 0; r < 16; ++r) accT[r] = 0.f;
    const float* wh = w_out + (size_t)(16 * half) * HD + h * DH + l31;
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int c = 0; c < 4; ++c) accT = __builtin_amdgcn_mfma_f32_32x32x2f32(wa[4 * j + c], by0[j][c], accT, 0, 0, 0);
    for (int cc = 1; cc < CC; ++cc) {
      f32x4 by[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) by[j] = *reinterpret_cast<const f32x4*>(dy + pa * ld_dy + 32 * cc + 16 * half + 4 * j);
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int c = 0; c < 4; ++c)
          accT = __builtin_amdgcn_mfma_f32_32x32x2f32(wh[(size_t)(32 * cc + 4 * j + c) * HD], by[j][c], accT, 0, 0, 0);
    }
    // dq'^T[d][pixel] = sum_e ctx[d][e] dA^T[e][pixel]: register r of accT holds e = (r&3) + 8(r>>2) + 4half
    f32x16 acc1;
    for (int r = 0; r < 16; ++r) acc1[r] = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r)
      acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(sc[l31 * 33 + (r & 3) + 8 * (r >> 2) + 4 * half], accT[r], acc1, 0, 0, 0);
    // this lane's pixel, d = (r&3) + 8(r>>2) + 4half = component (r&3) of q4[r>>2]
    f32x4 qs4[4];
    float dot = 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        qs4[j][c] = fexp(q4[j][c] - qst.x) * qst.y;
        dot = fmaf(qs4[j][c], acc1[4 * j + c], dot);
      }
    dot += pidm_other_half(dot);
    // one pixel per lane -> whole 128-byte lines through the wave's LDS tile (see la_bwd_dkdv_kernel)
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      f32x4 o;
#pragma unroll
      for (int c = 0; c < 4; ++c) o[c] = scale * qs4[j][c] * (acc1[4 * j + c] - dot);
      *reinterpret_cast<f32x4*>(tb + l31 * kLaTileLd + 8 * j + 4 * half) = o;
    }
    __builtin_amdgcn_wave_barrier();
    float* orow = dqkv + (pblk + wave * 32) * 3 * HD + h * DH;
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      const f32x4 v = *reinterpret_cast<const f32x4*>(tb + (trow + 8 * it) * kLaTileLd + tcol);
      *reinterpret_cast<f32x4*>(orow + (size_t)(trow + 8 * it) * 3 * HD + tcol) = v;
    }
  }
}

// dk = P (dP/N - rowdot), dP^T[d][pixel] = sum_e dctx[d][e] v[pixel][e];  dv^T[e][pixel] = (1/N) sum_d dctx[d][e] P[pixel][d]
static const int kLaDkvLds = 32 * 33 + 96;                  // floats per head: dctx, rowdot, k max, k 1/sum
__global__ void __launch_bounds__(256, 3) la_bwd_dkdv_kernel(const float* __restrict__ qkv, const float* __restrict__ kstat,
                                                             const float* __restrict__ dctx, const float* __restrict__ rowdot,
                                                             float* __restrict__ dqkv, int N, int heads) {
  HIP_DYNAMIC_SHARED(float, sm)
  const int HD = heads * DH;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, half = lane >> 5, l31 = lane & 31;
  const size_t pblk = (size_t)blockIdx.x * 128;
  const int b = (int)(pblk / N);
  const size_t pa = pblk + wave * 32 + l31;
  const float invN = 1.f / (float)N;
  for (int i = tid; i < heads * 1024; i += 256)
    sm[(i >> 10) * kLaDkvLds + ((i & 1023) >> 5) * 33 + (i & 31)] = dctx[(size_t)b * heads * 1024 + i];
  for (int i = tid; i < heads * 32; i += 256) {
    float* hs = sm + (i >> 5) * kLaDkvLds + 1056;
    hs[i & 31] = rowdot[(size_t)b * heads * 32 + i];
    hs[32 + (i & 31)] = kstat[((size_t)b * HD + i) * 2];
    hs[64 + (i & 31)] = kstat[((size_t)b * HD + i) * 2 + 1];
  }
  __syncthreads();
  float* tb = sm + heads * kLaDkvLds + wave * (32 * kLaTileLd);
  const float* rowK = qkv + pa * 3 * HD + HD;
  f32x4 n_k[4], n_v[4], n_k4[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    n_k[j] = *reinterpret_cast<const f32x4*>(rowK + 16 * half + 4 * j);            // MFMA k order: channel 16half + 4j + c
    n_v[j] = *reinterpret_cast<const f32x4*>(rowK + HD + 16 * half + 4 * j);
    n_k4[j] = *reinterpret_cast<const f32x4*>(rowK + 8 * j + 4 * half);            // result row order: channel 8j + 4half + c
  }
  for (int h = 0; h < heads; ++h) {
    const float* sd = sm + h * kLaDkvLds;
    const float* srd = sd + 1056;
    const float* skm = srd + 32;
    const float* skis = srd + 64;
    f32x4 a_k[4], a_v[4], k4[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) { a_k[j] = n_k[j]; a_v[j] = n_v[j]; k4[j] = n_k4[j]; }
    if (h + 1 < heads) {
      const float* rn = rowK + (h + 1) * DH;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        n_k[j] = *reinterpret_cast<const f32x4*>(rn + 16 * half + 4 * j);
        n_v[j] = *reinterpret_cast<const f32x4*>(rn + HD + 16 * half + 4 * j);
        n_k4[j] = *reinterpret_cast<const f32x4*>(rn + 8 * j + 4 * half);
      }
    }
    f32x16 acc2, acc3;
    for (int r = 0; r < 16; ++r) { acc2[r] = 0.f; acc3[r] = 0.f; }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const int kk = 16 * half + 4 * j + c;
        acc2 = __builtin_amdgcn_mfma_f32_32x32x2f32(sd[l31 * 33 + kk], a_v[j][c], acc2, 0, 0, 0);      // dP^T[d][n] += dctx[d][e] v[n][e]
        const float pB = fexp(a_k[j][c] - skm[kk]) * skis[kk];
        acc3 = __builtin_amdgcn_mfma_f32_32x32x2f32(sd[kk * 33 + l31], pB, acc3, 0, 0, 0);             // dv^T[e][n] += dctx[d][e] P[n][d]
      }
    }
    // results have one pixel per lane: written like that, a store instruction puts 32 bytes into each of 32 lines (measured
    // 3.6 TB/s marginal against 5.6 for the loads).  Through a wave-private LDS tile they leave as whole 128-byte lines.
    f32x4 o_k[4], o_v[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const int d = 8 * j + 4 * half + c;
        const float P = fexp(k4[j][c] - skm[d]) * skis[d];
        o_k[j][c] = P * (acc2[4 * j + c] * invN - srd[d]);
        o_v[j][c] = acc3[4 * j + c] * invN;
      }
    }
    float* ok = dqkv + (pblk + wave * 32) * 3 * HD + HD + h * DH;
    const int trow = lane >> 3, tcol = 4 * (lane & 7);
#pragma unroll
    for (int which = 0; which < 2; ++which) {
      __builtin_amdgcn_wave_barrier();                      // the previous tile has been read out
#pragma unroll
      for (int j = 0; j < 4; ++j) *reinterpret_cast<f32x4*>(tb + l31 * kLaTileLd + 8 * j + 4 * half) = which ? o_v[j] : o_k[j];
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int it = 0; it < 4; ++it) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(tb + (trow + 8 * it) * kLaTileLd + tcol);
        *reinterpret_cast<f32x4*>(ok + (size_t)(trow + 8 * it) * 3 * HD + which * HD + tcol) = v;
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------------
// bottleneck softmax attention over N <= 64 tokens, one workgroup per (b, h)
// ---------------------------------------------------------------------------------------------------
template <bool BWD>
__global__ void __launch_bounds__(256) mid_attn_kernel(const float* __restrict__ qkv, const float* __restrict__ dO,
                                                       float* __restrict__ out,   // FWD: O [B][N][HD] ; BWD: dqkv [B][N][3HD]
                                                       int N, int heads, float scale) {
  // dynamic LDS carve: q,k,v,(dO) as [N][33]; S,(dS) as [N][N+1]
  HIP_DYNAMIC_SHARED(float, smem)
  float (*sq)[33] = reinterpret_cast<float (*)[33]>(smem);
  float (*sk)[33] = reinterpret_cast<float (*)[33]>(smem + (size_t)N * 33);
  float (*sv)[33] = reinterpret_cast<float (*)[33]>(smem + (size_t)2 * N * 33);
  float (*sdo)[33] = reinterpret_cast<float (*)[33]>(smem + (size_t)3 * N * 33);
  float* sS_ = smem + (size_t)4 * N * 33;
  float* sdS_ = sS_ + (size_t)N * (N + 1);
  const int LS = N + 1;
#define sS(i, j) sS_[(i) * LS + (j)]
#define sdS(i, j) sdS_[(i) * LS + (j)]
  const int HD = heads * DH;
  const int bh = blockIdx.x, b = bh / heads, h = bh % heads, tid = threadIdx.x;
  for (int e = tid; e < N * 32; e += 256) {
    const int n = e >> 5, d = e & 31;
    const float* row = qkv + ((size_t)b * N + n) * 3 * HD + h * DH + d;
    sq[n][d] = row[0] * scale;
    sk[n][d] = row[HD];
    sv[n][d] = row[2 * HD];
    if (BWD) sdo[n][d] = dO[((size_t)b * N + n) * HD + h * DH + d];
  }
  __syncthreads();
  for (int e = tid; e < N * N; e += 256) {
    const int i = e / N, j = e % N;
    float s = 0.f;
#pragma unroll
    for (int d = 0; d < 32; ++d) s = fmaf(sq[i][d], sk[j][d], s);
    sS(i, j) = s;
  }
  __syncthreads();
  if (tid < N) {
    float m = -3.0e38f;
    for (int j = 0; j < N; ++j) m = fmaxf(m, sS(tid, j));
    float s = 0.f;
    for (int j = 0; j < N; ++j) {
      const float ex = fexp(sS(tid, j) - m);
      sS(tid, j) = ex;
      s += ex;
    }
    const float inv = 1.f / s;
    for (int j = 0; j < N; ++j) sS(tid, j) *= inv;
  }
  __syncthreads();
  if (!BWD) {
    for (int e = tid; e < N * 32; e += 256) {
      const int i = e >> 5, d = e & 31;
      float o = 0.f;
      for (int j = 0; j < N; ++j) o = fmaf(sS(i, j), sv[j][d], o);
      out[((size_t)b * N + i) * HD + h * DH + d] = o;
    }
    return;
  }
  // dV[j][d] = sum_i P[i][j] dO[i][d]
  for (int e = tid; e < N * 32; e += 256) {
    const int j = e >> 5, d = e & 31;
    float o = 0.f;
    for (int i = 0; i < N; ++i) o = fmaf(sS(i, j), sdo[i][d], o);
    out[((size_t)b * N + j) * 3 * HD + 2 * HD + h * DH + d] = o;
  }
  __syncthreads();
  // dP[i][j] = sum_d dO[i][d] v[j][d]; dS = P*(dP - sum_j dP*P)   (overwrite sv rows are still needed: keep)
  for (int e = tid; e < N * N; e += 256) {
    const int i = e / N, j = e % N;
    float s = 0.f;
#pragma unroll
    for (int d = 0; d < 32; ++d) s = fmaf(sdo[i][d], sv[j][d], s);
    sdS(i, j) = s;
  }
  __syncthreads();
  if (tid < N) {
    float dot = 0.f;
    for (int j = 0; j < N; ++j) dot = fmaf(sdS(tid, j), sS(tid, j), dot);
    for (int j = 0; j < N; ++j) sdS(tid, j) = sS(tid, j) * (sdS(tid, j) - dot);
  }
  __syncthreads();
  // dq[i][d] = scale * sum_j dS[i][j] k[j][d] ; dk[j][d] = sum_i dS[i][j] * (scale*q[i][d]) (sq already holds scale*q)
  for (int e = tid; e < N * 32; e += 256) {
    const int i = e >> 5, d = e & 31;
    float a = 0.f, c = 0.f;
    for (int j = 0; j < N; ++j) {
      a = fmaf(sdS(i, j), sk[j][d], a);
      c = fmaf(sdS(j, i), sq[j][d], c);
    }
    out[((size_t)b * N + i) * 3 * HD + h * DH + d] = a * scale;
    out[((size_t)b * N + i) * 3 * HD + HD + h * DH + d] = c;
  }
}

#undef sS
#undef sdS

// The same for exactly 64 tokens (the 8 x 8 bottleneck of every 64 x 64 configuration), with every product on the fp32 matrix cores
// (v_mfma_f32_32x32x2_f32: exact fp32 products) instead of scalar FMA loops over LDS, and the row softmax / Jacobian spread over all
// 256 threads (4 per row).  A 32 x 32 output tile per wave; operands are read from the padded LDS images as one scalar per lane and
// k-step (A: row l & 31, k = lane >> 5; B: column l & 31).  Round 4: 62 -> ~25 us (backward) per launch at batch 64.
template <bool BWD>
__global__ void __launch_bounds__(256) mid_attn64_kernel(const float* __restrict__ qkv, const float* __restrict__ dO,
                                                         float* __restrict__ out, int heads, float scale) {
  constexpr int N = 64, LS = 65;
  HIP_DYNAMIC_SHARED(float, smem)
  float (*sq)[33] = reinterpret_cast<float (*)[33]>(smem);
  float (*sk)[33] = reinterpret_cast<float (*)[33]>(smem + (size_t)N * 33);
  float (*sv)[33] = reinterpret_cast<float (*)[33]>(smem + (size_t)2 * N * 33);
  float (*sdo)[33] = reinterpret_cast<float (*)[33]>(smem + (size_t)3 * N * 33);
  float* sS_ = smem + (size_t)4 * N * 33;
  float* sdS_ = sS_ + (size_t)N * LS;
  const int HD = heads * DH;
  const int bh = blockIdx.x, b = bh / heads, h = bh % heads, tid = threadIdx.x;
  const int lane = tid & 63, half = lane >> 5, l31 = lane & 31, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  for (int e = tid; e < N * 32; e += 256) {
    const int n = e >> 5, d = e & 31;
    const float* row = qkv + ((size_t)b * N + n) * 3 * HD + h * DH + d;
    sq[n][d] = row[0] * scale;
    sk[n][d] = row[HD];
    sv[n][d] = row[2 * HD];
    if (BWD) sdo[n][d] = dO[((size_t)b * N + n) * HD + h * DH + d];
  }
  __syncthreads();
  const int ti = wave >> 1, tj = wave & 1;
  // S = (scale q) k^T: wave (ti, tj) owns tile rows 32 ti.., columns 32 tj..
  {
    f32x16 acc;
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
    for (int s2 = 0; s2 < 16; ++s2)
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(sq[32 * ti + l31][2 * s2 + half], sk[32 * tj + l31][2 * s2 + half], acc, 0, 0, 0);
#pragma unroll
    for (int r = 0; r < 16; ++r) sS_[(32 * ti + (r & 3) + 8 * (r >> 2) + 4 * half) * LS + 32 * tj + l31] = acc[r];
  }
  __syncthreads();
  // row softmax: 4 threads per row, 16 columns each
  {
    const int row = tid >> 2, c0 = (tid & 3) * 16;
    float* pr = sS_ + row * LS + c0;
    float m = pr[0];
#pragma unroll
    for (int j = 1; j < 16; ++j) m = fmaxf(m, pr[j]);
    m = fmaxf(m, pidm_quad_xor1(m));
    m = fmaxf(m, pidm_quad_xor2(m));
    float ex[16], sum = 0.f;
#pragma unroll
    for (int j = 0; j < 16; ++j) { ex[j] = fexp(pr[j] - m); sum += ex[j]; }
    sum += pidm_quad_xor1(sum);
    sum += pidm_quad_xor2(sum);
    const float inv = 1.f / sum;
#pragma unroll
    for (int j = 0; j < 16; ++j) pr[j] = ex[j] * inv;
  }
  __syncthreads();
  if (!BWD) {
    // O = P v: 64 x 32, K = 64: waves 0 / 1 take the two row tiles
    if (wave < 2) {
      f32x16 acc;
      for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll 8
      for (int s2 = 0; s2 < 32; ++s2)
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(sS_[(32 * wave + l31) * LS + 2 * s2 + half], sv[2 * s2 + half][l31], acc, 0, 0, 0);
#pragma unroll
      for (int r = 0; r < 16; ++r)
        out[((size_t)b * N + 32 * wave + (r & 3) + 8 * (r >> 2) + 4 * half) * HD + h * DH + l31] = acc[r];
    }
    return;
  }
  // dP = dO v^T (all four waves, one tile each), then dV = P^T dO (64 x 32, K = 64: waves 0 / 1)
  {
    f32x16 acc;
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
    for (int s2 = 0; s2 < 16; ++s2)
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(sdo[32 * ti + l31][2 * s2 + half], sv[32 * tj + l31][2 * s2 + half], acc, 0, 0, 0);
#pragma unroll
    for (int r = 0; r < 16; ++r) sdS_[(32 * ti + (r & 3) + 8 * (r >> 2) + 4 * half) * LS + 32 * tj + l31] = acc[r];
  }
  if (wave < 2) {
    f32x16 acc;
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll 8
    for (int s2 = 0; s2 < 32; ++s2)    // A[j][i] = P[i][j]: a column of P per lane (LS is odd: conflict-free)
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(sS_[(2 * s2 + half) * LS + 32 * wave + l31], sdo[2 * s2 + half][l31], acc, 0, 0, 0);
#pragma unroll
    for (int r = 0; r < 16; ++r)
      out[((size_t)b * N + 32 * wave + (r & 3) + 8 * (r >> 2) + 4 * half) * 3 * HD + 2 * HD + h * DH + l31] = acc[r];
  }
  __syncthreads();
  // dS = P (dP - sum_j dP P), 4 threads per row
  {
    const int row = tid >> 2, c0 = (tid & 3) * 16;
    const float* pp = sS_ + row * LS + c0;
    float* pd = sdS_ + row * LS + c0;
    float dot = 0.f;
#pragma unroll
    for (int j = 0; j < 16; ++j) dot = fmaf(pd[j], pp[j], dot);
    dot += pidm_quad_xor1(dot);
    dot += pidm_quad_xor2(dot);
#pragma unroll
    for (int j = 0; j < 16; ++j) pd[j] = pp[j] * (pd[j] - dot);
  }
  __syncthreads();
  // dq = scale dS k (waves 0 / 1), dk = dS^T (scale q) (waves 2 / 3): 64 x 32, K = 64 each
  {
    const int t = wave & 1;
    f32x16 acc;
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    if (wave < 2) {
#pragma unroll 8
      for (int s2 = 0; s2 < 32; ++s2)
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(sdS_[(32 * t + l31) * LS + 2 * s2 + half], sk[2 * s2 + half][l31], acc, 0, 0, 0);
    } else {
#pragma unroll 8
      for (int s2 = 0; s2 < 32; ++s2)
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(sdS_[(2 * s2 + half) * LS + 32 * t + l31], sq[2 * s2 + half][l31], acc, 0, 0, 0);
    }
    const float f = wave < 2 ? scale : 1.f;
    const size_t col = (wave < 2 ? 0 : HD) + h * DH + l31;
#pragma unroll
    for (int r = 0; r < 16; ++r)
      out[((size_t)b * N + 32 * t + (r & 3) + 8 * (r >> 2) + 4 * half) * 3 * HD + col] = acc[r] * f;
  }
}

// ---------------------------------------------------------------------------------------------------
// launchers
// ---------------------------------------------------------------------------------------------------
static int la_nsplit(int N) {
  int ns = N / 512;
  if (ns < 1) ns = 1;
  if (ns > 8) ns = 8;
  return ns;
}
static int la_kseg(int N) {
  int s = N / 256;
  if (s < 1) s = 1;
  if (s > 16) s = 16;
  return s;
}
// floats of scratch needed by launch_la_forward / launch_la_backward
size_t la_scratch_floats(int B, int N, int heads) {
  const size_t a = (size_t)B * la_kseg(N) * heads * DH * 2, b = (size_t)B * heads * la_nsplit(N) * 1024;
  return a + b + (size_t)heads * DH * 128 + 64;   // + the transposed to_out weights of the fused forward (Cout <= 128)
}

int launch_la_forward(const float* qkv, float* kstat, float* ctx, float* attn, float* qstat, int B, int N, int heads,
                      float* scratch, hipStream_t st) {
  const int HD = heads * DH;
  const float scale = 0.17677669529663687f;  // 32^-0.5
  const int nseg = la_kseg(N), NS = la_nsplit(N);
  float* kpart = scratch;
  float* dpart = scratch + (size_t)B * nseg * HD * 2;
  hipLaunchKernelGGL(la_kstats_kernel, dim3(cdiv(HD, 64), B, nseg), dim3(256), 0, st, qkv, N, HD, nseg, kpart, nseg == 1 ? kstat : nullptr);
  PIDM_CHECK_LAUNCH("la_kstats_kernel");
  if (nseg > 1) {
    hipLaunchKernelGGL(la_kstats_final_kernel, dim3(cdiv(B * HD, 256)), dim3(256), 0, st, kpart, B, HD, nseg, kstat);
    PIDM_CHECK_LAUNCH("la_kstats_final_kernel");
  }
  hipLaunchKernelGGL(HIP_KERNEL_NAME(la_nreduce_kernel<0>), dim3(B * heads, NS), dim3(256), 0, st, qkv, kstat, nullptr, nullptr,
                     dpart, N, heads, scale, NS == 1 ? ctx : nullptr, nullptr, 1.f / (float)N);
  PIDM_CHECK_LAUNCH("la_context");
  if (NS > 1) {
    hipLaunchKernelGGL(HIP_KERNEL_NAME(la_nreduce_final_kernel<0>), dim3(B * heads), dim3(256), 0, st, dpart, nullptr, ctx, nullptr, NS,
                       1.f / (float)N);
    PIDM_CHECK_LAUNCH("la_context_final");
  }
  const size_t npix = (size_t)B * N;
  if (N % 32 == 0)
    hipLaunchKernelGGL(HIP_KERNEL_NAME(la_out_kernel<true>), dim3((unsigned)((npix + 127) / 128), (npix + 127) / 128 < 1024 ? heads : 1), dim3(256), 0, st, qkv, ctx, attn,
                       qstat, N, heads, npix, scale);
  else
    hipLaunchKernelGGL(HIP_KERNEL_NAME(la_out_kernel<false>), dim3((unsigned)((npix + 127) / 128), (npix + 127) / 128 < 1024 ? heads : 1), dim3(256), 0, st, qkv, ctx, attn,
                       qstat, N, heads, npix, scale);
  PIDM_CHECK_LAUNCH("la_out_kernel");
  return 0;
}

int launch_la_backward(const float* qkv, const float* kstat, const float* qstat, const float* ctx, const float* dA, float* dctx,
                       float* rowdot, float* dqkv, int B, int N, int heads, float* scratch, hipStream_t st) {
  const float scale = 0.17677669529663687f;
  const int NS = la_nsplit(N);
  float* dpart = scratch;
  hipLaunchKernelGGL(HIP_KERNEL_NAME(la_nreduce_kernel<1>), dim3(B * heads, NS), dim3(256), 0, st, qkv, qstat, dA, ctx, dpart, N, heads,
                     scale, NS == 1 ? dctx : nullptr, rowdot, 1.f);
  PIDM_CHECK_LAUNCH("la_dctx");
  if (NS > 1) {
    hipLaunchKernelGGL(HIP_KERNEL_NAME(la_nreduce_final_kernel<1>), dim3(B * heads), dim3(256), 0, st, dpart, ctx, dctx, rowdot, NS, 1.f);
    PIDM_CHECK_LAUNCH("la_dctx_final");
  }
  if (N % 32 == 0 && N >= 64) {        // whole 32-pixel waves of one image per block: 4 waves, or fewer while that leaves CUs idle
    const size_t npix = (size_t)B * N;
    int ppb = N % 128 == 0 ? 128 : (N % 64 == 0 ? 64 : 32);
    (void)npix;
    if (const char* pe = knob("PIDM_LA_PPB")) { const int v = atoi(pe); if ((v == 32 || v == 64 || v == 128) && N % v == 0) ppb = v; }
    PIDM_PROF_NAME("la_bwd_pix_mfma_kernel");
    hipLaunchKernelGGL(la_bwd_pix_mfma_kernel, dim3((unsigned)((size_t)B * N / ppb), heads), dim3(2 * ppb), 0, st, qkv, kstat, qstat, ctx,
                       dctx, rowdot, dA, dqkv, N, heads, scale, ppb);
  } else {
    hipLaunchKernelGGL(la_bwd_pix_kernel, dim3(cdiv(N, 256), B * heads), dim3(256), 0, st, qkv, kstat, qstat, ctx, dctx, rowdot,
                       dA, dqkv, N, heads, scale);
  }
  PIDM_CHECK_LAUNCH("la_bwd_pix_kernel");
  return 0;
}

// backward of attention + to_out projection; dy [B][N][Cout] (leading dimension ld_dy), w_out [Cout][HD] (reference layout);
// dwpart [B][Cout][HD]: per-image shares of the to_out weight gradient.  Eligibility: la_fused_ok.
bool la_fused_ok(int N, int heads, int Cout, int ld_dy) {
  const bool off = knob("PIDM_NO_LA_FUSED") != nullptr;
  return !off && N % 128 == 0 && (Cout == 32 || Cout == 64 || Cout == 128) && (ld_dy & 3) == 0 && heads >= 1 && heads <= 14;
}
// forward: k statistics, context, then attention output x projection (+ bias + residual) in one kernel; y [B][N][Cout]
int launch_la_forward_fused(const float* qkv, float* kstat, float* ctx, float* qstat, const float* w_out, const float* bias,
                            const float* resid, float* y, int Cout, int B, int N, int heads, float* scratch, hipStream_t st) {
  if (!la_fused_ok(N, heads, Cout, Cout)) return fail("fused attention forward: N=%d Cout=%d not eligible", N, Cout);
  const int HD = heads * DH;
  const float scale = 0.17677669529663687f;
  const int nseg = la_kseg(N), NS = la_nsplit(N);
  float* kpart = scratch;
  float* dpart = scratch + (size_t)B * nseg * HD * 2;
  hipLaunchKernelGGL(la_kstats_kernel, dim3(cdiv(HD, 64), B, nseg), dim3(256), 0, st, qkv, N, HD, nseg, kpart, nseg == 1 ? kstat : nullptr);
  PIDM_CHECK_LAUNCH("la_kstats_kernel");
  if (nseg > 1) {
    hipLaunchKernelGGL(la_kstats_final_kernel, dim3(cdiv(B * HD, 256)), dim3(256), 0, st, kpart, B, HD, nseg, kstat);
    PIDM_CHECK_LAUNCH("la_kstats_final_kernel");
  }
  hipLaunchKernelGGL(HIP_KERNEL_NAME(la_nreduce_kernel<0>), dim3(B * heads, NS), dim3(256), 0, st, qkv, kstat, nullptr, nullptr,
                     dpart, N, heads, scale, NS == 1 ? ctx : nullptr, nullptr, 1.f / (float)N);
  PIDM_CHECK_LAUNCH("la_context");
  if (NS > 1) {
    hipLaunchKernelGGL(HIP_KERNEL_NAME(la_nreduce_final_kernel<0>), dim3(B * heads), dim3(256), 0, st, dpart, nullptr, ctx, nullptr, NS,
                       1.f / (float)N);
    PIDM_CHECK_LAUNCH("la_context_final");
  }
  float* wt = dpart + (size_t)B * heads * NS * 1024;
  hipLaunchKernelGGL(la_wt_kernel, dim3(cdiv(Cout * HD, 256)), dim3(256), 0, st, w_out, wt, Cout, HD);
  PIDM_CHECK_LAUNCH("la_wt_kernel");
  const dim3 grid((unsigned)((size_t)B * N / 128));
  if (Cout == 32)
    hipLaunchKernelGGL(HIP_KERNEL_NAME(la_out_proj_kernel<1>), grid, dim3(256), 0, st, qkv, ctx, wt, bias, resid, y, qstat, N, heads, scale);
  else if (Cout == 64)
    hipLaunchKernelGGL(HIP_KERNEL_NAME(la_out_proj_kernel<2>), grid, dim3(256), 0, st, qkv, ctx, wt, bias, resid, y, qstat, N, heads, scale);
  else
    hipLaunchKernelGGL(HIP_KERNEL_NAME(la_out_proj_kernel<4>), grid, dim3(256), 0, st, qkv, ctx, wt, bias, resid, y, qstat, N, heads, scale);
  PIDM_CHECK_LAUNCH("la_out_proj_kernel");
  return 0;
}

// the engine's choice: the fused per-pixel kernels do the projection's work per 128-pixel workgroup, which pays once the grid
// fills the chip (>= one workgroup per CU); below that (16x16 level at batch 64: 128 workgroups) the separate projection convs
// with their own tiling win by ~30 %.  PIDM_LA_FUSED_MIN_WGS overrides (tests: 1).
bool la_fused_pays(int B, int N) {
  const char* mw = knob("PIDM_LA_FUSED_MIN_WGS");
  const long min_wgs = mw ? atol(mw) : 256;
  return (long)B * N / 128 >= min_wgs;
}
size_t la_fused_scratch_floats(int B, int N, int heads, int Cout) {
  return (size_t)B * heads * la_nsplit(N) * (Cout / 32) * 1024 + 64;
}
int launch_la_backward_fused(const float* qkv, const float* kstat, const float* qstat, const float* ctx, const float* dy, int ld_dy,
                             const float* w_out, int Cout, float* dctx, float* rowdot, float* dqkv, float* dwpart, int B, int N,
                             int heads, float* scratch, hipStream_t st) {
  if (!la_fused_ok(N, heads, Cout, ld_dy)) return fail("fused attention backward: N=%d Cout=%d not eligible", N, Cout);
  const float scale = 0.17677669529663687f;
  const int NS = la_nsplit(N);
  float* gpart = scratch;
  const dim3 grid(B * heads, NS);
#define PIDM_LA_G(NT_)                                                                                                        \
  {                                                                                                                           \
    hipLaunchKernelGGL(HIP_KERNEL_NAME(la_g_kernel<NT_>), grid, dim3(256), 0, st, qkv, qstat, dy, ld_dy, gpart, N, heads, scale); \
    PIDM_CHECK_LAUNCH("la_g_kernel");                                                                                         \
    hipLaunchKernelGGL(HIP_KERNEL_NAME(la_g_final_kernel<NT_>), dim3(B * heads), dim3(256), 0, st, gpart, ctx, w_out, dctx,    \
                       rowdot, dwpart, NS, heads);                                                                            \
    PIDM_CHECK_LAUNCH("la_g_final_kernel");                                                                                   \
  }
  if (Cout == 32) PIDM_LA_G(1)
  else if (Cout == 64) PIDM_LA_G(2)
  else PIDM_LA_G(4)
#undef PIDM_LA_G
  const dim3 gridp((unsigned)((size_t)B * N / 128));
  static bool attr_done = false;
  if (!attr_done) {   // more than 8 heads: the staged matrices + transposition tiles pass the 64 KiB default
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&la_bwd_dq_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&la_bwd_dkdv_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
    attr_done = true;
  }
  hipLaunchKernelGGL(la_bwd_dq_kernel, gridp, dim3(256), ((size_t)heads * 1056 + 4 * 32 * kLaTileLd) * sizeof(float), st, qkv, qstat, ctx, dy, ld_dy, w_out, Cout,
                     dqkv, N, heads, scale);
  PIDM_CHECK_LAUNCH("la_bwd_dq_kernel");
  hipLaunchKernelGGL(la_bwd_dkdv_kernel, gridp, dim3(256), ((size_t)heads * kLaDkvLds + 4 * 32 * kLaTileLd) * sizeof(float), st, qkv, kstat, dctx, rowdot, dqkv, N,
                     heads);
  PIDM_CHECK_LAUNCH("la_bwd_dkdv_kernel");
  return 0;
}

int launch_mid_attn(const float* qkv, const float* dO, float* out, int B, int N, int heads, bool bwd, hipStream_t st) {
  if (N > 64) return fail("mid attention: %d tokens > 64", N);
  const float scale = 0.17677669529663687f;
  const size_t lds = ((size_t)4 * N * 33 + (size_t)2 * N * (N + 1)) * sizeof(float);
  static bool attr_done = false;
  if (!attr_done) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&mid_attn_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&mid_attn_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
    attr_done = true;
  }
  const bool mfma_off = [] { const char* e = knob("PIDM_MID_ATTN_MFMA"); return e && !atoi(e); }();
  if (N == 64 && !mfma_off) {
    static bool attr64 = false;
    if (!attr64) {
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&mid_attn64_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&mid_attn64_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
      attr64 = true;
    }
    if (bwd) hipLaunchKernelGGL(HIP_KERNEL_NAME(mid_attn64_kernel<true>), dim3(B * heads), dim3(256), lds, st, qkv, dO, out, heads, scale);
    else hipLaunchKernelGGL(HIP_KERNEL_NAME(mid_attn64_kernel<false>), dim3(B * heads), dim3(256), lds, st, qkv, dO, out, heads, scale);
    PIDM_CHECK_LAUNCH("mid_attn64_kernel");
    return 0;
  }
  if (bwd)
    hipLaunchKernelGGL(HIP_KERNEL_NAME(mid_attn_kernel<true>), dim3(B * heads), dim3(256), lds, st, qkv, dO, out, N, heads, scale);
  else
    hipLaunchKernelGGL(HIP_KERNEL_NAME(mid_attn_kernel<false>), dim3(B * heads), dim3(256), lds, st, qkv, dO, out, N, heads, scale);
  PIDM_CHECK_LAUNCH("mid_attn_kernel");
  return 0;
}

}  // namespace pidm

// ---- unit-level C ABI (include/pidm.h): workspace = [dctx | rowdot | kernel scratch] ---------------------------------
extern "C" size_t pidm_linear_attention_ws(int B, int N, int heads) {
  return ((size_t)B * heads * (1024 + 32) + pidm::la_scratch_floats(B, N, heads)) * sizeof(float);
}
extern "C" int pidm_linear_attention_forward(const float* qkv, float* out, float* kstat, float* ctx, float* qstat, int B, int N,
                                             int heads, void* workspace, void* stream) {
  if (!qkv || !out || !kstat || !ctx || !qstat || !workspace || B < 1 || N < 1 || heads < 1)
    return pidm::fail("linear attention: bad arguments");
  float* scratch = reinterpret_cast<float*>(workspace) + (size_t)B * heads * (1024 + 32);
  return pidm::launch_la_forward(qkv, kstat, ctx, out, qstat, B, N, heads, scratch, reinterpret_cast<hipStream_t>(stream));
}
extern "C" int pidm_linear_attention_backward(const float* qkv, const float* kstat, const float* qstat, const float* ctx,
                                              const float* d_out, float* dqkv, int B, int N, int heads, void* workspace,
                                              void* stream) {
  if (!qkv || !kstat || !qstat || !ctx || !d_out || !dqkv || !workspace || B < 1 || N < 1 || heads < 1)
    return pidm::fail("linear attention backward: bad arguments");
  float* dctx = reinterpret_cast<float*>(workspace);
  float* rowdot = dctx + (size_t)B * heads * 1024;
  float* scratch = rowdot + (size_t)B * heads * 32;
  return pidm::launch_la_backward(qkv, kstat, qstat, ctx, d_out, dctx, rowdot, dqkv, B, N, heads, scratch,
                                  reinterpret_cast<hipStream_t>(stream));
}
extern "C" size_t pidm_linear_attention_out_backward_ws(int B, int N, int heads, int Cout) {
  return ((size_t)B * heads * (1024 + 32) + (size_t)B * Cout * heads * 32 + pidm::la_fused_scratch_floats(B, N, heads, Cout)) *
         sizeof(float);
}
extern "C" int pidm_linear_attention_out_backward(const float* qkv, const float* kstat, const float* qstat, const float* ctx,
                                                  const float* d_y, int ld_dy, const float* w_out, int Cout, float* dqkv,
                                                  float* dw_out, int B, int N, int heads, void* workspace, void* stream) {
  if (!qkv || !kstat || !qstat || !ctx || !d_y || !w_out || !dqkv || !dw_out || !workspace || B < 1)
    return pidm::fail("fused attention backward: bad arguments");
  const int HD = heads * 32;
  float* dctx = reinterpret_cast<float*>(workspace);
  float* rowdot = dctx + (size_t)B * heads * 1024;
  float* dwpart = rowdot + (size_t)B * heads * 32;
  float* scratch = dwpart + (size_t)B * Cout * HD;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  if (pidm::launch_la_backward_fused(qkv, kstat, qstat, ctx, d_y, ld_dy, w_out, Cout, dctx, rowdot, dqkv, dwpart, B, N, heads,
                                     scratch, st))
    return -1;
  return pidm::launch_split_reduce(dwpart, dw_out, nullptr, nullptr, B, Cout, HD, 1, Cout, HD, st);
}
extern "C" int pidm_linear_attention_out_forward(const float* qkv, const float* w_out, const float* bias, const float* residual,
                                                 float* y, int Cout, float* kstat, float* ctx, float* qstat, int B, int N,
                                                 int heads, void* workspace, void* stream) {
  if (!qkv || !w_out || !y || !kstat || !ctx || !qstat || !workspace || B < 1) return pidm::fail("fused attention forward: bad arguments");
  float* scratch = reinterpret_cast<float*>(workspace) + (size_t)B * heads * (1024 + 32);
  return pidm::launch_la_forward_fused(qkv, kstat, ctx, qstat, w_out, bias, residual, y, Cout, B, N, heads, scratch,
                                       reinterpret_cast<hipStream_t>(stream));
}
