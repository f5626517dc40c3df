// Linear-elasticity residual  r = K(rho) u - f  for topology optimisation, MATRIX-FREE, forward + adjoint (gfx950).
//
// Replaces ResidualsMechanics.compute_residual src/residuals_mechanics_K.py:198-274: the reference assembles a
// DENSE 8450 x 8450 stiffness matrix per sample with index_put(accumulate=True) (285.6 MB/sample, ~2.5 GB of HBM
// traffic) and multiplies it with u.  Here every dof gathers its <= 4 incident elements in a fixed order
// (deterministic, no atomics): (K u)_i = sum_{(e,a): D[e][a]=i} rho_e sum_b kloc[e][a][b] u[D[e][b]].
// One workgroup per sample: the bilinearly up-sampled displacement field (64x64 -> 65x65, torchvision Resize with
// antialias=False == F.interpolate(align_corners=False)) and rho live in LDS (34 KB + 16 KB), HBM traffic is the
// compulsory 48 KB in + ~85 KB out per sample.  Dirichlet rows (mask != 0) are identity rows with f -> 0, only rows
// are replaced (K is not symmetrised), exactly as the reference does (:226-240).
#include "pidm_launch.h"

namespace pidm {

struct MechMesh {
  const int* elem_dofs;   // [E][8]
  const int* dof_elems;   // [ndof][4][2] = (element, local index) or (-1, -1)
  const float* kloc;      // [E][8][8] or [1][8][8] when kloc_stride == 0
  int kloc_stride;        // 0 (uniform mesh) or 64
  int E, ndof, nel, nn;   // nel = 64 elements per side, nn = 65 nodes per side
};

// source index / weight of torch's bilinear, align_corners=False, for output coordinate o
__device__ __forceinline__ void bil_src(int o, int n_in, int n_out, int* i0, int* i1, float* lam) {
  float src = ((float)o + 0.5f) * ((float)n_in / (float)n_out) - 0.5f;
  if (src < 0.f) src = 0.f;
  int a = (int)src;
  if (a > n_in - 1) a = n_in - 1;
  *i0 = a;
  *i1 = (a < n_in - 1) ? a + 1 : a;
  *lam = src - (float)a;
}

// generic square bilinear resize of [BC][Hi][Hi] -> [BC][Ho][Ho] (forward only: used on network INPUTS)
__global__ void bilinear_resize_kernel(const float* __restrict__ in, float* __restrict__ out, int BC, int Hi, int Ho) {
  const size_t total = (size_t)BC * Ho * Ho;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int ox = (int)(i % Ho), oy = (int)((i / Ho) % Ho);
    const size_t bc = i / ((size_t)Ho * Ho);
    int y0, y1, x0, x1;
    float ly, lx;
    bil_src(oy, Hi, Ho, &y0, &y1, &ly);
    bil_src(ox, Hi, Ho, &x0, &x1, &lx);
    const float* p = in + bc * Hi * Hi;
    const float top = p[y0 * Hi + x0] * (1.f - lx) + p[y0 * Hi + x1] * lx;
    const float bot = p[y1 * Hi + x0] * (1.f - lx) + p[y1 * Hi + x1] * lx;
    out[i] = top * (1.f - ly) + bot * ly;
  }
}

// LDS layout: U[ndof] | rho[E] | (bwd) Z[ndof] | GU[ndof]
template <bool BWD>
__global__ void __launch_bounds__(256) mech_kernel(MechMesh ms, const float* __restrict__ x0,   // [B,3,nel,nel] NCHW
                                                   const float* __restrict__ bcs,               // [B,4,nn,nn] (bc_x, bc_y, load_x, load_y)
                                                   const float* __restrict__ vf,                // [B]
                                                   float* __restrict__ residual,                // FWD out [B,ndof]
                                                   float* __restrict__ model_out,               // FWD out [B,3,nn,nn]
                                                   float* __restrict__ comp_shift,              // FWD out [B][2]
                                                   const float* __restrict__ g_res,             // BWD in  [B,ndof]
                                                   const float* __restrict__ g_mo,              // BWD in  [B,3,nn,nn] (may be null)
                                                   const float* __restrict__ g_cs,              // BWD in  [B][2] (d/d compliance, d/d shift)
                                                   float* __restrict__ g_x0) {                  // BWD out [B,3,nel,nel]
  HIP_DYNAMIC_SHARED(float, smem)
  __shared__ double red[2][4];
  const int b = blockIdx.x, tid = threadIdx.x;
  const int nel = ms.nel, nn = ms.nn, E = ms.E, ndof = ms.ndof;
  float* sU = smem;
  float* sR = smem + ndof;
  float* sZ = sR + E;
  float* sG = sZ + ndof;
  const float* xb = x0 + (size_t)b * 3 * nel * nel;
  const float* bb = bcs + (size_t)b * 4 * nn * nn;
  // ---- phase 1: U = bilinear(u, 65), rho -> LDS (+ model_out) ----
  for (int node = tid; node < nn * nn; node += 256) {
    const int r = node / nn, c = node - r * nn;
    int y0, y1, x0i, x1i;
    float ly, lx;
    bil_src(r, nel, nn, &y0, &y1, &ly);
    bil_src(c, nel, nn, &x0i, &x1i, &lx);
#pragma unroll
    for (int d = 0; d < 2; ++d) {
      const float* p = xb + (size_t)d * nel * nel;
      const float top = p[y0 * nel + x0i] * (1.f - lx) + p[y0 * nel + x1i] * lx;
      const float bot = p[y1 * nel + x0i] * (1.f - lx) + p[y1 * nel + x1i] * lx;
      const float u = top * (1.f - ly) + bot * ly;
      sU[2 * node + d] = u;
      if (!BWD) model_out[((size_t)b * 3 + d) * nn * nn + node] = u;
    }
    if (!BWD) model_out[((size_t)b * 3 + 2) * nn * nn + node] = (r < nel && c < nel) ? xb[(size_t)2 * nel * nel + r * nel + c] : 0.f;
  }
  double rsum = 0.0;
  for (int e = tid; e < E; e += 256) {
    const float rv = xb[(size_t)2 * nel * nel + e];
    sR[e] = rv;
    rsum += rv;
  }
  __syncthreads();
  // ---- phase 2: per dof i: KU_i, masked row, residual / (bwd) z_i ----
  const float gc = BWD ? g_cs[2 * b] : 0.f;
  double csum = 0.0;
  for (int i = tid; i < ndof; i += 256) {
    float ku = 0.f;
    for (int s = 0; s < 4; ++s) {
      const int e = ms.dof_elems[(i * 4 + s) * 2], a = ms.dof_elems[(i * 4 + s) * 2 + 1];
      if (e < 0) continue;
      const float* k = ms.kloc + (size_t)e * ms.kloc_stride + a * 8;
      const int* D = ms.elem_dofs + (size_t)e * 8;
      float acc = 0.f;
#pragma unroll
      for (int q = 0; q < 8; ++q) acc = fmaf(k[q], sU[D[q]], acc);
      ku = fmaf(sR[e], acc, ku);
    }
    const int node = i >> 1, d = i & 1;
    const bool masked = bb[(size_t)d * nn * nn + node] != 0.f;
    const float ui = sU[i];
    const float kbc = masked ? ui : ku;
    if (!BWD) {
      const float f = masked ? 0.f : bb[(size_t)(2 + d) * nn * nn + node];
      residual[(size_t)b * ndof + i] = kbc - f;
      csum += (double)(ui * kbc);
    } else {
      const float w = g_res[(size_t)b * ndof + i] + gc * ui;   // d L / d Kbc_i
      sZ[i] = masked ? 0.f : w;                                // d L / d KU_i
      float gu = gc * kbc + (masked ? w : 0.f);                // direct d L / d U_i
      if (g_mo) gu += g_mo[((size_t)b * 3 + d) * nn * nn + node];
      sG[i] = gu;
    }
  }
  if (!BWD) {
    // compliance and volume shift
    for (int off = 32; off > 0; off >>= 1) {
      csum += __shfl_down(csum, off);
      rsum += __shfl_down(rsum, off);
    }
    if ((tid & 63) == 0) {
      red[0][tid >> 6] = csum;
      red[1][tid >> 6] = rsum;
    }
    __syncthreads();
    if (tid == 0) {
      comp_shift[2 * b] = (float)(red[0][0] + red[0][1] + red[0][2] + red[0][3]);
      comp_shift[2 * b + 1] = (float)((red[1][0] + red[1][1] + red[1][2] + red[1][3]) / E) - vf[b];
    }
    return;
  }
  __syncthreads();
  // ---- phase 3 (bwd): gU += K^T z (gather), g_rho_e = z_e^T kloc u_e + g_shift/E + g_mo[rho] ----
  for (int i = tid; i < ndof; i += 256) {
    float acc = 0.f;
    for (int s = 0; s < 4; ++s) {
      const int e = ms.dof_elems[(i * 4 + s) * 2], bq = ms.dof_elems[(i * 4 + s) * 2 + 1];
      if (e < 0) continue;
      const float* k = ms.kloc + (size_t)e * ms.kloc_stride;
      const int* D = ms.elem_dofs + (size_t)e * 8;
      float t = 0.f;
#pragma unroll
      for (int a = 0; a < 8; ++a) t = fmaf(k[a * 8 + bq], sZ[D[a]], t);   // column bq of kloc
      acc = fmaf(sR[e], t, acc);
    }
    sG[i] += acc;
  }
  const float gs = g_cs[2 * b + 1] / (float)E;
  float* gx = g_x0 + (size_t)b * 3 * nel * nel;
  for (int e = tid; e < E; e += 256) {
    const float* k = ms.kloc + (size_t)e * ms.kloc_stride;
    const int* D = ms.elem_dofs + (size_t)e * 8;
    float acc = 0.f;
#pragma unroll
    for (int a = 0; a < 8; ++a) {
      float t = 0.f;
#pragma unroll
      for (int q = 0; q < 8; ++q) t = fmaf(k[a * 8 + q], sU[D[q]], t);
      acc = fmaf(sZ[D[a]], t, acc);
    }
    float g = acc + gs;
    if (g_mo) {
      const int r = e / nel, c = e - r * nel;
      g += g_mo[((size_t)b * 3 + 2) * nn * nn + r * nn + c];
    }
    gx[(size_t)2 * nel * nel + e] = g;
  }
  __syncthreads();
  // ---- phase 4 (bwd): adjoint of the bilinear up-sampling: each 64x64 pixel gathers the <= 3x3 nodes that read it ----
  for (int pix = tid; pix < nel * nel; pix += 256) {
    const int y = pix / nel, x = pix - y * nel;
    float g0 = 0.f, g1 = 0.f;
    for (int r = y - 1; r <= y + 2; ++r) {
      if (r < 0 || r >= nn) continue;
      int y0, y1;
      float ly;
      bil_src(r, nel, nn, &y0, &y1, &ly);
      const float wy = (y0 == y ? 1.f - ly : 0.f) + (y1 == y ? ly : 0.f);
      if (wy == 0.f) continue;
      for (int c = x - 1; c <= x + 2; ++c) {
        if (c < 0 || c >= nn) continue;
        int x0i, x1i;
        float lx;
        bil_src(c, nel, nn, &x0i, &x1i, &lx);
        const float wx = (x0i == x ? 1.f - lx : 0.f) + (x1i == x ? lx : 0.f);
        if (wx == 0.f) continue;
        const int node = r * nn + c;
        g0 = fmaf(wy * wx, sG[2 * node], g0);
        g1 = fmaf(wy * wx, sG[2 * node + 1], g1);
      }
    }
    gx[pix] = g0;
    gx[(size_t)nel * nel + pix] = g1;
  }
}


// =====================================================================================================================
// Topology-optimisation EVALUATION block (reference src/residuals_mechanics_K.py:276-347,369-380):
//   * mech_apply_kernel: r = K_closed(rho) u - f and c = u.f for nodal displacement IMAGES (the data fields)
//   * mech_pcg_kernel:   u = K_closed(rho_bin)^-1 f  - the reference calls torch.linalg.solve on the dense 8450^2 matrix
//                        per sample; here a Jacobi-preconditioned conjugate gradient on the matrix-free operator, one
//                        workgroup per sample, fp64 vectors (the search direction lives in LDS), compliance c = f.u
//   * floating_material_kernel: number of 8-connected foreground components (replaces cv2.connectedComponents)
// With Dirichlet dofs pinned (identity rows, f = 0 there) the iterates keep u_D = 0, so the row-replaced operator acts
// as the symmetric positive definite K_FF on the free dofs.
// =====================================================================================================================
__device__ __forceinline__ float rho_eff(float r, float thr, float hi, float lo) { return thr < 0.f ? r : (r > thr ? hi : lo); }

__device__ __forceinline__ double block_sum(double v, double* red4) {   // all 256 threads; result broadcast
  for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red4[threadIdx.x >> 6] = v;
  __syncthreads();
  return (red4[0] + red4[1]) + (red4[2] + red4[3]);
}

// =====================================================================================================================
// Fused mechanics training loss (src/denoising_utils.py:666-708 with gov_eqs == 'mechanics') and its gradient with respect
// to the network output, one workgroup per sample, nothing but the 5 partial sums and the gradient leave the chip:
//   data  = c_data * mean_b( p2w_b * mean((x_0 - model_out)^2) )           model_out = (U, rho zero-padded)  [3,nn,nn]
//   res   = mean( c_res * 0.5 * r^2 / var_b ),  r = K_closed(rho) U - f    [B, ndof]
//   ineq  = mean_{i,j}( c_ineq * 0.5 * shift_j^2 / var_i )                 (the reference's [B] / [B,1] broadcast, :697)
//   opt   = mean_b( lambda_opt * compliance_b ),  compliance = U . K_closed U
// The upstream gradients are closed-form in the forward quantities, so forward, loss derivative and adjoint run back to back
// on the LDS-resident U / rho: g_res = c_res r inv_var_b / (B ndof), g_model_out = 2 c_data p2w_b (model_out - x_0) / (3 nn^2 B),
// g_compliance = lambda_opt / B, g_shift = c_ineq shift_b (sum_i inv_var_i) / B^2.
// LDS: U[ndof] | rho[E] | Z[ndof] | GU[ndof] (the layout of mech_kernel<true>).
// =====================================================================================================================
__global__ void __launch_bounds__(256) mech_loss_kernel(MechMesh ms, const float* __restrict__ x0,        // [B,3,nel,nel] network output
                                                        const float* __restrict__ target,    // [B,3,nn,nn] data fields x_0
                                                        const float* __restrict__ bcs,       // [B,4,nn,nn]
                                                        const float* __restrict__ vf,        // [B]
                                                        const float* __restrict__ p2w,       // [B]
                                                        const float* __restrict__ inv_var,   // [B]
                                                        const float* __restrict__ ivs_dev,   // null, or 1 float: sum_i inv_var_i to use
                                                        float c_data, float c_res, float c_ineq, float lambda_opt, int B,
                                                        float* __restrict__ g_x0,            // [B,3,nel,nel]
                                                        double* __restrict__ partial) {      // [B][8]
  HIP_DYNAMIC_SHARED(float, smem)
  __shared__ double red4[4];
  const int b = blockIdx.x, tid = threadIdx.x;
  const int nel = ms.nel, nn = ms.nn, E = ms.E, ndof = ms.ndof;
  float* sU = smem;
  float* sR = smem + ndof;
  float* sZ = sR + E;
  float* sG = sZ + ndof;
  const float* xb = x0 + (size_t)b * 3 * nel * nel;
  const float* bb = bcs + (size_t)b * 4 * nn * nn;
  const float* tb = target + (size_t)b * 3 * nn * nn;
  const float dscale = 2.f * c_data * p2w[b] / ((float)B * 3.f * (float)(nn * nn));
  // ---- phase 1: U = bilinear(u, nn), rho -> LDS; data term and its gradient at the nodes ----
  double a_data = 0.0;
  for (int node = tid; node < nn * nn; node += 256) {
    const int r = node / nn, c = node - r * nn;
    int y0, y1, x0i, x1i;
    float ly, lx;
    bil_src(r, nel, nn, &y0, &y1, &ly);
    bil_src(c, nel, nn, &x0i, &x1i, &lx);
#pragma unroll
    for (int d = 0; d < 2; ++d) {
      const float* p = xb + (size_t)d * nel * nel;
      const float top = p[y0 * nel + x0i] * (1.f - lx) + p[y0 * nel + x1i] * lx;
      const float bot = p[y1 * nel + x0i] * (1.f - lx) + p[y1 * nel + x1i] * lx;
      const float u = top * (1.f - ly) + bot * ly;
      sU[2 * node + d] = u;
      const float diff = u - tb[(size_t)d * nn * nn + node];
      a_data += (double)(diff * diff);
      sG[2 * node + d] = dscale * diff;
    }
    const float rho_pad = (r < nel && c < nel) ? xb[(size_t)2 * nel * nel + r * nel + c] : 0.f;
    const float d2 = rho_pad - tb[(size_t)2 * nn * nn + node];
    a_data += (double)(d2 * d2);
  }
  double rsum_l = 0.0;
  for (int e = tid; e < E; e += 256) {
    const float rv = xb[(size_t)2 * nel * nel + e];
    sR[e] = rv;
    rsum_l += rv;
  }
  double ivs_l = 0.0;
  for (int i = tid; i < B; i += 256) ivs_l += (double)inv_var[i];
  const double rsum = block_sum(rsum_l, red4);      // (publishes sU / sR / sG as well)
  // data parallel: the caller passes (sum over ALL ranks' samples of inv_var) / world instead of the local sum, which makes the
  // rank-averaged loss and gradient of the [B,B] term equal to the single-process global-batch values
  const double ivs = ivs_dev ? (double)ivs_dev[0] : block_sum(ivs_l, red4);
  const float shift = (float)(rsum / E) - vf[b];
  const float gc = lambda_opt / (float)B;                                             // d loss / d compliance_b
  const float gs = (c_ineq > 0.f ? c_ineq * shift * (float)ivs / ((float)B * (float)B) : 0.f) / (float)E;   // d loss / d rho_e via the shift
  const float rscale = c_res * inv_var[b] / ((float)B * (float)ndof);
  // ---- phase 2: per dof: (K_closed U)_i, residual, compliance; d loss / d (K U)_i and the direct d loss / d U_i ----
  double a_r2 = 0.0, a_rabs = 0.0, a_comp = 0.0;
  for (int i = tid; i < ndof; i += 256) {
    float ku = 0.f;
    for (int s = 0; s < 4; ++s) {
      const int e = ms.dof_elems[(i * 4 + s) * 2], a = ms.dof_elems[(i * 4 + s) * 2 + 1];
      if (e < 0) continue;
      const float* k = ms.kloc + (size_t)e * ms.kloc_stride + a * 8;
      const int* D = ms.elem_dofs + (size_t)e * 8;
      float acc = 0.f;
#pragma unroll
      for (int q = 0; q < 8; ++q) acc = fmaf(k[q], sU[D[q]], acc);
      ku = fmaf(sR[e], acc, ku);
    }
    const int node = i >> 1, d = i & 1;
    const bool masked = bb[(size_t)d * nn * nn + node] != 0.f;
    const float ui = sU[i];
    const float kbc = masked ? ui : ku;
    const float f = masked ? 0.f : bb[(size_t)(2 + d) * nn * nn + node];
    const float res = kbc - f;
    a_r2 += (double)(res * res);
    a_rabs += (double)fabsf(res);
    a_comp += (double)(ui * kbc);
    const float w = rscale * res + gc * ui;          // d L / d Kbc_i
    sZ[i] = masked ? 0.f : w;                        // d L / d KU_i
    sG[i] += gc * kbc + (masked ? w : 0.f);          // direct d L / d U_i (on top of the data term)
  }
  __syncthreads();
  // ---- phase 3: gU += K^T z (gather); g_rho_e = z_e^T kloc u_e + shift term + data term ----
  for (int i = tid; i < ndof; i += 256) {
    float acc = 0.f;
    for (int s = 0; s < 4; ++s) {
      const int e = ms.dof_elems[(i * 4 + s) * 2], bq = ms.dof_elems[(i * 4 + s) * 2 + 1];
      if (e < 0) continue;
      const float* k = ms.kloc + (size_t)e * ms.kloc_stride;
      const int* D = ms.elem_dofs + (size_t)e * 8;
      float t = 0.f;
#pragma unroll
      for (int a = 0; a < 8; ++a) t = fmaf(k[a * 8 + bq], sZ[D[a]], t);
      acc = fmaf(sR[e], t, acc);
    }
    sG[i] += acc;
  }
  float* gx = g_x0 + (size_t)b * 3 * nel * nel;
  for (int e = tid; e < E; e += 256) {
    const float* k = ms.kloc + (size_t)e * ms.kloc_stride;
    const int* D = ms.elem_dofs + (size_t)e * 8;
    float acc = 0.f;
#pragma unroll
    for (int a = 0; a < 8; ++a) {
      float t = 0.f;
#pragma unroll
      for (int q = 0; q < 8; ++q) t = fmaf(k[a * 8 + q], sU[D[q]], t);
      acc = fmaf(sZ[D[a]], t, acc);
    }
    const int r = e / nel, c = e - r * nel;
    gx[(size_t)2 * nel * nel + e] = acc + gs + dscale * (sR[e] - tb[(size_t)2 * nn * nn + r * nn + c]);
  }
  __syncthreads();
  // ---- phase 4: adjoint of the bilinear up-sampling ----
  for (int pix = tid; pix < nel * nel; pix += 256) {
    const int y = pix / nel, x = pix - y * nel;
    float g0 = 0.f, g1 = 0.f;
    for (int r = y - 1; r <= y + 2; ++r) {
      if (r < 0 || r >= nn) continue;
      int y0, y1;
      float ly;
      bil_src(r, nel, nn, &y0, &y1, &ly);
      const float wy = (y0 == y ? 1.f - ly : 0.f) + (y1 == y ? ly : 0.f);
      if (wy == 0.f) continue;
      for (int c = x - 1; c <= x + 2; ++c) {
        if (c < 0 || c >= nn) continue;
        int x0i, x1i;
        float lx;
        bil_src(c, nel, nn, &x0i, &x1i, &lx);
        const float wx = (x0i == x ? 1.f - lx : 0.f) + (x1i == x ? lx : 0.f);
        if (wx == 0.f) continue;
        const int node = r * nn + c;
        g0 = fmaf(wy * wx, sG[2 * node], g0);
        g1 = fmaf(wy * wx, sG[2 * node + 1], g1);
      }
    }
    gx[pix] = g0;
    gx[(size_t)nel * nel + pix] = g1;
  }
  // ---- per-sample partial sums (fixed order) ----
  const double s_data = block_sum(a_data, red4), s_r2 = block_sum(a_r2, red4), s_rabs = block_sum(a_rabs, red4),
               s_comp = block_sum(a_comp, red4);
  if (tid == 0) {
    double* o = partial + (size_t)b * 8;
    o[0] = s_data; o[1] = s_r2; o[2] = s_rabs; o[3] = s_comp; o[4] = (double)shift;
  }
}

// out[0] = loss, out[1] = data loss, out[2] = mean |r|, out[3] = mean shift (0 unless c_ineq > 0), out[4] = mean compliance
__global__ void mech_loss_finalize(const double* __restrict__ partial, const float* __restrict__ p2w, const float* __restrict__ inv_var,
                                   const float* __restrict__ ivs_dev, float c_data, float c_res, float c_ineq, float lambda_opt, int B, int nn, int ndof,
                                   float* __restrict__ out) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  double data = 0.0, res = 0.0, rabs = 0.0, comp = 0.0, sh = 0.0, sh2 = 0.0, ivs = 0.0;
  for (int b = 0; b < B; ++b) {
    const double* p = partial + (size_t)b * 8;
    data += p[0] / (3.0 * nn * nn) * (double)p2w[b];
    res += p[1] * (double)inv_var[b];
    rabs += p[2];
    comp += p[3];
    sh += p[4];
    sh2 += p[4] * p[4];
    ivs += (double)inv_var[b];
  }
  if (ivs_dev) ivs = (double)ivs_dev[0];
  data = data / B * c_data;
  double loss = data + 0.5 * c_res * res / ((double)B * ndof) + lambda_opt * comp / B;
  if (c_ineq > 0.f) loss += 0.5 * c_ineq * ivs * sh2 / ((double)B * B);
  out[0] = (float)loss;
  out[1] = (float)data;
  out[2] = (float)(rabs / ((double)B * ndof));
  out[3] = c_ineq > 0.f ? (float)(sh / B) : 0.f;
  out[4] = (float)(comp / B);
  out[5] = out[6] = out[7] = 0.f;
}

__global__ void __launch_bounds__(256) mech_apply_kernel(MechMesh ms, const float* __restrict__ rho,     // [B][E]
                                                         const float* __restrict__ u_img,   // [B][2][nn][nn]
                                                         const float* __restrict__ bcs,     // [B][4][nn][nn]
                                                         float* __restrict__ residual,      // [B][ndof]
                                                         float* __restrict__ comp_uf) {     // [B]  u . f
  HIP_DYNAMIC_SHARED(float, smem)
  __shared__ double red4[4];
  const int b = blockIdx.x, tid = threadIdx.x;
  const int nn = ms.nn, E = ms.E, ndof = ms.ndof;
  float* sU = smem;
  float* sR = smem + ndof;
  const float* bb = bcs + (size_t)b * 4 * nn * nn;
  for (int i = tid; i < ndof; i += 256) sU[i] = u_img[((size_t)b * 2 + (i & 1)) * nn * nn + (i >> 1)];
  for (int e = tid; e < E; e += 256) sR[e] = rho[(size_t)b * E + e];
  __syncthreads();
  double cs = 0.0;
  for (int i = tid; i < ndof; i += 256) {
    double ku = 0.0;
    for (int s = 0; s < 4; ++s) {
      const int e = ms.dof_elems[(i * 4 + s) * 2], a = ms.dof_elems[(i * 4 + s) * 2 + 1];
      if (e < 0) continue;
      const float* k = ms.kloc + (size_t)e * ms.kloc_stride + a * 8;
      const int* D = ms.elem_dofs + (size_t)e * 8;
      double acc = 0.0;
#pragma unroll
      for (int q = 0; q < 8; ++q) acc += (double)k[q] * (double)sU[D[q]];
      ku += (double)sR[e] * acc;
    }
    const int node = i >> 1, d = i & 1;
    const bool masked = bb[(size_t)d * nn * nn + node] != 0.f;
    const double f = masked ? 0.0 : (double)bb[(size_t)(2 + d) * nn * nn + node];
    residual[(size_t)b * ndof + i] = (float)((masked ? (double)sU[i] : ku) - f);
    cs += (double)sU[i] * f;
  }
  const double c = block_sum(cs, red4);
  if (tid == 0) comp_uf[b] = (float)c;
}

// workspace per sample: x | r | Ap | Minv  (4 * ndof doubles); LDS: p (ndof doubles) | rho_eff (E floats)
__global__ void __launch_bounds__(256) mech_pcg_kernel(MechMesh ms, const float* __restrict__ rho, const float* __restrict__ bcs,
                                                       float thr, float hi, float lo, int max_iter, double rtol,
                                                       double* __restrict__ ws, float* __restrict__ u_out,
                                                       float* __restrict__ comp_out, float* __restrict__ rho_mean_out,
                                                       int* __restrict__ iters_out, float* __restrict__ relres_out) {
  HIP_DYNAMIC_SHARED(float, smem)
  __shared__ double red4[4];
  const int b = blockIdx.x, tid = threadIdx.x;
  const int nn = ms.nn, E = ms.E, ndof = ms.ndof;
  double* sP = reinterpret_cast<double*>(smem);
  float* sR = reinterpret_cast<float*>(sP + ndof);
  double* X = ws + (size_t)b * 4 * ndof;
  double* R = X + ndof;
  double* AP = R + ndof;
  double* MI = AP + ndof;
  const float* bb = bcs + (size_t)b * 4 * nn * nn;
  double rs = 0.0;
  for (int e = tid; e < E; e += 256) {
    const float rv = rho_eff(rho[(size_t)b * E + e], thr, hi, lo);
    sR[e] = rv;
    rs += rv;
  }
  const double rsum = block_sum(rs, red4);   // (also publishes sR)
  if (tid == 0 && rho_mean_out) rho_mean_out[b] = (float)(rsum / E);
  // x = 0, r = f, Minv = 1/diag(K_closed), p = z = Minv r
  double rz_l = 0.0, rr_l = 0.0;
  for (int i = tid; i < ndof; i += 256) {
    const int node = i >> 1, d = i & 1;
    const bool masked = bb[(size_t)d * nn * nn + node] != 0.f;
    double diag = 0.0;
    for (int s = 0; s < 4; ++s) {
      const int e = ms.dof_elems[(i * 4 + s) * 2], a = ms.dof_elems[(i * 4 + s) * 2 + 1];
      if (e < 0) continue;
      diag += (double)sR[e] * (double)ms.kloc[(size_t)e * ms.kloc_stride + a * 8 + a];
    }
    const double mi = masked ? 1.0 : 1.0 / diag;
    const double f = masked ? 0.0 : (double)bb[(size_t)(2 + d) * nn * nn + node];
    X[i] = 0.0;
    R[i] = f;
    MI[i] = mi;
    const double z = mi * f;
    sP[i] = z;
    rz_l += f * z;
    rr_l += f * f;
  }
  double rz = block_sum(rz_l, red4);
  const double r0 = sqrt(block_sum(rr_l, red4));
  int it = 0;
  double rr = r0 * r0;
  if (r0 > 0.0) {
    for (it = 0; it < max_iter; ++it) {
      // Ap = K_closed p
      double pap_l = 0.0;
      for (int i = tid; i < ndof; i += 256) {
        const int node = i >> 1, d = i & 1;
        const bool masked = bb[(size_t)d * nn * nn + node] != 0.f;
        double ku = 0.0;
        if (!masked) {
          for (int s = 0; s < 4; ++s) {
            const int e = ms.dof_elems[(i * 4 + s) * 2], a = ms.dof_elems[(i * 4 + s) * 2 + 1];
            if (e < 0) continue;
            const float* k = ms.kloc + (size_t)e * ms.kloc_stride + a * 8;
            const int* D = ms.elem_dofs + (size_t)e * 8;
            double acc = 0.0;
#pragma unroll
            for (int q = 0; q < 8; ++q) acc += (double)k[q] * sP[D[q]];
            ku += (double)sR[e] * acc;
          }
        } else {
          ku = sP[i];
        }
        AP[i] = ku;
        pap_l += sP[i] * ku;
      }
      const double pap = block_sum(pap_l, red4);
      const double alpha = rz / pap;
      double rz_n = 0.0, rr_n = 0.0;
      for (int i = tid; i < ndof; i += 256) {
        X[i] += alpha * sP[i];
        const double r = R[i] - alpha * AP[i];
        R[i] = r;
        rz_n += r * MI[i] * r;
        rr_n += r * r;
      }
      const double rz_new = block_sum(rz_n, red4);
      rr = block_sum(rr_n, red4);
      if (sqrt(rr) <= rtol * r0) { ++it; break; }
      const double beta = rz_new / rz;
      rz = rz_new;
      for (int i = tid; i < ndof; i += 256) sP[i] = MI[i] * R[i] + beta * sP[i];
      __syncthreads();
    }
  }
  double c_l = 0.0;
  for (int i = tid; i < ndof; i += 256) {
    const int node = i >> 1, d = i & 1;
    const bool masked = bb[(size_t)d * nn * nn + node] != 0.f;
    const double f = masked ? 0.0 : (double)bb[(size_t)(2 + d) * nn * nn + node];
    const double x = X[i];
    if (u_out) u_out[(size_t)b * ndof + i] = (float)x;
    c_l += f * x;
  }
  const double c = block_sum(c_l, red4);
  if (tid == 0) {
    comp_out[b] = (float)c;
    if (iters_out) iters_out[b] = it;
    if (relres_out) relres_out[b] = (float)(r0 > 0.0 ? sqrt(rr) / r0 : 0.0);
  }
}

// 8-connected components of {rho > thr} by min-label propagation in LDS; ncomp[b] = number of foreground components
__global__ void __launch_bounds__(256) floating_material_kernel(const float* __restrict__ rho, float thr, int nel,
                                                                int* __restrict__ ncomp) {
  HIP_DYNAMIC_SHARED(float, smem)
  __shared__ int changed, count;
  int* lab = reinterpret_cast<int*>(smem);
  const int b = blockIdx.x, tid = threadIdx.x, E = nel * nel;
  for (int e = tid; e < E; e += 256) lab[e] = rho[(size_t)b * E + e] > thr ? e + 1 : 0;
  __syncthreads();
  for (int sweep = 0; sweep < 4 * E; ++sweep) {
    if (tid == 0) changed = 0;
    __syncthreads();
    bool any = false;
    for (int e = tid; e < E; e += 256) {
      int l = lab[e];
      if (l == 0) continue;
      const int y = e / nel, x = e - y * nel;
      int m = l;
      for (int dy = -1; dy <= 1; ++dy) {
        const int yy = y + dy;
        if (yy < 0 || yy >= nel) continue;
        for (int dx = -1; dx <= 1; ++dx) {
          const int xx = x + dx;
          if (xx < 0 || xx >= nel) continue;
          const int ln = lab[yy * nel + xx];
          if (ln != 0 && ln < m) m = ln;
        }
      }
      if (m < l) { lab[e] = m; any = true; }   // monotone decreasing labels: races only delay convergence
    }
    if (any) changed = 1;
    __syncthreads();
    if (!changed) break;
    __syncthreads();
  }
  if (tid == 0) count = 0;
  __syncthreads();
  int c = 0;
  for (int e = tid; e < E; e += 256) c += (lab[e] == e + 1) ? 1 : 0;
  for (int off = 32; off > 0; off >>= 1) c += __shfl_down(c, off);
  if ((tid & 63) == 0 && c) atomicAdd(&count, c);   // integer add: order independent
  __syncthreads();
  if (tid == 0) ncomp[b] = count;
}

}  // namespace pidm

using namespace pidm;

extern "C" int pidm_bilinear_resize(const float* in, float* out, int BC, int Hi, int Ho, void* stream) {
  size_t n = (size_t)BC * Ho * Ho;
  int blocks = (int)((n + 255) / 256);
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(bilinear_resize_kernel, dim3(blocks), dim3(256), 0, as_stream(stream), in, out, BC, Hi, Ho);
  PIDM_CHECK_LAUNCH("bilinear_resize_kernel");
  return 0;
}

static int mech_args(int nel, int E, int ndof, const void* a, const void* b2, const void* c) {
  if (!a || !b2 || !c) return fail("mech: null mesh table");
  if (E != nel * nel || ndof != 2 * (nel + 1) * (nel + 1)) return fail("mech: mesh must be nel x nel quads with all dofs free");
  return 0;
}

extern "C" int pidm_mech_residual_fwd(const float* x0_pred, const float* bcs, const float* vf, const float* kloc,
                                      int kloc_stride, const int32_t* elem_dofs, const int32_t* dof_elems, int nel,
                                      float* residual, float* model_out, float* comp_shift, int B, void* stream) {
  const int E = nel * nel, nn = nel + 1, ndof = 2 * nn * nn;
  if (mech_args(nel, E, ndof, kloc, elem_dofs, dof_elems)) return -1;
  MechMesh ms{elem_dofs, dof_elems, kloc, kloc_stride, E, ndof, nel, nn};
  const size_t lds = (size_t)(ndof + E) * sizeof(float);
  static bool attr = false;
  if (!attr) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&mech_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&mech_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
    attr = true;
  }
  hipLaunchKernelGGL(HIP_KERNEL_NAME(mech_kernel<false>), dim3(B), dim3(256), lds, as_stream(stream), ms, x0_pred, bcs, vf, residual,
                     model_out, comp_shift, nullptr, nullptr, nullptr, nullptr);
  PIDM_CHECK_LAUNCH("mech_kernel<fwd>");
  return 0;
}

extern "C" int pidm_mech_residual_bwd(const float* x0_pred, const float* bcs, const float* kloc, int kloc_stride,
                                      const int32_t* elem_dofs, const int32_t* dof_elems, int nel, const float* g_residual,
                                      const float* g_model_out, const float* g_comp_shift, float* g_x0_pred, int B,
                                      void* stream) {
  const int E = nel * nel, nn = nel + 1, ndof = 2 * nn * nn;
  if (mech_args(nel, E, ndof, kloc, elem_dofs, dof_elems)) return -1;
  MechMesh ms{elem_dofs, dof_elems, kloc, kloc_stride, E, ndof, nel, nn};
  const size_t lds = (size_t)(3 * ndof + E) * sizeof(float);
  if (lds > 150 * 1024) return fail("mech: mesh does not fit LDS");
  static bool attr = false;
  if (!attr) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&mech_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
    attr = true;
  }
  hipLaunchKernelGGL(HIP_KERNEL_NAME(mech_kernel<true>), dim3(B), dim3(256), lds, as_stream(stream), ms, x0_pred, bcs, nullptr, nullptr,
                     nullptr, nullptr, g_residual, g_model_out, g_comp_shift, g_x0_pred);
  PIDM_CHECK_LAUNCH("mech_kernel<bwd>");
  return 0;
}


extern "C" size_t pidm_mech_loss_ws(int B) { return (size_t)B * 8 * sizeof(double) + 256; }

extern "C" int pidm_mech_loss_fwd_bwd(const float* x0_pred, const float* target, const float* bcs, const float* vf, const float* p2w,
                                      const float* inv_var, const float* inv_var_sum, float c_data, float c_residual, float c_ineq,
                                      float lambda_opt, const float* kloc, int kloc_stride, const int32_t* elem_dofs,
                                      const int32_t* dof_elems, int nel, float* grad_x0_pred, float* out_scalars, void* workspace,
                                      int B, void* stream) {
  const int E = nel * nel, nn = nel + 1, ndof = 2 * nn * nn;
  if (mech_args(nel, E, ndof, kloc, elem_dofs, dof_elems)) return -1;
  if (!x0_pred || !target || !bcs || !vf || !p2w || !inv_var || !grad_x0_pred || !out_scalars || !workspace) return fail("mech_loss: null buffer");
  if (B <= 0) return fail("mech_loss: B must be positive");
  MechMesh ms{elem_dofs, dof_elems, kloc, kloc_stride, E, ndof, nel, nn};
  const size_t lds = (size_t)(3 * ndof + E) * sizeof(float);
  if (lds > 150 * 1024) return fail("mech: mesh does not fit LDS");
  static bool attr = false;
  if (!attr) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&mech_loss_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
    attr = true;
  }
  double* partial = reinterpret_cast<double*>((reinterpret_cast<size_t>(workspace) + 255) & ~(size_t)255);
  hipLaunchKernelGGL(mech_loss_kernel, dim3(B), dim3(256), lds, as_stream(stream), ms, x0_pred, target, bcs, vf, p2w, inv_var, inv_var_sum,
                     c_data, c_residual, c_ineq, lambda_opt, B, grad_x0_pred, partial);
  PIDM_CHECK_LAUNCH("mech_loss_kernel");
  hipLaunchKernelGGL(mech_loss_finalize, dim3(1), dim3(64), 0, as_stream(stream), partial, p2w, inv_var, inv_var_sum, c_data, c_residual,
                     c_ineq, lambda_opt, B, nn, ndof, out_scalars);
  PIDM_CHECK_LAUNCH("mech_loss_finalize");
  return 0;
}

extern "C" int pidm_mech_apply(const float* rho, const float* u_img, const float* bcs, const float* kloc, int kloc_stride,
                               const int32_t* elem_dofs, const int32_t* dof_elems, int nel, float* residual, float* comp_uf,
                               int B, void* stream) {
  const int E = nel * nel, nn = nel + 1, ndof = 2 * nn * nn;
  if (mech_args(nel, E, ndof, kloc, elem_dofs, dof_elems)) return -1;
  if (!rho || !u_img || !bcs || !residual || !comp_uf) return fail("mech_apply: null buffer");
  MechMesh ms{elem_dofs, dof_elems, kloc, kloc_stride, E, ndof, nel, nn};
  const size_t lds = (size_t)(ndof + E) * sizeof(float);
  if (lds > 150 * 1024) return fail("mech: mesh does not fit LDS");
  static bool attr = false;
  if (!attr) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&mech_apply_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
    attr = true;
  }
  hipLaunchKernelGGL(mech_apply_kernel, dim3(B), dim3(256), lds, as_stream(stream), ms, rho, u_img, bcs, residual, comp_uf);
  PIDM_CHECK_LAUNCH("mech_apply_kernel");
  return 0;
}

extern "C" size_t pidm_mech_solve_ws_bytes(int nel, int B) {
  return (size_t)B * 4 * (2 * (size_t)(nel + 1) * (nel + 1)) * sizeof(double) + 256;
}

extern "C" int pidm_mech_solve(const float* rho, const float* bcs, const float* kloc, int kloc_stride, const int32_t* elem_dofs,
                               const int32_t* dof_elems, int nel, float bin_threshold, float bin_hi, float bin_lo, int max_iter,
                               double rtol, float* u_dofs, float* compliance, float* rho_mean, int32_t* iters, float* relres,
                               void* workspace, int B, void* stream) {
  const int E = nel * nel, nn = nel + 1, ndof = 2 * nn * nn;
  if (mech_args(nel, E, ndof, kloc, elem_dofs, dof_elems)) return -1;
  if (!rho || !bcs || !compliance || !workspace) return fail("mech_solve: null buffer");
  if (max_iter < 1 || !(rtol > 0.0)) return fail("mech_solve: max_iter >= 1 and rtol > 0 required");
  MechMesh ms{elem_dofs, dof_elems, kloc, kloc_stride, E, ndof, nel, nn};
  const size_t lds = (size_t)ndof * sizeof(double) + (size_t)E * sizeof(float);
  if (lds > 150 * 1024) return fail("mech_solve: mesh does not fit LDS");
  static bool attr = false;
  if (!attr) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&mech_pcg_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
    attr = true;
  }
  double* ws = reinterpret_cast<double*>((reinterpret_cast<size_t>(workspace) + 255) & ~(size_t)255);
  hipLaunchKernelGGL(mech_pcg_kernel, dim3(B), dim3(256), lds, as_stream(stream), ms, rho, bcs, bin_threshold, bin_hi, bin_lo, max_iter,
                     rtol, ws, u_dofs, compliance, rho_mean, iters, relres);
  PIDM_CHECK_LAUNCH("mech_pcg_kernel");
  return 0;
}

extern "C" int pidm_floating_material(const float* rho, float threshold, int nel, int32_t* n_components, int B, void* stream) {
  if (!rho || !n_components) return fail("floating_material: null buffer");
  const size_t lds = (size_t)nel * nel * sizeof(int);
  if (nel < 1 || lds > 150 * 1024) return fail("floating_material: nel=%d out of range", nel);
  static bool attr = false;
  if (!attr) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&floating_material_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
    attr = true;
  }
  hipLaunchKernelGGL(floating_material_kernel, dim3(B), dim3(256), lds, as_stream(stream), rho, threshold, nel, n_components);
  PIDM_CHECK_LAUNCH("floating_material_kernel");
  return 0;
}
