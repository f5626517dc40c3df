/* pidm.h - C ABI of libpidm_hip.so, the MI355X (gfx950) engine behind the reference's Python API.
 *
 * The reference (jhbastek/PhysicsInformedDiffusionModels) is pure Python/PyTorch and has no FFI of its
 * own; the "binding a maintainer would add" is a ctypes stub.  Each entry point below
 * names the reference code it replaces (paths relative to the reference repo root).
 *
 * Conventions
 *   - extern "C", plain pointers and sizes, no torch types.  All pointers are DEVICE pointers unless
 *     the name ends in _host.  The library never allocates or frees caller memory, never synchronises
 *     the device, and enqueues every kernel on the caller-supplied stream (a hipStream_t passed as
 *     void*; NULL = the null stream).
 *   - return 0 on success, negative on error; message via pidm_last_error() (thread-local).
 *   - fp32 everywhere; image tensors are channels-last (NHWC, i.e. the reference's own [B, P*P, C]
 *     "b_xy_c" interchange layout) unless stated.
 */
#ifndef PIDM_H
#define PIDM_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PIDM_ABI_VERSION 1

int pidm_version(void);
const char* pidm_last_error(void);
/* "hip" for the product library, "hipemu" for the host-emulated test build (tests/hipemu). */
const char* pidm_backend(void);

/* bench-only: per-launch timing of the dominant kernels with HIP events recorded on the launch stream.
 * class 0 = conv forward/dgrad on the fp32 MFMA (work = algorithmic FLOPs), class 1 = conv wgrad on the fp32 MFMA,
 * class 2 / 3 = the same two in split form on the bf16 pipe (each launch is in exactly one class). */
int pidm_prof_enable(int on);
int pidm_prof_collect(double* ms4, long long* launches4, double* work4);
/* bench-only, per KERNEL: between _begin and _collect every launch of the library records one event on `stream` (graph replay and
 * the side-stream overlap are off meanwhile, so launches are back to back on that stream); a launch's time is the interval since
 * the previous launch's event.  _collect writes one line per kernel name, largest total first:
 *   name \t launches \t total_ms \t work \t class \n   (work: FLOPs the launcher declared, 0 = none; class as above, -1 = none)
 * and returns the bytes the table needs incl. the terminating 0 (the text is truncated to `cap`), or -1.
 * Single-threaded by contract: between _begin and _collect only ONE host thread may call into the library. */
int pidm_prof_kernels_begin(void* stream);
long long pidm_prof_kernels_collect(char* buf, size_t cap);

/* ---------------------------------------------------------------------------------------------
 * Darcy PDE residual                    replaces ResidualsDarcy.compute_residual's stencil part
 *   src/residuals_darcy.py:137-183 + StencilGradientComputation.forward src/grad_utils.py:64-146
 * x0      [B,2,P,P] NCHW (ch0 = pressure, ch1 = permeability) - the layout the reference UNet returns
 * residual[B,P*P,3]  (eq, bc0, bc1)
 * inv_h0 = 1/d0, inv_h1 = 1/d1 (d1 negative when reverse_d1).  f_s [P*P] source field.
 * ------------------------------------------------------------------------------------------- */
int pidm_darcy_residual_fwd(const float* x0, const float* f_s, float inv_h0, float inv_h1,
                            float* residual, int B, int P, void* stream);
/* adjoint: grad_x0[B,2,P,P] = (d residual / d x0)^T grad_res[B,P*P,3] */
int pidm_darcy_residual_bwd(const float* x0, const float* grad_res, float inv_h0, float inv_h1,
                            float* grad_x0, int B, int P, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Fused PIDM loss (Darcy, mean estimation)          replaces DenoisingDiffusion.model_estimation_loss
 *   src/denoising_utils.py:666-692  (data term with min-SNR weight + residual virtual likelihood)
 * x0, x0_pred [B,2,P,P] NCHW; p2w[B] = p2_loss_weight[t_b]; inv_var[B] = 1/posterior_variance_clipped[t_b]
 * out_scalars[4] = {loss, c_data*data_loss, mean|r|, 0};  grad_x0_pred [B,2,P,P] = d loss / d x0_pred.
 * residual [B,P*P,3] is written as a by-product (the API returns it).  workspace: >= pidm_darcy_loss_ws(B,P) bytes.
 * ------------------------------------------------------------------------------------------- */
size_t pidm_darcy_loss_ws(int B, int P);
int pidm_darcy_loss_fwd_bwd(const float* x0, const float* x0_pred, const float* f_s, const float* p2w,
                            const float* inv_var, float c_data, float c_residual, float inv_h0, float inv_h1,
                            float* residual, float* grad_x0_pred, float* out_scalars, void* workspace,
                            int B, int P, void* stream);
/* The same with the per-sample weights looked up inside the kernel: t int64 [B] (the step's time levels), p2w_table =
 * diff_dict['p2_loss_weight'], var_table = diff_dict['posterior_variance_clipped'] (its reciprocal is taken in the kernel) -
 * replaces the two extract() gathers + the reciprocal of src/denoising_utils.py:677,689-692 as well. */
int pidm_darcy_loss_fwd_bwd_t(const float* x0, const float* x0_pred, const float* f_s, const int64_t* t,
                              const float* p2w_table, const float* var_table, float c_data, float c_residual,
                              float inv_h0, float inv_h1, float* residual, float* grad_x0_pred, float* out_scalars,
                              void* workspace, int B, int P, void* stream);

/* CoCoGen residual correction: max over all entries of d residual / d p per sample
 *   replaces the dense vmap(jacfwd) Jacobian of src/residuals_darcy.py:217-231 (400 MB per 64x64 sample) by the
 *   analytic stencil rows.  x0: [B,2,P,P] (p, K); max_dr_dp: [B]. */
int pidm_darcy_jacobian_max(const float* x0, float inv_h0, float inv_h1, float* max_dr_dp, int B, int P, void* stream);

/* q-sample  x_t = a[t_b] x0 + am1[t_b] eps          replaces src/denoising_utils.py:633-638
 * writes x_t in channels-last [B,P*P,C] (what the UNet consumes) from NCHW x0/eps. */
int pidm_qsample_nhwc(const float* x0, const float* eps, const float* a_t, const float* am1_t, float* xt_nhwc,
                      int B, int C, int HW, void* stream);
/* the same with the extract() gathers folded in: t int64 [B], a_table = diff_dict['alphas_bar_sqrt'],
 * am1_table = diff_dict['one_minus_alphas_bar_sqrt']  (src/denoising_utils.py:302-306,633-638) */
int pidm_qsample_nhwc_t(const float* x0, const float* eps, const int64_t* t, const float* a_table, const float* am1_table,
                        float* xt_nhwc, int B, int C, int HW, void* stream);
/* ancestral update x_{t-1} = c1 x0_pred + c2 x_t + sigma z   replaces src/denoising_utils.py:441-455 */
int pidm_psample_update(const float* x0_pred, const float* x_t, const float* z, float c1, float c2, float sigma,
                        float* x_prev, size_t n, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Fused global-norm clip + Adam over flat fp32 buffers    replaces main.py:165-166
 *   torch.nn.utils.clip_grad_norm_(model.parameters(), 1.) ; optimizer.step()   (torch.optim.Adam, no weight decay/amsgrad)
 * param/grad/exp_avg/exp_avg_sq: n floats each, 16-byte aligned; `step` is the 1-based update count (bias correction);
 * max_norm <= 0 disables clipping; total_norm_out (device float, may be NULL) receives the pre-clip global L2 norm.
 * workspace: pidm_clip_adam_ws_bytes() bytes.  Deterministic (fixed-order sums). */
size_t pidm_clip_adam_ws_bytes(void);
int pidm_clip_adam_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, size_t n, double lr, double beta1,
                        double beta2, double eps, long long step, double max_norm, float* total_norm_out, void* workspace,
                        void* stream);

/* The same step with the parameter EMA of main.py:178-179 folded in (EMA.update, src/denoising_utils.py:174-177):
 *   shadow = (1 - mu) * p_new + mu * shadow   with the reference's fp32 roundings (two products, one sum; no fma),
 * so the shadow is bit-identical to the reference's per-tensor update applied to the same p_new.  ema_shadow: n floats.
 * pidm_ema_update is the stand-alone form (one read of p and shadow, one write of shadow) for steps taken by another
 * optimizer. */
int pidm_clip_adam_ema_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, float* ema_shadow, size_t n,
                            double lr, double beta1, double beta2, double eps, long long step, double max_norm, double ema_mu,
                            float* total_norm_out, void* workspace, void* stream);
int pidm_ema_update(float* ema_shadow, const float* param, size_t n, double ema_mu, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Mechanics residual r = K(rho) u - f, matrix-free      replaces ResidualsMechanics.compute_residual
 *   src/residuals_mechanics_K.py:198-274 (dense 8450x8450 index_put assembly + einsum) and resize_image :10-21
 * x0_pred [B,3,nel,nel] NCHW (u1,u2,rho); bcs [B,4,nel+1,nel+1] (bc_x, bc_y, load_x, load_y); vf [B]
 * mesh: kloc [E,8,8] (kloc_stride 64) or one 8x8 (stride 0); elem_dofs int32 [E,8]; dof_elems int32 [ndof,4,2]
 * outputs: residual [B,ndof]; model_out [B,3,nel+1,nel+1] (u bilinearly resized, rho zero padded);
 * comp_shift [B,2] = (compliance u^T K u, mean(rho) - vf).  The adjoint takes gradients of all four.
 * ------------------------------------------------------------------------------------------- */
int pidm_bilinear_resize(const float* in, float* out, int BC, int Hi, int Ho, void* stream);
int pidm_mech_residual_fwd(const float* x0_pred, const float* bcs, const float* vf, const float* kloc, int kloc_stride,
                           const int32_t* elem_dofs, const int32_t* dof_elems, int nel, float* residual, float* model_out,
                           float* comp_shift, int B, void* stream);
int pidm_mech_residual_bwd(const float* x0_pred, const float* bcs, const float* kloc, int kloc_stride,
                           const int32_t* elem_dofs, const int32_t* dof_elems, int nel, const float* g_residual,
                           const float* g_model_out, const float* g_comp_shift, float* g_x0_pred, int B, void* stream);

/* Fused mechanics training loss + its gradient wrt the network output     replaces src/denoising_utils.py:666-708 (loss
 * algebra on top of compute_residual) for gov_eqs == 'mechanics': data term on model_out = (u resized to nn x nn, rho zero padded)
 * against target x_0 [B,3,nn,nn], residual term, the [B,B]-broadcast inequality term (:697, only when c_ineq > 0) and the
 * compliance term, all with injected per-sample p2_loss_weight[t_b] and 1/posterior_variance_clipped[t_b].
 * inv_var_sum (device float, may be NULL): value to use for sum_i inv_var_i in the inequality term instead of the local sum -
 * under data parallelism the caller passes the all-rank sum divided by the number of ranks (it depends on t only), which makes
 * the rank-averaged loss and gradient equal to the single-process global-batch ones.
 * out_scalars[8]: loss, data loss, mean |r|, mean shift (0 unless c_ineq > 0), mean compliance, 0, 0, 0.
 * grad_x0_pred [B,3,nel,nel] = d loss / d x0_pred.  workspace: pidm_mech_loss_ws(B) bytes. */
size_t pidm_mech_loss_ws(int B);
int pidm_mech_loss_fwd_bwd(const float* x0_pred, const float* target, const float* bcs, const float* vf, const float* p2w,
                           const float* inv_var, const float* inv_var_sum, float c_data, float c_residual, float c_ineq,
                           float lambda_opt, const float* kloc, int kloc_stride, const int32_t* elem_dofs, const int32_t* dof_elems,
                           int nel, float* grad_x0_pred, float* out_scalars, void* workspace, int B, void* stream);

/* Topology-optimisation evaluation block      replaces src/residuals_mechanics_K.py:276-347,369-380
 *   pidm_mech_apply:  residual = K_closed(rho) u - f and comp_uf = u.f for nodal displacement images u [B,2,nn,nn]
 *                     (the "residual of opt_disp should be zero" check and compliance_data, :293-296)
 *   pidm_mech_solve:  u = K_closed(rho')^-1 f per sample (reference: torch.linalg.solve on the dense 8450^2 matrix, :321-323)
 *                     by Jacobi-preconditioned CG on the matrix-free operator in fp64; rho' = rho if bin_threshold < 0,
 *                     else (rho > bin_threshold ? bin_hi : bin_lo)  (:299-301).  compliance[b] = f.u, rho_mean[b] = mean(rho').
 *                     Stops at ||r|| <= rtol ||f|| or max_iter; iters / relres (may be NULL) report what happened.
 *   pidm_floating_material: number of 8-connected components of {rho > threshold} per sample (cv2.connectedComponents, :369-380)
 * rho: [B, nel*nel]; bcs: [B,4,nn,nn]; workspace: pidm_mech_solve_ws_bytes(nel, B). */
int pidm_mech_apply(const float* rho, const float* u_img, const float* bcs, const float* kloc, int kloc_stride,
                    const int32_t* elem_dofs, const int32_t* dof_elems, int nel, float* residual, float* comp_uf, int B,
                    void* stream);
size_t pidm_mech_solve_ws_bytes(int nel, int B);
int pidm_mech_solve(const float* rho, const float* bcs, const float* kloc, int kloc_stride, const int32_t* elem_dofs,
                    const int32_t* dof_elems, int nel, float bin_threshold, float bin_hi, float bin_lo, int max_iter,
                    double rtol, float* u_dofs, float* compliance, float* rho_mean, int32_t* iters, float* relres,
                    void* workspace, int B, void* stream);
int pidm_floating_material(const float* rho, float threshold, int nel, int32_t* n_components, int B, void* stream);

/* ---------------------------------------------------------------------------------------------
 * UNet engine                           replaces Unet3D.forward src/unet_model.py:542-623 + autograd
 * ------------------------------------------------------------------------------------------- */
typedef struct pidm_unet pidm_unet;
typedef struct pidm_unet_cfg {
  int dim;            /* base width (32 Darcy, 128 mechanics) */
  int channels;       /* input channels */
  int out_dim;        /* output channels */
  int n_levels;       /* len(dim_mults) */
  int dim_mults[8];
  int heads;          /* 8 */
  int dim_head;       /* 32 */
  int groups;         /* GroupNorm groups, 8 */
  int init_kernel;    /* 7 */
  int image_size;     /* P (square images) */
  int sigmoid_last_channel;
  int self_condition; /* 1: init_conv reads 2*channels inputs, cat(x_self_cond, x) (src/unet_model.py:428,564-566) */
} pidm_unet_cfg;

int pidm_unet_create(const pidm_unet_cfg* cfg, pidm_unet** out);
void pidm_unet_destroy(pidm_unet* h);
/* canonical list of the parameter tensors forward() reads (state_dict names, 259 for the Darcy model) */
int pidm_unet_num_params(const pidm_unet* h);
const char* pidm_unet_param_name(const pidm_unet* h, int i);
size_t pidm_unet_param_numel(const pidm_unet* h, int i);
/* bytes of caller-provided device scratch needed for batch B (persistent region + per-call arena) */
size_t pidm_unet_workspace_bytes(const pidm_unet* h, int B, int training);
/* param_ptrs_host[i]: device pointer of parameter i in the reference's own layout
 * (Conv3d [Cout,Cin,1,k,k], ConvTranspose3d [Cin,Cout,1,k,k], Linear [out,in], ...).  grad_ptrs_host[i]:
 * where backward WRITES (not accumulates) d loss / d param i, same layout; may be NULL for inference. */
int pidm_unet_bind(pidm_unet* h, const void* const* param_ptrs_host, void* const* grad_ptrs_host);
/* x: [B,P*P,C] channels-last; t: int64 [B]; out: [B,out_dim,P,P] NCHW (reference output layout).
 * save_for_backward != 0 keeps activations in the workspace until pidm_unet_backward. */
int pidm_unet_forward(pidm_unet* h, const float* x_nhwc, const int64_t* t, float* out_nchw, int B,
                      int save_for_backward, int repack_weights, void* workspace, size_t workspace_bytes, void* stream);
/* Gradient-guidance conditioning branch (src/unet_model.py:521-528,571-587: x = combine_conv(cat(init_conv(x), emb_conv(cond)))).
 * Its 6 parameters (emb_conv.0/2, combine_conv; weight+bias) are the LAST pidm_unet_num_cond_params() entries of the
 * canonical list.  pidm_unet_enable_cond sizes the workspace for the branch (call it before pidm_unet_workspace_bytes);
 * pidm_unet_set_condition hands the (already classifier-free-masked) field [B,P*P,C] to the NEXT pidm_unet_forward only.
 * A backward whose forward had no conditioning input zero-fills the gradients of those 6 parameters. */
int pidm_unet_num_cond_params(const pidm_unet* h);
int pidm_unet_enable_cond(pidm_unet* h, int on);
int pidm_unet_set_condition(pidm_unet* h, const float* cond_nhwc);
/* grad_out: [B,out_dim,P,P] NCHW.  grad_x (may be NULL): [B,P*P,C].  Writes all bound grads. */
int pidm_unet_backward(pidm_unet* h, const float* grad_out_nchw, float* grad_x_nhwc, int B, void* workspace,
                       size_t workspace_bytes, void* stream);

/* Data-parallel overlap (no reference counterpart: the reference is single-process).  The deferred gradient
 * reduction of pidm_unet_backward runs in n_phases (1..3) launches - after the decoder half (ups.*, final_conv.*), after the
 * encoder half (downs.*, mid_*), at the end (everything else) - and records events[k] (hipEvent_t, may be NULL) on the
 * backward's stream after phase k, so the caller can start the collective over that phase's gradients while the rest of
 * backward is still running.  pidm_unet_grad_phase_range: canonical parameter index range [first, end) finalised by a phase;
 * the LAST phase additionally finalises every parameter outside the earlier phases' ranges. */
int pidm_unet_set_grad_events(pidm_unet* h, int n_phases, void* const* events);
int pidm_unet_grad_phase_range(const pidm_unet* h, int n_phases, int phase, int* first_param, int* end_param);

/* ---------------------------------------------------------------------------------------------
 * Unit-level kernel entry points (used by the parity tests; the engine calls the same launchers)
 * ------------------------------------------------------------------------------------------- */
typedef struct pidm_conv_desc {
  int B, Hi, Wi;        /* input  [B,Hi,Wi,C0(+C1)] channels-last */
  int C0, C1;           /* channels taken from src0 / src1 (concat elimination), C1 may be 0 */
  int ld0, ld1;         /* channel strides (floats per pixel) of src0 / src1 */
  int Cout;
  int KH, KW, stride, pad;
  int transposed;       /* 1: ConvTranspose 4x4 s2 p1 executed as 4 output-parity 2x2 convolutions */
  int out_nchw;         /* 1: write [B,Cout,Ho,Wo] instead of channels-last */
  int ldo;              /* channel stride of the output / residual (channels-last) */
} pidm_conv_desc;
size_t pidm_conv_packed_weight_floats(const pidm_conv_desc* d);
/* mode 0: forward pack from [Cout,Cin,KH,KW]; 1: dgrad pack (flipped+transposed) from the same tensor;
 * for transposed convs the source is [Cin,Cout,KH,KW]. */
int pidm_conv_pack_weights(const pidm_conv_desc* d, const float* w_ref, float* w_packed, int mode, void* stream);
int pidm_conv_forward(const pidm_conv_desc* d, const float* src0, const float* src1, const float* w_packed,
                      const float* bias, const float* residual, float* out, void* stream);
/* forward convolution feeding a GroupNorm (Block: proj -> norm, src/unet_model.py:227-241): besides `out`, the epilogue leaves
 * per-(image, chunk, group) sums and sums of squares of the output in partial[B][chunks][groups][2] (doubles; size it for H*W/32
 * chunks).  A chunk is a set of pixels of one image, the chunks of an image cover it once: 32 consecutive pixels behind the tile
 * kernels (H*W/32 chunks), a strip of 32 pixels x R rows behind the row-streaming kernel.  Returns the chunks per image
 * written, 0 when the shape has no statistics epilogue (then only `out` is written), < 0 on error. */
int pidm_conv_forward_gn_partials(const pidm_conv_desc* d, const float* src0, const float* src1, const float* w_packed,
                                  const float* bias, float* out, int groups, double* partial, void* stream);
/* adjoint wrt the input: dx[B,Hi,Wi,Cin] (+ residual) from dy[B,Ho,Wo,Cout]; weights packed with mode 1 */
size_t pidm_conv_dgrad_packed_weight_floats(const pidm_conv_desc* d);
int pidm_conv_dgrad(const pidm_conv_desc* d, const float* dy, int ld_dy, const float* w_packed_dgrad,
                    const float* residual, float* dx, int ld_dx, void* stream);
size_t pidm_conv_wgrad_ws(const pidm_conv_desc* d);
/* dW (reference layout) and dbias (may be NULL) from input x (src0/src1) and output-gradient dy */
int pidm_conv_wgrad(const pidm_conv_desc* d, const float* src0, const float* src1, const float* dy, int ld_dy,
                    float* dw_ref, float* dbias, void* workspace, void* stream);

/* Linear attention core of SpatialLinearAttention.forward (src/unet_model.py:281-299) between the to_qkv and to_out 1x1
 * convolutions: q.softmax(dim=-2) * scale, k.softmax(dim=-1), context = k v^T / N, out = context^T q.
 *   qkv  [B][N][3*heads*32] channels-last (q | k | v, head-major), N = h*w pixels;  out [B][N][heads*32]
 *   saved for the backward: kstat [B][heads*32][2], ctx [B][heads][32][32], qstat [B][N][heads][2]
 *   backward: d_out [B][N][heads*32] -> dqkv [B][N][3*heads*32]                                                     */
size_t pidm_linear_attention_ws(int B, int N, int heads);
int pidm_linear_attention_forward(const float* qkv, float* out, float* kstat, float* ctx, float* qstat, int B, int N,
                                  int heads, void* workspace, void* stream);
int pidm_linear_attention_backward(const float* qkv, const float* kstat, const float* qstat, const float* ctx,
                                   const float* d_out, float* dqkv, int B, int N, int heads, void* workspace, void* stream);
/* The same backward fused with the to_out 1x1 projection (src/unet_model.py:298): takes the gradient d_y [B][N][Cout] of the
 * projection's output and its weights w_out [Cout][heads*32] (reference layout), returns dqkv and the projection's weight
 * gradient dw_out [Cout][heads*32] without materialising d_out.  N % 128 == 0, Cout in {32, 64, 128}. */
/* forward counterpart: y [B][N][Cout] = to_out(attention(qkv)) + bias + residual (either may be NULL) without storing the
 * heads*32-channel attention output; workspace as pidm_linear_attention_ws. */
int pidm_linear_attention_out_forward(const float* qkv, const float* w_out, const float* bias, const float* residual, float* y,
                                      int Cout, float* kstat, float* ctx, float* qstat, int B, int N, int heads, void* workspace,
                                      void* stream);
size_t pidm_linear_attention_out_backward_ws(int B, int N, int heads, int Cout);
int pidm_linear_attention_out_backward(const float* qkv, const float* kstat, const float* qstat, const float* ctx,
                                       const float* d_y, int ld_dy, const float* w_out, int Cout, float* dqkv, float* dw_out,
                                       int B, int N, int heads, void* workspace, void* stream);

/* Linear attention WITHOUT a qkv tensor (csrc/k_attn_proj.hip)      replaces Residual(PreNorm(SpatialLinearAttention)) between the
 * LayerNorm and the residual add, src/unet_model.py:281-299 + :139-145, and its autograd.
 * xn [B,N,C] = LayerNorm output (channels-last); w_qkv [3*heads*32, C] and w_out [C, heads*32] in the reference layouts
 * (to_qkv.weight, to_out.weight); bias [C]; resid [B,N,C] = the block input x; y [B,N,C] = to_out(attention) + bias + x.
 * The forward keeps `saved` (pidm_lap_saved_floats floats: k-softmax statistics, M, ctx, P per image and head) and
 * qstat [B,N,heads,2] (q-softmax max, 1/sum) for the backward.  The backward takes dY = d loss / d y and WRITES d_xn (without
 * the residual's own dY -> dx share), d_w_qkv, d_w_out.  C in {32, 64}, N % 32 == 0, heads <= 8.  workspace: pidm_lap_ws bytes. */
size_t pidm_lap_ws(int B, int N, int heads, int C);
size_t pidm_lap_saved_floats(int B, int heads, int C);
int pidm_lap_forward(const float* xn, const float* w_qkv, const float* w_out, const float* bias, const float* resid, float* y,
                     float* saved, float* qstat, int C, int B, int N, int heads, void* workspace, void* stream);
int pidm_lap_backward(const float* xn, const float* dy, const float* w_qkv, const float* w_out, const float* saved,
                      const float* qstat, float* d_xn, float* d_w_qkv, float* d_w_out, int C, int B, int N, int heads,
                      void* workspace, void* stream);

/* measurement aid (tools/conv_trace.py): cycle stamps of the streaming 3x3 convolution kernel, written when PIDM_STREAM_TRACE is set */
int pidm_debug_stream_trace(unsigned long long* out256);
/* measurement aid (bench.py roofline.clock_probe): stamps of the last conv3x3_rs_kernel launch made with PIDM_RS_TRACE set -
 * out4 = {shader-clock counter, 100 MHz real-time counter} before and {.., ..} after the row loop of workgroup 0 / wave 0 */
int pidm_debug_conv_rs_trace(unsigned long long* out4);
/* cycle stamps of lap_bwd_kernel (PIDM_LAP_TRACE=1; tools/lap_trace.py): [wave slot < 2][tile round < 8][16] */
int pidm_debug_lap_trace(unsigned long long* out256);
/* test aid: host-to-device uploads of the deferred-reduction descriptor table since the library was loaded (a second identical
 * backward pass must not upload anything: the table is compared with what the device already holds) */
long long pidm_debug_reduce_table_uploads(void);
/* launch accounting since the library was loaded: out4 = {kernels enqueued launch by launch, hipGraphLaunch calls, kernels inside
 * the launched graphs, stream captures}.  pidm_unet_forward / pidm_unet_backward replay a captured hipGraph from the third call
 * with the same arguments on (PIDM_GRAPH=0: always launch by launch). */
int pidm_debug_launch_counts(long long* out4);
/* ---- data-parallel gradient exchange ----------------------------------------------------------------------
 * No reference counterpart: /root/reference/main.py:157-166 is single-device; north_star shards the batch over the GPUs of a node
 * and averages the gradients with an RCCL all-reduce over xGMI.  One communicator per process (= per GPU).  RCCL is bound with
 * dlopen at first use.  The reference-side binding: parallel.py (`PIDM_DP_NATIVE=1`) or any host with a way to hand rank 0's 128-byte
 * id to the other ranks.
 *   pidm_comm_available  every rank: 0 when RCCL could be bound, -1 + pidm_last_error() otherwise; starts no bootstrap thread
 *   pidm_comm_unique_id  rank 0: 128 opaque bytes to be sent to every rank (ncclGetUniqueId)
 *   pidm_comm_init       every rank, with the same id; uses the calling thread's current HIP device (ncclCommInitRank)
 *   pidm_allreduce_f32   in place on `buf[0..count)`, enqueued on `stream`; average != 0: mean over ranks, else sum
 *   pidm_comm_destroy    */
int pidm_comm_available(void);
int pidm_comm_unique_id(void* out128);
int pidm_comm_init(int rank, int world, const void* unique_id128, void** comm);
int pidm_allreduce_f32(void* comm, float* buf, size_t count, int average, void* stream);
int pidm_comm_destroy(void* comm);

/* The library reads each PIDM_* tuning variable from the environment once per process (the first time a launcher asks for it).
 * A process that changes one afterwards - the unit tests, A/B legs of bench.py - calls this to drop the snapshot; the next launch
 * re-reads.  No reference counterpart (the knobs select between kernels that compute the same result). */
int pidm_reload_knobs(void);

#ifdef __cplusplus
}
#endif
#endif /* PIDM_H */
