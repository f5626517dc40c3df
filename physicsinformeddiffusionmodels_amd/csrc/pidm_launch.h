// Host-side launcher declarations shared between the kernel translation units and the UNet engine.
#pragma once
#include <vector>

#include "pidm_common.h"

namespace pidm {
// queue of deferred reductions (split-K weight gradients, per-sample norm-parameter gradients): the producers leave their
// partial buffers alive and the engine runs ONE reduce_multi launch at the end of backward instead of ~120 tiny launches
struct ReduceQueue {
  std::vector<ReduceDesc> v;
  unsigned nblocks = 0;
  void push(const float* src, float* dst, const float* bsrc, float* bdst, size_t sstride, int nsplit, int M, int N, int T,
            int MP, int NP) {
    ReduceDesc d{};   // value-initialised: flush_reductions memcmp()s whole structs, tail padding included
    d.src = src; d.dst = dst; d.bsrc = bsrc; d.bdst = bdst; d.sstride = sstride;
    d.nsplit = nsplit; d.M = M; d.N = N; d.T = T; d.MP = MP; d.NP = NP;
    const size_t total = (size_t)M * N * T + (bdst ? M : 0);
    const size_t flat = (size_t)MP * T * NP;
    d.blk0 = nblocks;
    d.mode = (nsplit <= 8 && T <= 16 && (size_t)M * N * T >= 32768) ? 1 : 0;
    // mode 2 (round 4): many splits of a large contiguous slab (the 3x3 weight gradients: 64-256 slabs of 36 KB - 2.4 MB): the slab is
    // walked as a flat array in 16-byte pieces (512 contiguous bytes per split and quarter wave instead of 128), the [m][t][n] ->
    // [m][n][t] turn happens on the way out; the bias column sums become a small descriptor of their own
    if (d.mode == 0 && nsplit > 8 && flat >= 4096 && (flat & 3) == 0 && (sstride & 3) == 0 && (reinterpret_cast<size_t>(src) & 15) == 0) {
      d.mode = 2;
      d.bsrc = nullptr;
      d.bdst = nullptr;
      d.nblk = (unsigned)((flat / 4 + 31) / 32);
      nblocks += d.nblk;
      v.push_back(d);
      if (bdst) push(bsrc, bdst, nullptr, nullptr, (size_t)MP, nsplit, 1, M, 1, 1, MP);
      return;
    }
    d.nblk = d.mode ? (unsigned)M * (unsigned)((N + 63) / 64) : (unsigned)((total + 31) / 32);
    nblocks += d.nblk;
    v.push_back(d);
  }
};
// Weight-gradient problems whose launch is DEFERRED and grouped (round 5): nothing inside a backward pass reads a weight gradient
// before the deferred reduction, so the problems of a pass that one of three streaming kernel families takes (kWgFam*: 3x3 /
// stride-1 row-streaming, 4x4 / stride-2 row-streaming, 1x1 pixel streams) are collected here and run by ONE launch per family
// and flush (conv_wgrad_rs_multi_kernel, _rs4_multi_, _1x1_multi_).  Together they fill the chip, so each problem is planned with
// a quarter of the splits it would need alone (wgrad_group_splitdiv): fewer, longer work items and a quarter of the partial slabs
// - that, not the saved launch boundaries (~1 us each inside a replayed graph), is what pays.  The operands
// (taped activations, gradient buffers) and the partial slabs must stay alive until the flush: the engine keeps the backward
// arena's frames while grouping is on.
struct WgradQueue {
  struct Fam {
    std::vector<WgradItem> v;
    std::vector<double> fl;  // algorithmic FLOPs per item (profiling hooks)
    unsigned nblocks = 0;
    size_t done = 0;         // rows already launched
  } f[kWgFams];
  void push(int fam, WgradItem it, double flops) {  // `it` value-initialised and filled field by field by the launcher (tables are memcmp'd)
    Fam& q = f[fam];
    it.blk0 = q.nblocks;
    q.nblocks += it.gx * it.gy;
    q.v.push_back(it);
    q.fl.push_back(flops);
  }
  size_t size() const {
    size_t n = 0;
    for (const auto& q : f) n += q.v.size();
    return n;
  }
  size_t pending() const {
    size_t n = size();
    for (const auto& q : f) n -= q.done;
    return n;
  }
  void clear() {
    for (auto& q : f) { q.v.clear(); q.fl.clear(); q.nblocks = 0; q.done = 0; }
  }
};
// rows [first, first + n) of the family's device table, whose workgroups are [blk_base, blk_base + nblocks) of its flat numbering
int launch_wgrad_multi(int fam, const WgradItem* table_dev, int first, int n, unsigned blk_base, unsigned nblocks, hipStream_t st);
int launch_wgrad_rs_multi(const WgradItem* table_dev, int first, int n, unsigned blk_base, unsigned nblocks, hipStream_t st);
int launch_wgrad_rs4_multi(const WgradItem* table_dev, int first, int n, unsigned blk_base, unsigned nblocks, hipStream_t st);
int launch_wgrad_1x1_multi(const WgradItem* table_dev, int first, int n, unsigned blk_base, unsigned nblocks, hipStream_t st);
int launch_wgrad_1x1_split_multi(const WgradItem* table_dev, int first, int n, unsigned blk_base, unsigned nblocks, hipStream_t st);
int wgrad_group_splitdiv();     // PIDM_WGRAD_GROUP_SPLITDIV (default 4)
// the device-side description of one problem: the geometry fields the kernels read + operands + its own grid (k_wgrad_rs.hip)
WgradItem wgrad_item(const WgradGeom& wg, const float* src0, const float* src1, const float* dy, float* partial, float* bias_partial,
                     unsigned gx, unsigned gy, int kind);
int launch_reduce_multi(const ReduceDesc* table_dev, int ndesc, unsigned nblocks, hipStream_t st, unsigned blk_base = 0);
int launch_split_reduce(const float* partial, float* dst, const float* bias_partial, float* dbias, int nsplit, int M, int N, int T,
                        int MP, int NP, hipStream_t st);
// k_conv.hip
int make_geom(ConvGeom* g, int kind, int B, int Hi, int Wi, int C0, int C1, int ld0, int ld1, int Cout, int KH, int KW,
              int stride, int pad, int out_nchw, int ldo, int ldr);
int geom_dgrad(const pidm_conv_desc* d, int ld_dy, int ld_dx, ConvGeom* g, int* pack_kind);
int geom_fwd(const pidm_conv_desc* d, ConvGeom* g);   // geometry of the forward op described by a public pidm_conv_desc
// re-tile a stride-1 geometry for a bm-pixel workgroup tile (bm = 256: two m-tiles per wave)
inline bool retile_bm(ConvGeom* g, int bm) {
  if (g->Wv > bm || g->nz != 1 || g->nph != 1) return false;
  const int TH = bm / g->Wv < g->Hv ? bm / g->Wv : g->Hv;
  if (g->Hv % TH) return false;
  int tsh = 0;
  while ((1 << tsh) < TH) ++tsh;
  if ((1 << tsh) != TH) return false;
  const int NI = bm / (g->Wv * TH);
  if (NI * g->Wv * TH != bm) return false;
  g->TH = TH; g->tsh = tsh; g->NI = NI;
  g->IHt = (TH - 1) * g->stride + g->KH;
  g->mIHt = g->IHt > 1 ? (unsigned)((0x100000000ULL + g->IHt - 1) / g->IHt) : 0;
  g->tiles_m = (NI > 1) ? cdiv(g->B, NI) : g->B * (g->Hv / TH);
  return true;
}
int pick_kc(int Cin);                         // K-chunk of the fp32 packing (k_conv.hip)
int pick_nt(int Cout, int tiles_m);
int packed_np(int Cout);
bool conv_nt4_ok(const ConvGeom& g, int KC);   // k_conv_fp32.hip: the permuted 128-channel tile of 1x1 convolutions applies
// k_conv_fp32.hip: the fp32-MFMA kernels behind launch_conv
// k_conv_rs.hip: the row-streaming 3x3 kernel of the wide levels (0: launched, 1: not its shape, < 0: error)
int launch_conv_rs(const ConvGeom& g, const float* src0, const float* src1, const unsigned short* wsplit, const float* bias,
                   const float* residual, float* out, hipStream_t st);
int launch_conv_fp32(const ConvGeom& g, const float* src0, const float* src1, const float* wp, const float* bias, const float* residual,
                     float* out, int sigmoid_last, hipStream_t st, int KC, int NT, bool nt4);
size_t packed_floats(const ConvGeom& g);
int packed_kp(const ConvGeom& g);
int launch_pack(const ConvGeom& g, int kind, const float* w_ref, float* w_packed, int srcKH, int srcKW, int n_off, int k_off,
                int n_src, int k_src, hipStream_t st);
unsigned make_pack_desc(const ConvGeom& g, int kind, const float* w_ref, float* w_packed, int srcKH, int srcKW, int n_off,
                        int k_off, int n_src, int k_src, PackDesc* d);
int launch_pack_multi(const PackDesc* table_dev, int ndesc, unsigned nblocks, hipStream_t st);
int launch_conv(const ConvGeom& g, const float* src0, const float* src1, const float* wp, const float* bias,
                const float* residual, float* out, int sigmoid_last, hipStream_t st);
// out[m][n] = sum_k A[m][k] W[n][k], few rows, very long k: split-K over workgroups + fixed-order reduction (k_conv.hip)
size_t smallm_splitk_ws_floats(int M, int N, int K);
bool smallm_splitk_ok(int M, int N, int K, int lda, int ldw);
int launch_smallm_splitk(const float* A, int lda, const float* W, int ldw, float* out, int M, int N, int K, float* scratch, hipStream_t st);
size_t wgrad_ws_bytes(const ConvGeom& g);
// wq (only together with defer): problems the row-streaming 3x3 kernel takes are queued for a grouped launch instead of launched
int launch_wgrad(const ConvGeom& g, const float* src0, const float* src1, const float* dy, int ld_dy, float* dw_ref,
                 float* dbias, void* workspace, hipStream_t st, ReduceQueue* defer = nullptr, WgradQueue* wq = nullptr);
// k_wgrad_rs.hip: 3x3 / stride-1 weight gradient, row-streaming split form (false: geometry not eligible)
bool launch_wgrad_rs(const WgradGeom& plan, const float* src0, const float* src1, const float* dy, int ld_dy, float* partial,
                     float* bias_partial, hipStream_t st, WgradGeom* used, WgradQueue* wq = nullptr);
bool wgrad_rs_queueable(const ConvGeom& g, const float* src0, const float* dy, int ld_dy);   // launch_wgrad_rs would take (queue) it
bool wgrad_rs4_eligible(const ConvGeom& g, int ld_dy);
bool launch_wgrad_rs4(const WgradGeom& plan, const float* src0, const float* src1, const float* dy, int ld_dy, float* partial,
                      float* bias_partial, hipStream_t st, WgradGeom* used, WgradQueue* wq = nullptr);
bool launch_wgrad_rs7(const WgradGeom& plan, const float* src0, const float* dy, int ld_dy, float* partial, float* bias_partial,
                      hipStream_t st, WgradGeom* used);
size_t colsum_ws_bytes(size_t rows, int C);
int launch_colsum(const float* x, size_t rows, int C, int ld, float* out, void* workspace, hipStream_t st,
                  ReduceQueue* defer = nullptr);
// k_norm.hip
size_t gn_ws_bytes(int B, int HW, int C, int G);
int launch_gn_stats(const float* x, int B, int HW, int C, int G, float* stats, void* ws, hipStream_t st);
bool gn_apply_ln_ok(int C);
int launch_gn_apply(const float* x, float* stats, const float* gamma, const float* beta, const float* ss, const float* ssb,
                    int ldss, const float* res, float* y, int B, int HW, int C, int G, void* ws, hipStream_t st, int part_chunks = 0,
                    const float* ln_gamma = nullptr, float* ln_out = nullptr);
int launch_gn_bwd(const float* x, const float* dy, const float* stats, const float* gamma, const float* beta, const float* ss,
                  const float* ssb, int ldss, float* dss, float* dx, float* dgamma, float* dbeta, int B, int HW, int C, int G,
                  void* ws, hipStream_t st, float* dgb_persist = nullptr, ReduceQueue* defer = nullptr, int part_chunks = 0);
int launch_layernorm_fwd(const float* x, const float* gamma, float* y, size_t npix, int C, hipStream_t st);
size_t layernorm_bwd_ws_bytes(int C);
int layernorm_bwd_gn_chunks(int B, int HW);
int launch_layernorm_bwd(const float* x, const float* gamma, const float* dy, const float* res, float* dx, float* dgamma,
                         size_t npix, int C, void* ws, hipStream_t st, ReduceQueue* defer = nullptr, float* res_colsum = nullptr,
                         const float* gn_x = nullptr, const float* gn_stats = nullptr, const float* gn_gamma = nullptr,
                         const float* gn_beta = nullptr, int gn_G = 0, int gn_B = 0, int gn_HW = 0, void* gn_part = nullptr);
int launch_act_fwd(const float* x, float* y, size_t n, int act, hipStream_t st);
int launch_act_bwd(const float* x, const float* dy, float* dx, size_t n, int act, hipStream_t st);
int launch_sinusoid(const int64_t* t, float* emb, int B, int dim, hipStream_t st);
int launch_copy_add(float* dst, int ldd, const float* a, int lda, const float* b, int ldb, size_t rows, int cols, hipStream_t st);
int launch_nchw_to_nhwc(const float* src, float* dst, int B, int C, int HW, const float* y_sig, hipStream_t st);
// k_attn.hip
size_t la_scratch_floats(int B, int N, int heads);
int launch_la_forward(const float* qkv, float* kstat, float* ctx, float* attn, float* qstat, int B, int N, int heads,
                      float* scratch, hipStream_t st);
bool la_fused_ok(int N, int heads, int Cout, int ld_dy);
bool la_fused_pays(int B, int N);
size_t la_fused_scratch_floats(int B, int N, int heads, int Cout);
int launch_la_forward_fused(const float* qkv, float* kstat, float* ctx, float* qstat, const float* w_out, const float* bias,
                            const float* resid, float* y, int Cout, int B, int N, int heads, float* scratch, hipStream_t st);
int launch_la_backward_fused(const float* qkv, const float* kstat, const float* qstat, const float* ctx, const float* dy, int ld_dy,
                             const float* w_out, int Cout, float* dctx, float* rowdot, float* dqkv, float* dwpart, int B, int N,
                             int heads, float* scratch, hipStream_t st);
int launch_la_backward(const float* qkv, const float* kstat, const float* qstat, const float* ctx, const float* dA, float* dctx,
                       float* rowdot, float* dqkv, int B, int N, int heads, float* scratch, hipStream_t st);
// k_attn_proj.hip: linear attention without a qkv tensor (q / k tiles rebuilt from xn on the matrix cores, v never formed)
bool lap_ok(int N, int heads, int C, int Cout);
size_t lap_scratch_floats(int B, int N, int heads, int C);
size_t lap_saved_floats(int B, int heads, int C);
size_t lap_dw_ranges(int N, int C);
size_t lap_bwd_tmp_floats(int B, int heads, int C);
int launch_lap_forward(const float* xn, const float* wqkv, const float* wout, const float* bias, const float* resid, float* y,
                       float* saved, float* qstat, int C, int B, int N, int heads, float* scratch, hipStream_t st);
int launch_lap_backward(const float* xn, const float* dy, const float* wqkv, const float* wout, const float* saved, const float* qstat,
                        float* dxn, float* dwqk_part, float* dwv_part, float* dwout_part, float* tmp, int C, int B, int N, int heads,
                        float* scratch, hipStream_t st);
int launch_mid_attn(const float* qkv, const float* dO, float* out, int B, int N, int heads, bool bwd, hipStream_t st);
}  // namespace pidm
