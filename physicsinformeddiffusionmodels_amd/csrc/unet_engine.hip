// Host orchestration of the denoiser UNet forward / backward on gfx950.
//
// Replaces Unet3D.forward (reference src/unet_model.py:542-623) and everything `loss.backward()` replays
// through it (main.py:164).  The layer graph is SURVEY Appendix A; every op is one of the hand-written
// kernels in k_conv / k_norm / k_attn.  No autograd, no tracing compiler: forward records raw pointers of the
// tensors the backward needs ("tape" region of the caller-provided workspace), backward walks the graph in
// reverse.  All launches go to the caller's stream; no allocation, no synchronisation.
//
// Workspace layout:  [ packed weights (fwd + dgrad packings, persistent across the step) | tape | temporaries ]
#include <map>
#include <string>
#include <vector>

#include <stdlib.h>

#include "pidm_launch.h"

extern char** environ;

namespace pidm {

struct Arena {
  char* base = nullptr;
  size_t off = 0, hwm = 0, cap = 0;
  bool dry = false;
  float* alloc(size_t nfloats) {
    off = align_up(off, 256);
    char* p = base + off;
    off += nfloats * sizeof(float);
    if (off > hwm) hwm = off;
    return reinterpret_cast<float*>(p);
  }
  size_t mark() const { return off; }
  void release(size_t m) { off = m; }
  bool overflow() const { return !dry && hwm > cap; }
};

struct ConvLayer {
  int w = -1, b = -1;  // parameter indices (b = -1: no bias)
  int C0 = 0, C1 = 0, Cout = 0, K = 1, stride = 1, pad = 0, transposed = 0, H = 1;
  size_t off_f = 0, off_d = 0;  // float offsets into the packed-weight region
  bool dgrad = true;
};

struct ResBlock {
  ConvLayer c1, c2, cr;
  bool has_mlp = false, has_res = false;
  int C0 = 0, C1 = 0, Co = 0, H = 0;
  int gn1w = -1, gn1b = -1, gn2w = -1, gn2b = -1, mlpw = -1, mlpb = -1;
  int ss_off = 0;
  // tape
  const float *x0 = nullptr, *x1 = nullptr;
  float *a = nullptr, *st1 = nullptr, *bact = nullptr, *c = nullptr, *st2 = nullptr;
};

struct AttnBlock {
  bool mid = false;
  int C = 0, H = 0, gamma = -1;
  ConvLayer qkv, out;
  const float* x = nullptr;
  float *xn = nullptr, *qkvb = nullptr, *kstat = nullptr, *qstat = nullptr, *ctx = nullptr, *attn = nullptr;
  bool xn_ready = false;     // forward: xn was written by the preceding ResnetBlock's last gn_apply (LayerNorm fused, round 6);
  float* out_pre = nullptr;  // ... which then also allocated this block's output IN FRONT of xn (and took the arena mark between
  size_t mk_pre = 0;         // them), so that an inference pass can still release xn when the block is done
  float* lsaved = nullptr;   // projected form (k_attn_proj.hip): k statistics | M | ctx | P; qstat as above; no qkv tensor
  bool projected = false;    // form the latest forward took (latched: the backward must match it whatever the knobs say by then)
};

// ---- hipGraph replay ---------------------------------------------------------------------------------------
// The launch sequence of a forward or backward pass is a pure function of (batch size, mode, knobs, every pointer the kernels
// receive): the second time a pass is asked for with the same key it is stream-captured, afterwards it is ONE hipGraphLaunch per
// segment instead of ~200 launches.  A backward pass under the data-parallel overlap is cut into one segment per gradient phase so
// that the caller's (external) phase events are recorded for real between them.
struct GraphSeg {
  hipGraph_t graph = nullptr;
  hipGraphExec_t exec = nullptr;
  hipEvent_t ev_after = nullptr;   // external event recorded on the launch stream after this segment (phase events)
  long long kernels = 0;
};
// host-side record of a forward pass that its backward walks ("tape" pointers inside the workspace + the arena plan)
struct TapeState {
  std::vector<ResBlock> rb;
  std::vector<AttnBlock> attn;
  std::vector<const float*> skip, down_in, up_in;
  int tape_B = 0;
  const float* x_in = nullptr;
  float *emb = nullptr, *h1 = nullptr, *h1g = nullptr, *temb = nullptr, *st = nullptr, *ss = nullptr, *h0 = nullptr, *xfinal = nullptr;
  const float* out_nchw = nullptr;
  bool tape_cond = false;
  const float* cond_in = nullptr;
  float *e1 = nullptr, *e1g = nullptr, *e2 = nullptr, *h0pre = nullptr;
  int plan_B = 0;
  size_t plan_tape_b = 0, plan_tmp_b = 0, plan_defer_b = 0;
};
struct GraphEntry {
  std::vector<uint64_t> key;
  std::vector<GraphSeg> segs;
  TapeState tape;                  // forward entries (training mode): the host state the pass leaves behind
  const void* red_dev = nullptr;   // backward entries: where the pass expects its reduction descriptor table on the device
  uint64_t stamp = 0;
};
struct GraphCapture {              // an open capture
  GraphEntry* entry = nullptr;
  hipStream_t user_st = nullptr;   // stream the finished segments are launched on
  bool open = false, failed = false;
  long long kernels0 = 0;
};
static const size_t kMaxGraphs = 6;      // per handle and pass kind (LRU)
static const size_t kMaxSeenKeys = 16;
static long long g_graph_launches = 0, g_graph_kernels = 0, g_graph_captures = 0;
static void graph_entry_free(GraphEntry& e) {
  for (auto& sg : e.segs) {
    if (sg.exec) (void)hipGraphExecDestroy(sg.exec);
    if (sg.graph) (void)hipGraphDestroy(sg.graph);
  }
  e.segs.clear();
}

}  // namespace pidm

using namespace pidm;

struct pidm_unet {
  pidm_unet_cfg cfg;
  std::vector<std::string> names;
  std::vector<size_t> numels;
  std::map<std::string, int> index;
  std::vector<const float*> P;
  std::vector<float*> G;
  bool have_grads = false;

  int tdim = 0, ss_total = 0, n_lv = 0, heads = 8, groups = 8;
  std::vector<int> dims;  // [init_dim, dim*m0, dim*m1, ...]
  ConvLayer init_conv, lin1, lin2, lincat, final_conv;
  // gradient-guidance conditioning branch (src/unet_model.py:521-528,571-587): x = combine_conv(cat(init_conv(x),
  // emb_conv(cond))).  Its 6 parameters are the LAST entries of the canonical list; they are used (and get gradients)
  // only by a forward that was given a conditioning field (pidm_unet_set_condition).
  ConvLayer emb1, emb2, comb;
  int cond_first_param = -1;
  bool cond_enabled = false;          // workspace sized for the conditioning branch
  const float* cond_next = nullptr;   // conditioning input of the NEXT forward (consumed by it)
  bool cond_grads_dirty = true;       // the conditioning gradient slots may hold non-zero values
  // backward overlap: weight-gradient kernels (nothing inside backward depends on them before the final reduction) run on
  // a side stream while the input-gradient chain continues on the caller's stream; fork/join through two events
  hipStream_t side = nullptr;
  hipEvent_t ev_fork = nullptr, ev_join = nullptr;
  bool side_ok = false;
  std::vector<ResBlock> rb;       // order: downs (2 per level), mid1, mid2, ups (2 per level), final
  std::vector<AttnBlock> attn;    // order: downs (1 per level), mid, ups (1 per level)
  std::vector<ConvLayer> down, up;
  size_t packed_floats_total = 0;
  const void* packed_zeroed_for = nullptr;
  std::vector<PackDesc> pack_table;   // host copy of the device-side descriptor table (lives after the packed weights)
  unsigned pack_blocks = 0;
  bool pack_failed = false;
  bool pack_table_valid = false;

  // tape of the latest forward
  int tape_B = 0;
  const float* x_in = nullptr;
  float *emb = nullptr, *h1 = nullptr, *h1g = nullptr, *temb = nullptr, *st = nullptr, *ss = nullptr, *h0 = nullptr;
  float* xfinal = nullptr;
  const float* out_nchw = nullptr;
  bool tape_cond = false;
  const float* cond_in = nullptr;
  float *e1 = nullptr, *e1g = nullptr, *e2 = nullptr, *h0pre = nullptr;
  std::vector<const float*> skip;     // per level
  std::vector<const float*> down_in;  // input of each downsample
  std::vector<const float*> up_in;    // input of each upsample
  std::map<int, std::pair<size_t, size_t>> ws_cache[2];  // B -> (tape bytes, tmp bytes) for inference/training
  std::map<int, size_t> defer_cache;                     // B -> bytes of the deferred-reduction arena (training)
  long knob_sig = 0;                                     // attention-form knobs the cached plans were made under
  // arena plan of the latest training-mode forward: its backward lays the workspace out the same way even if the knobs (and
  // with them the cached plans) changed in between
  int plan_B = 0;
  size_t plan_tape_b = 0, plan_tmp_b = 0, plan_defer_b = 0;
  std::vector<WgradItem> wq_table[kWgFams];              // host copies of the last uploaded grouped-weight-gradient tables
  const void* wq_table_dev = nullptr;                    // (invalidated together with red_table_dev: same arena region)
  std::vector<ReduceDesc> red_table;                     // host copy of the last uploaded reduction table
  const void* red_table_dev = nullptr;
  // data-parallel overlap (pidm_unet_set_grad_events): the deferred gradient reduction runs in up to 3 phases - after the
  // decoder half, after the encoder half, at the end - and an event is recorded after each, so the caller's collective of
  // the parameters that phase finalises can start while the rest of backward still runs
  int n_phases = 1;
  hipEvent_t phase_ev[3] = {nullptr, nullptr, nullptr};
  int idx_downs_first = 0, idx_ups_first = 0;            // canonical parameter indices that bound the phases
  // hipGraph replay
  hipStream_t cap_stream = nullptr;                      // private capture stream (the caller's may be the null stream, which cannot capture)
  bool cap_stream_ok = false;
  std::vector<GraphEntry> graphs[2];                     // [0] forward, [1] backward
  std::vector<std::vector<uint64_t>> seen[2];            // keys seen once (captured on their second sighting)
  uint64_t graph_stamp = 0, bind_sig = 0, param_sig = 0, fwd_key_hash = 0;
  uint64_t arena_sig = 0;                                // layout of the latest pass that used the workspace arena
  uint64_t red_table_owner = 0;                          // layout that uploaded the reduction descriptor table
};

namespace pidm {

static const size_t kMaxPackDesc = 1024;
static const size_t kMaxReduceDesc = 1024;
static long long g_red_table_uploads = 0;   // pidm_debug_reduce_table_uploads(): steady state must not upload (tests)

// The backward pass is enqueued launch by launch - the only mode in which weight gradients can go to the side stream - with
// PIDM_GRAPH=0, with PIDM_GRAPH_BWD=0 (the backward pass alone; =1: always replayed) and, by default, for WIDE models (>= 512
// channels at the deepest level: the mechanics configuration, dim 128 x 8).  Measured (tools/archive/r03_r.sh, same box): their deep
// levels have fewer work items than the chip has CUs, a concurrent weight-gradient kernel fills them - mechanics 798.8 -> 815.4
// samples/s (main.py's loop unchanged: 750.0 -> 762.6); the Darcy model (256 channels, full launches) ties: 6018 / 6021 / 6037
// for graph / forward-graph + eager backward / all eager at batch 64, 7999 / 8077 / 7988 at batch 256 - it keeps the replay, which
// costs the host a third of the launch-by-launch work.
// Grouped weight gradients (round 5): PIDM_WGRAD_GROUP = queued problems after which the grouped launches run (0: off - every
// problem its own launch, as before; default kWgradGroupDefault; a launch takes at most 64 rows: the kernels' table lookup is one
// row per lane).  Three kernel families - 3x3 / stride-1 row-streaming, 4x4 / stride-2 row-streaming, 1x1 pixel streams - with a
// table and a launch each.  A flush also happens before every deferred reduction (the gradient phases of the data-parallel exchange).
static const int kWgradGroupDefault = 1 << 20;   // = one flush per deferred reduction (per gradient phase)
static const size_t kMaxWgradItems = 128;      // rows of a family's device table (problems of one backward pass)
// Which passes group: the linear (graph-replayed) ones - the wide models keep their launch-by-launch backward with the weight
// gradients on the side stream (measured faster there, see backward_eager) - up to PIDM_WGRAD_GROUP_MAXWORK batch x image pixels.
// Measured per step on one box (profiles/r05_d_wgrad_group_maxpix.txt, r05_c_wgrad_group_fams.txt): batch 16 5.07 -> 4.67 ms,
// batch 64 8.36 -> 7.99, batch 256 24.9 -> 25.1, batch 512 48.3 -> 49.2-49.8: at large batches every problem fills the chip with
// long work items by itself, runs right behind the kernel that produced its dY (Infinity-Cache hits) and the backward arena's
// frames are recycled instead of kept to the end of the pass.
static const long kWgradGroupMaxWork = 128L * 64 * 64;
static bool wgrad_group_on(int B, int image, bool side_allowed) {
  const char* e = knob("PIDM_WGRAD_GROUP_MAXWORK");
  const long maxw = e ? atol(e) : kWgradGroupMaxWork;
  const char* g = knob("PIDM_WGRAD_GROUP");
  return !(g && atoi(g) <= 0) && !side_allowed && (long)B * image * image <= maxw;
}
static int wgrad_group_limit() {
  const char* e = knob("PIDM_WGRAD_GROUP");
  int n = e ? atoi(e) : kWgradGroupDefault;
  return n < 0 ? 0 : n;
}
static bool backward_eager(int widest) {
  const char* e = knob("PIDM_GRAPH");
  if (e && !atoi(e)) return true;
  const char* b = knob("PIDM_GRAPH_BWD");
  if (b) return !atoi(b);
  const char* w = knob("PIDM_GRAPH_BWD_WIDE");     // the width threshold (tests reach the rule with small models)
  return widest >= (w && atoi(w) > 0 ? atoi(w) : 512);
}
struct Run {
  pidm_unet* U;
  int B;
  bool train, dry;
  hipStream_t st;
  float* wpack;
  Arena tape, tmp;
  float* scratch = nullptr;  // shared scratch for wgrad partials / norm reductions
  size_t scratch_floats = 0;
  // backward: split-K / per-sample partial buffers that must outlive their producer are taken from `defer` and their
  // fixed-order sums are queued in `rq`, run by ONE reduce_multi launch at the end of backward
  Arena defer;
  ReduceQueue rq;
  WgradQueue wq;               // weight-gradient problems waiting for their grouped launch (group_on), per kernel family
  WgradItem* wq_dev = nullptr; // device tables (deferred arena, behind the reduction table): kWgFams x kMaxWgradItems rows
  bool group_on = false;
  bool defer_on = false;
  bool overlap = false;        // weight gradients on the side stream (real backward runs only)
  // Backward arena frames are kept to the end of the pass while side-stream weight gradients may still read them; the sizing dry
  // run assumes that only where a real pass can take the side stream at all (PIDM_GRAPH=0: captured passes are linear, and with
  // graph replay on - the default - the two eager passes before the capture are linear too), otherwise frames are recycled and
  // the plan is the smaller one.
  // (... and while queued weight-gradient problems may still read them: group_on)
  bool keep_frames() const { return overlap || group_on || (dry && side_allowed); }
  bool side_allowed = false;   // backward_eager(widest level) of this handle (set by setup / the dry run)
  bool side_pending = false;   // side-stream work issued since the last join
  GraphCapture* cap = nullptr; // non-null while this pass is being stream-captured (r.st is the capture stream then)
  float* part_alloc(size_t bytes) { return defer_on ? defer.alloc(bytes / 4 + 64) : scratch; }
  ReduceQueue* q() { return defer_on ? &rq : nullptr; }
  // (a family's device table holds kMaxWgradItems rows per pass: once one is full the remaining problems of the pass get their own
  // launches - same decision in the dry run, which queues exactly as the real pass does)
  WgradQueue* wgq() {
    if (!(defer_on && group_on)) return nullptr;
    for (const auto& q : wq.f)
      if (q.v.size() >= kMaxWgradItems) return nullptr;
    return &wq;
  }
};

#define RUN(call)                 \
  do {                            \
    if (!r.dry) {                 \
      int rc__ = (call);          \
      if (rc__) return rc__;      \
    }                             \
  } while (0)

static int add_param(pidm_unet* U, const std::string& name, size_t numel) {
  U->index[name] = (int)U->names.size();
  U->names.push_back(name);
  U->numels.push_back(numel);
  return (int)U->names.size() - 1;
}

static pidm_conv_desc desc_of(const ConvLayer& L, int B) {
  pidm_conv_desc d;
  memset(&d, 0, sizeof(d));
  d.B = B; d.Hi = L.H; d.Wi = L.H; d.C0 = L.C0; d.C1 = L.C1; d.ld0 = L.C0; d.ld1 = L.C1; d.Cout = L.Cout;
  d.KH = d.KW = L.K; d.stride = L.stride; d.pad = L.pad; d.transposed = L.transposed; d.out_nchw = 0; d.ldo = L.Cout;
  return d;
}

static int geom_fwd_layer(const ConvLayer& L, int B, int out_nchw, ConvGeom* g) {
  if (L.transposed) return make_geom(g, 1, B, L.H, L.H, L.C0, L.C1, L.C0, L.C1, L.Cout, 4, 4, 2, 1, out_nchw, L.Cout, L.Cout);
  return make_geom(g, 0, B, L.H, L.H, L.C0, L.C1, L.C0, L.C1, L.Cout, L.K, L.K, L.stride, L.pad, out_nchw, L.Cout, L.Cout);
}

static int out_h(const ConvLayer& L) { return L.transposed ? 2 * L.H : (L.H + 2 * L.pad - L.K) / L.stride + 1; }

static void conv_param(pidm_unet* U, ConvLayer& L, const std::string& pre, bool bias) {
  const int Cin = L.C0 + L.C1;
  L.w = add_param(U, pre + ".weight", (size_t)L.Cout * Cin * L.K * L.K);
  L.b = bias ? add_param(U, pre + ".bias", (size_t)L.Cout) : -1;
}

static int reserve_packed(pidm_unet* U, ConvLayer& L) {
  ConvGeom g;
  if (geom_fwd_layer(L, 1, 0, &g)) return -1;
  L.off_f = U->packed_floats_total;
  U->packed_floats_total += align_up(packed_floats(g), 64);
  if (L.dgrad) {
    pidm_conv_desc d = desc_of(L, 1);
    int kind;
    if (geom_dgrad(&d, L.Cout, L.C0 + L.C1, &g, &kind)) return -1;
    L.off_d = U->packed_floats_total;
    U->packed_floats_total += align_up(packed_floats(g), 64);
  }
  return 0;
}

static void make_resblock(pidm_unet* U, ResBlock& m, const std::string& pre, int C0, int C1, int Co, int H, bool mlp) {
  m.C0 = C0; m.C1 = C1; m.Co = Co; m.H = H; m.has_mlp = mlp; m.has_res = (C0 + C1 != Co);
  if (mlp) {
    m.mlpw = U->index.at(pre + "mlp.1.weight");
    m.mlpb = U->index.at(pre + "mlp.1.bias");
  }
  m.c1.C0 = C0; m.c1.C1 = C1; m.c1.Cout = Co; m.c1.K = 3; m.c1.pad = 1; m.c1.H = H;
  conv_param(U, m.c1, pre + "block1.proj", true);
  m.gn1w = add_param(U, pre + "block1.norm.weight", Co);
  m.gn1b = add_param(U, pre + "block1.norm.bias", Co);
  m.c2.C0 = Co; m.c2.Cout = Co; m.c2.K = 3; m.c2.pad = 1; m.c2.H = H;
  conv_param(U, m.c2, pre + "block2.proj", true);
  m.gn2w = add_param(U, pre + "block2.norm.weight", Co);
  m.gn2b = add_param(U, pre + "block2.norm.bias", Co);
  if (m.has_res) {
    m.cr.C0 = C0; m.cr.C1 = C1; m.cr.Cout = Co; m.cr.K = 1; m.cr.H = H;
    conv_param(U, m.cr, pre + "res_conv", true);
  }
}

static void make_attn(pidm_unet* U, AttnBlock& a, const std::string& pre, int C, int H, bool mid) {
  a.mid = mid; a.C = C; a.H = H;
  const int HD = U->heads * 32;
  const std::string f = mid ? pre + "fn.fn.fn." : pre + "fn.fn.";
  a.qkv.C0 = C; a.qkv.Cout = 3 * HD; a.qkv.K = 1; a.qkv.H = H;
  conv_param(U, a.qkv, f + "to_qkv", false);
  a.out.C0 = HD; a.out.Cout = C; a.out.K = 1; a.out.H = H;
  conv_param(U, a.out, f + "to_out", !mid);
  a.gamma = add_param(U, pre + "fn.norm.gamma", C);
}

}  // namespace pidm

extern "C" int pidm_unet_create(const pidm_unet_cfg* cfg, pidm_unet** out) {
  if (!cfg || !out) return fail("unet_create: null argument");
  if (cfg->dim_head != 32) return fail("unet_create: dim_head must be 32 (got %d)", cfg->dim_head);
  if (cfg->n_levels < 1 || cfg->n_levels > 8) return fail("unet_create: bad n_levels");
  if (cfg->dim % 8 || cfg->dim < 8) return fail("unet_create: dim must be a multiple of 8");
  const int P = cfg->image_size;
  if (P <= 0 || (P & (P - 1)) || (P >> (cfg->n_levels - 1)) < 1 || P > 128) return fail("unet_create: image_size %d must be a power of two <= 128", P);
  pidm_unet* U = new pidm_unet();
  U->cfg = *cfg;
  U->n_lv = cfg->n_levels;
  U->heads = cfg->heads;
  U->groups = cfg->groups;
  U->tdim = 4 * cfg->dim;
  const int dim = cfg->dim, n = U->n_lv, td = U->tdim;
  U->dims.push_back(dim);
  for (int i = 0; i < n; ++i) U->dims.push_back(dim * cfg->dim_mults[i]);
  std::vector<int> res(n);
  for (int i = 0; i < n; ++i) res[i] = P >> i;

  // ---- canonical parameter order: the 18 FiLM linears first (weights, then biases) so that their gradients
  //      are one contiguous [sum 2*Cout][tdim] matrix when the caller lays grads out in this order ----
  std::vector<std::pair<std::string, int>> film;  // (prefix, Cout)
  for (int i = 0; i < n; ++i) {
    film.push_back({"downs." + std::to_string(i) + ".0.", U->dims[i + 1]});
    film.push_back({"downs." + std::to_string(i) + ".1.", U->dims[i + 1]});
  }
  film.push_back({"mid_block1.", U->dims[n]});
  film.push_back({"mid_block2.", U->dims[n]});
  for (int j = 0; j < n; ++j) {
    const int din = U->dims[n - 1 - j];
    film.push_back({"ups." + std::to_string(j) + ".0.", din});
    film.push_back({"ups." + std::to_string(j) + ".1.", din});
  }
  std::map<std::string, int> ss_off;
  int off = 0;
  for (auto& f : film) {
    add_param(U, f.first + "mlp.1.weight", (size_t)2 * f.second * td);
    ss_off[f.first] = off;
    off += 2 * f.second;
  }
  U->ss_total = off;
  for (auto& f : film) add_param(U, f.first + "mlp.1.bias", (size_t)2 * f.second);

  U->init_conv.C0 = cfg->channels * (cfg->self_condition ? 2 : 1); U->init_conv.Cout = dim; U->init_conv.K = cfg->init_kernel;
  U->init_conv.pad = cfg->init_kernel / 2; U->init_conv.H = P; U->init_conv.dgrad = true;
  conv_param(U, U->init_conv, "init_conv", true);
  U->lin1.C0 = dim; U->lin1.Cout = td; conv_param(U, U->lin1, "time_mlp.1", true);
  U->lin2.C0 = td; U->lin2.Cout = td; conv_param(U, U->lin2, "time_mlp.3", true);
  U->lincat.C0 = td; U->lincat.Cout = U->ss_total;  // virtual layer: weights gathered from the FiLM linears

  U->rb.resize(4 * n + 3);
  U->attn.resize(2 * n + 1);
  U->down.resize(n);
  U->up.resize(n);
  int irb = 0, iat = 0;
  U->idx_downs_first = (int)U->names.size();
  for (int i = 0; i < n; ++i) {
    const std::string pre = "downs." + std::to_string(i) + ".";
    const int din = U->dims[i], dout = U->dims[i + 1];
    make_resblock(U, U->rb[irb], pre + "0.", din, 0, dout, res[i], true); U->rb[irb++].ss_off = ss_off[pre + "0."];
    make_resblock(U, U->rb[irb], pre + "1.", dout, 0, dout, res[i], true); U->rb[irb++].ss_off = ss_off[pre + "1."];
    make_attn(U, U->attn[iat++], pre + "2.", dout, res[i], false);
    if (i < n - 1) {
      ConvLayer& d = U->down[i];
      d.C0 = dout; d.Cout = dout; d.K = 4; d.stride = 2; d.pad = 1; d.H = res[i];
      conv_param(U, d, pre + "3", true);
    }
  }
  make_resblock(U, U->rb[irb], "mid_block1.", U->dims[n], 0, U->dims[n], res[n - 1], true); U->rb[irb++].ss_off = ss_off["mid_block1."];
  make_attn(U, U->attn[iat++], "mid_spatial_attn.", U->dims[n], res[n - 1], true);
  make_resblock(U, U->rb[irb], "mid_block2.", U->dims[n], 0, U->dims[n], res[n - 1], true); U->rb[irb++].ss_off = ss_off["mid_block2."];
  U->idx_ups_first = (int)U->names.size();
  for (int j = 0; j < n; ++j) {
    const std::string pre = "ups." + std::to_string(j) + ".";
    const int lvl = n - 1 - j;
    const int din = U->dims[lvl], dout = U->dims[lvl + 1];
    make_resblock(U, U->rb[irb], pre + "0.", dout, dout, din, res[lvl], true); U->rb[irb++].ss_off = ss_off[pre + "0."];
    make_resblock(U, U->rb[irb], pre + "1.", din, 0, din, res[lvl], true); U->rb[irb++].ss_off = ss_off[pre + "1."];
    make_attn(U, U->attn[iat++], pre + "2.", din, res[lvl], false);
    if (j < n - 1) {
      ConvLayer& u = U->up[j];
      u.C0 = din; u.Cout = din; u.K = 4; u.stride = 2; u.pad = 1; u.transposed = 1; u.H = res[lvl];
      conv_param(U, u, pre + "3", true);
    }
  }
  make_resblock(U, U->rb[irb], "final_conv.0.", dim, dim, dim, P, false); irb++;
  U->final_conv.C0 = dim; U->final_conv.Cout = cfg->out_dim; U->final_conv.K = 1; U->final_conv.H = P;
  conv_param(U, U->final_conv, "final_conv.1", true);
  U->cond_first_param = (int)U->names.size();
  U->emb1.C0 = cfg->channels; U->emb1.Cout = dim; U->emb1.K = 1; U->emb1.H = P; U->emb1.dgrad = false;
  conv_param(U, U->emb1, "emb_conv.0", true);
  U->emb2.C0 = dim; U->emb2.Cout = dim; U->emb2.K = 3; U->emb2.pad = 1; U->emb2.H = P;
  conv_param(U, U->emb2, "emb_conv.2", true);
  U->comb.C0 = dim; U->comb.C1 = dim; U->comb.Cout = dim; U->comb.K = 1; U->comb.H = P;
  conv_param(U, U->comb, "combine_conv", true);

  // ---- packed-weight region ----
  int rc = 0;
  rc |= reserve_packed(U, U->init_conv);
  rc |= reserve_packed(U, U->lin1);
  rc |= reserve_packed(U, U->lin2);
  rc |= reserve_packed(U, U->lincat);
  for (auto& m : U->rb) {
    rc |= reserve_packed(U, m.c1);
    rc |= reserve_packed(U, m.c2);
    if (m.has_res) rc |= reserve_packed(U, m.cr);
  }
  for (auto& a : U->attn) {
    rc |= reserve_packed(U, a.qkv);
    rc |= reserve_packed(U, a.out);
  }
  for (int i = 0; i < n - 1; ++i) {
    rc |= reserve_packed(U, U->down[i]);
    rc |= reserve_packed(U, U->up[i]);
  }
  rc |= reserve_packed(U, U->final_conv);
  rc |= reserve_packed(U, U->emb1);
  rc |= reserve_packed(U, U->emb2);
  rc |= reserve_packed(U, U->comb);
  if (rc) {
    delete U;
    return -1;
  }
  U->P.assign(U->names.size(), nullptr);
  U->G.assign(U->names.size(), nullptr);
  U->skip.assign(n, nullptr);
  U->down_in.assign(n, nullptr);
  U->up_in.assign(n, nullptr);
  *out = U;
  return 0;
}

extern "C" void pidm_unet_destroy(pidm_unet* h) {
  if (!h) return;
  for (int k = 0; k < 2; ++k)
    for (auto& e : h->graphs[k]) pidm::graph_entry_free(e);
  if (h->cap_stream_ok) (void)hipStreamDestroy(h->cap_stream);
  if (h->side_ok) {
    (void)hipStreamDestroy(h->side);
    (void)hipEventDestroy(h->ev_fork);
    (void)hipEventDestroy(h->ev_join);
  }
  delete h;
}
extern "C" int pidm_unet_num_params(const pidm_unet* h) { return (int)h->names.size(); }
extern "C" const char* pidm_unet_param_name(const pidm_unet* h, int i) {
  return (i >= 0 && i < (int)h->names.size()) ? h->names[i].c_str() : "";
}
extern "C" size_t pidm_unet_param_numel(const pidm_unet* h, int i) {
  return (i >= 0 && i < (int)h->numels.size()) ? h->numels[i] : 0;
}

extern "C" int pidm_unet_bind(pidm_unet* h, const void* const* param_ptrs_host, void* const* grad_ptrs_host) {
  if (!h || !param_ptrs_host) return fail("unet_bind: null argument");
  for (size_t i = 0; i < h->names.size(); ++i) {
    if (!param_ptrs_host[i]) return fail("unet_bind: parameter %s is null", h->names[i].c_str());
    h->P[i] = reinterpret_cast<const float*>(param_ptrs_host[i]);
    h->G[i] = grad_ptrs_host ? reinterpret_cast<float*>(grad_ptrs_host[i]) : nullptr;
  }
  h->have_grads = grad_ptrs_host != nullptr;
  bool keep_pack_table = false;
  {
    // graphs are keyed on the bound pointer sets (flipping back to an earlier set - EMA swap in / out, alternating gradient
    // buffers - finds its graphs again): the forward only on the parameters, the backward on parameters and gradients
    uint64_t ps = 1469598103934665603ull, gs = 1099511628211ull;
    bool params_changed = false;
    for (size_t i = 0; i < h->names.size(); ++i) {
      ps ^= (uint64_t)reinterpret_cast<uintptr_t>(h->P[i]); ps *= 1099511628211ull; ps ^= ps >> 29;
      gs ^= (uint64_t)reinterpret_cast<uintptr_t>(h->G[i]); gs *= 1099511628211ull; gs ^= gs >> 29;
    }
    params_changed = ps != h->param_sig;
    h->param_sig = ps;
    h->bind_sig = ps ^ (gs * 0x9E3779B97F4A7C15ull);
    if (!params_changed && h->pack_table_valid) keep_pack_table = true;   // same parameters: the device-side pack table still describes them
  }
  h->cond_grads_dirty = true;    // new gradient buffers: contents unknown
  if (!keep_pack_table) h->pack_table_valid = false;   // parameter pointers changed
  if (h->have_grads) {
    // the FiLM linear gradients must be contiguous (see pidm_unet_create)
    const int nf = 4 * h->n_lv + 2;
    for (int i = 1; i < nf; ++i) {
      if (h->G[i] != h->G[i - 1] + h->numels[i - 1]) return fail("unet_bind: gradients of the %d FiLM linear weights must be contiguous in canonical order", nf);
      if (h->G[nf + i] != h->G[nf + i - 1] + h->numels[nf + i - 1]) return fail("unet_bind: gradients of the FiLM linear biases must be contiguous");
    }
  }
  return 0;
}

namespace pidm {

// ------------------------------------------------------------------------------------------------------
// weight packing (once per step)
// ------------------------------------------------------------------------------------------------------
static void push_desc(Run& r, const ConvGeom& g, int kind, const float* src, float* dst, int K, int n_off, int k_off, int n_src,
                      int k_src) {
  PackDesc d;
  if (!make_pack_desc(g, kind, src, dst, K, K, n_off, k_off, n_src, k_src, &d)) r.U->pack_failed = true;   // (message set by fail())
  d.blk0 = r.U->pack_blocks;
  r.U->pack_blocks += d.nblk;
  r.U->pack_table.push_back(d);
}

static int pack_layer(Run& r, const ConvLayer& L) {
  pidm_unet* U = r.U;
  ConvGeom g;
  if (geom_fwd_layer(L, 1, 0, &g)) return -1;
  push_desc(r, g, L.transposed ? 1 : 0, U->P[L.w], r.wpack + L.off_f, L.K, 0, 0, 0, 0);
  if (L.dgrad) {
    pidm_conv_desc d = desc_of(L, 1);
    int kind;
    if (geom_dgrad(&d, L.Cout, L.C0 + L.C1, &g, &kind)) return -1;
    push_desc(r, g, kind, U->P[L.w], r.wpack + L.off_d, L.K, 0, 0, 0, 0);
  }
  return 0;
}

static int pack_all(Run& r) {
  pidm_unet* U = r.U;
  PackDesc* table_dev = reinterpret_cast<PackDesc*>(r.wpack + U->packed_floats_total);
  if (U->packed_zeroed_for == r.wpack && U->pack_table_valid) {
    // same workspace, same parameter pointers: the descriptor table on the device is still valid
    return launch_pack_multi(table_dev, (int)U->pack_table.size(), U->pack_blocks, r.st);
  }
  if (r.cap) {        // needs a host -> device upload of the descriptor table: never captured, the pass is re-run eagerly
    r.cap->failed = true;
    return 0;
  }
  if (hipMemsetAsync(r.wpack, 0, U->packed_floats_total * sizeof(float), r.st) != hipSuccess) return fail("memset failed");
  U->packed_zeroed_for = r.wpack;
  U->pack_table.clear();
  U->pack_blocks = 0;
  U->pack_failed = false;
  int rc = 0;
  rc |= pack_layer(r, U->init_conv);
  rc |= pack_layer(r, U->lin1);
  rc |= pack_layer(r, U->lin2);
  // concatenated FiLM linear: forward rows [ss_off, ss_off+2C), dgrad columns likewise
  {
    ConvGeom gf, gd;
    if (geom_fwd_layer(U->lincat, 1, 0, &gf)) return -1;
    pidm_conv_desc d = desc_of(U->lincat, 1);
    int kind;
    if (geom_dgrad(&d, U->lincat.Cout, U->lincat.C0, &gd, &kind)) return -1;
    for (auto& m : U->rb) {
      if (!m.has_mlp) continue;
      push_desc(r, gf, 0, U->P[m.mlpw], r.wpack + U->lincat.off_f, 1, m.ss_off, 0, 2 * m.Co, U->tdim);
      push_desc(r, gd, kind, U->P[m.mlpw], r.wpack + U->lincat.off_d, 1, 0, m.ss_off, U->tdim, 2 * m.Co);
    }
  }
  for (auto& m : U->rb) {
    rc |= pack_layer(r, m.c1);
    rc |= pack_layer(r, m.c2);
    if (m.has_res) rc |= pack_layer(r, m.cr);
  }
  for (auto& a : U->attn) {
    rc |= pack_layer(r, a.qkv);
    rc |= pack_layer(r, a.out);
  }
  for (int i = 0; i < U->n_lv - 1; ++i) {
    rc |= pack_layer(r, U->down[i]);
    rc |= pack_layer(r, U->up[i]);
  }
  rc |= pack_layer(r, U->final_conv);
  rc |= pack_layer(r, U->emb1);
  rc |= pack_layer(r, U->emb2);
  rc |= pack_layer(r, U->comb);
  if (rc) return rc;
  if (U->pack_failed) return -1;
  if (U->pack_table.size() > kMaxPackDesc) return fail("pack: descriptor table overflow");
  if (hipMemcpyAsync(table_dev, U->pack_table.data(), U->pack_table.size() * sizeof(PackDesc), hipMemcpyHostToDevice, r.st) != hipSuccess)
    return fail("pack: descriptor upload failed");
  U->pack_table_valid = true;
  return launch_pack_multi(table_dev, (int)U->pack_table.size(), U->pack_blocks, r.st);
}

// ------------------------------------------------------------------------------------------------------
// forward building blocks
// ------------------------------------------------------------------------------------------------------
static float* act_alloc(Run& r, size_t n) { return r.train ? r.tape.alloc(n) : r.tmp.alloc(n); }
static bool knob_on(const char* name) {
  const char* e = knob(name);
  return e && atoi(e);
}

static int conv_fwd(Run& r, const ConvLayer& L, const float* x0, const float* x1, const float* residual, float* out,
                    int out_nchw = 0, int sigmoid_last = 0) {
  ConvGeom g;
  if (geom_fwd_layer(L, r.B, out_nchw, &g)) return -1;
  RUN(launch_conv(g, x0, x1, r.wpack + L.off_f, L.b >= 0 ? r.U->P[L.b] : nullptr, residual, out, sigmoid_last, r.st));
  return 0;
}

// Convolution whose output feeds a GroupNorm: the epilogue leaves the per-(image, 32-pixel chunk, group) sums in r.scratch and
// gn_apply finalises them - one launch and one full read of the activation less than conv -> gn_stats -> gn_apply.
// *part_chunks = chunks per image written (0: not produced - shape not eligible or kernel without the epilogue; the caller
// then runs launch_gn_stats).
static int conv_fwd_gn(Run& r, const ConvLayer& L, const float* x0, const float* x1, float* out, int G, int* part_chunks) {
  const bool off = knob("PIDM_NO_GN_EPILOGUE") != nullptr;
  *part_chunks = 0;
  ConvGeom g;
  if (geom_fwd_layer(L, r.B, 0, &g)) return -1;
  const int HW = g.Ho * g.Wo, cpg = (G > 0 && L.Cout % G == 0) ? L.Cout / G : 0;
  int chunks = HW / 32;               // (the row-streaming kernel writes one chunk per strip: ConvGeom::part_chunks_out)
  const bool ok = !off && L.Cout % 32 == 0 && cpg >= 4 && cpg <= 32 && (cpg & (cpg - 1)) == 0 && HW % 32 == 0 && g.nz == 1 &&
                  g.nph == 1 && g.os == 1 && g.soc == 1 && g.KH == 3 && (32 % g.Wv == 0 || g.Wv % 32 == 0);
  if (ok) {
    g.gn_part = reinterpret_cast<double*>(r.scratch);
    g.gn_cpg = cpg;
    g.gn_G = G;
    g.gn_nchunk = HW / 32;
    g.part_chunks_out = &chunks;
  }
  if (r.dry) return 0;
  const int rc = launch_conv(g, x0, x1, r.wpack + L.off_f, L.b >= 0 ? r.U->P[L.b] : nullptr, nullptr, out, 0, r.st);
  if (rc < 0) return rc;
  if (ok && rc == 0) *part_chunks = chunks;
  return 0;
}

// next_attn: the attention block that consumes this block's output.  Its PreNorm LayerNorm rides in the block's last gn_apply
// (one launch and one read of the activation less per attention block; PIDM_NO_GN_LN_FUSE=1: off) when the block ends in that
// kernel (no res_conv) and a pixel's channels fit one wave.
static int resblock_fwd(Run& r, ResBlock& m, const float* x0, const float* x1, float** out_p, AttnBlock* next_attn = nullptr) {
  pidm_unet* U = r.U;
  const int B = r.B, HW = m.H * m.H, Co = m.Co, G = U->groups;
  const size_t n = (size_t)B * HW * Co;
  float* out = act_alloc(r, n);
  float* ln_out = nullptr;
  if (next_attn) {
    next_attn->xn_ready = false;
    const char* nf = knob("PIDM_NO_GN_LN_FUSE");
    if (!m.has_res && next_attn->C == Co && next_attn->H == m.H && gn_apply_ln_ok(Co) && !(nf && atoi(nf))) {
      next_attn->out_pre = act_alloc(r, n);
      next_attn->mk_pre = r.tmp.mark();
      ln_out = act_alloc(r, n);
      next_attn->xn = ln_out;
      next_attn->xn_ready = true;
    }
  }
  const size_t mk = r.tmp.mark();
  m.x0 = x0; m.x1 = x1;
  m.a = act_alloc(r, n);
  int pc1 = 0, pc2 = 0;
  if (conv_fwd_gn(r, m.c1, x0, x1, m.a, G, &pc1)) return -1;
  m.st1 = act_alloc(r, (size_t)B * G * 2);
  if (!pc1) RUN(launch_gn_stats(m.a, B, HW, Co, G, m.st1, r.scratch, r.st));
  // inference (no tape): GroupNorm + FiLM + SiLU in place - half the footprint of the pass in the caches, a smaller workspace
  // (PIDM_NO_GN_INPLACE=1: off)
  const bool inplace = !r.train && !knob_on("PIDM_NO_GN_INPLACE");
  m.bact = inplace ? m.a : act_alloc(r, n);
  const float* ss = m.has_mlp ? U->ss + m.ss_off : nullptr;
  const float* ssb = m.has_mlp ? U->P[m.mlpb] : nullptr;
  RUN(launch_gn_apply(m.a, m.st1, U->P[m.gn1w], U->P[m.gn1b], ss, ssb, U->ss_total, nullptr, m.bact, B, HW, Co, G, r.scratch, r.st, pc1));
  m.c = (inplace && !m.has_res) ? out : act_alloc(r, n);      // (in place: the second convolution writes the block's output buffer)
  if (conv_fwd_gn(r, m.c2, m.bact, nullptr, m.c, G, &pc2)) return -1;
  m.st2 = act_alloc(r, (size_t)B * G * 2);
  if (!pc2) RUN(launch_gn_stats(m.c, B, HW, Co, G, m.st2, r.scratch, r.st));
  if (m.has_res) {
    float* d = inplace ? m.c : r.tmp.alloc(n);
    RUN(launch_gn_apply(m.c, m.st2, U->P[m.gn2w], U->P[m.gn2b], nullptr, nullptr, 0, nullptr, d, B, HW, Co, G, r.scratch, r.st, pc2));
    if (conv_fwd(r, m.cr, x0, x1, d, out)) return -1;
  } else {
    RUN(launch_gn_apply(m.c, m.st2, U->P[m.gn2w], U->P[m.gn2b], nullptr, nullptr, 0, x0, out, B, HW, Co, G, r.scratch, r.st, pc2,
                        ln_out ? U->P[next_attn->gamma] : nullptr, ln_out));
  }
  if (!r.train) r.tmp.release(mk);
  else r.tmp.release(mk);
  *out_p = out;
  return 0;
}

// Linear attention without the qkv tensor (k_attn_proj.hip) where the level is large enough for the recomputation to beat the
// 768-channel round trips: the 64x64 and 32x32 levels of the Darcy model.  PIDM_NO_LAP=1 keeps the qkv form everywhere (A/B
// measurements); PIDM_LAP_MIN_N lowers the gate (the unit tests reach the path with small images).
static void lap_knobs(bool* off, int* min_n) {
  const char* e = knob("PIDM_NO_LAP");       // read per call (a handful of calls per step): tests and A/B runs flip it
  *off = e && atoi(e);
  const char* m = knob("PIDM_LAP_MIN_N");
  *min_n = m ? atoi(m) : 1024;
}
// the knobs as one number: workspace plans are cached per batch size and must be dropped when the knobs change on a live handle
static long lap_knob_signature() {
  bool off; int min_n;
  lap_knobs(&off, &min_n);
  const char* g = knob("PIDM_GRAPH");
  const char* gb = knob("PIDM_GRAPH_BWD");        // (the backward plan depends on the side stream too)
  const char* gw = knob("PIDM_GRAPH_BWD_WIDE");
  return 8 * (off ? -1 : (long)min_n) + (g && !atoi(g) ? 1 : 0) + (gb ? (atoi(gb) ? 2 : 4) : 0) + 1000003L * (gw ? atoi(gw) : 0) +
         7919L * (wgrad_group_on(1, 1, false) ? 1 : 0) +  // (grouped weight gradients keep the backward arena's frames)
         104729L * (knob("PIDM_WGRAD_GROUP_MAXWORK") ? atol(knob("PIDM_WGRAD_GROUP_MAXWORK")) % 65521 : 0) +
         15485863L * (knob("PIDM_LAP_NPER_DIV") ? atol(knob("PIDM_LAP_NPER_DIV")) % 8 : 0);   // (scratch of the pixel-sum kernels)
}
static bool attn_shape_projectable(const AttnBlock& a, int heads) { return !a.mid && a.out.b >= 0 && lap_ok(a.H * a.H, heads, a.C, a.C); }
// decided by the FORWARD (and stored in AttnBlock::projected); the backward replays the stored decision
static bool attn_projected(const AttnBlock& a, int heads) {
  bool off; int min_n;
  lap_knobs(&off, &min_n);
  return !off && a.H * a.H >= min_n && attn_shape_projectable(a, heads);
}

static int attn_fwd(Run& r, AttnBlock& a, const float* x, float** out_p) {
  pidm_unet* U = r.U;
  const int B = r.B, N = a.H * a.H, C = a.C, heads = U->heads, HD = heads * 32;
  const size_t npix = (size_t)B * N;
  float* out = a.xn_ready ? a.out_pre : act_alloc(r, npix * C);
  const size_t mk = a.xn_ready ? a.mk_pre : r.tmp.mark();
  a.x = x;
  if (!a.xn_ready) {
    a.xn = act_alloc(r, npix * C);
    RUN(launch_layernorm_fwd(x, U->P[a.gamma], a.xn, npix, C, r.st));
  }
  a.xn_ready = false;
  a.projected = attn_projected(a, heads);
  if (a.projected) {
    a.qkvb = nullptr;
    a.lsaved = act_alloc(r, lap_saved_floats(B, heads, C));
    a.qstat = act_alloc(r, npix * heads * 2);
    RUN(launch_lap_forward(a.xn, U->P[a.qkv.w], U->P[a.out.w], U->P[a.out.b], x, out, a.lsaved, a.qstat, C, B, N, heads, r.scratch, r.st));
    r.tmp.release(mk);
    *out_p = out;
    return 0;
  }
  a.qkvb = act_alloc(r, npix * 3 * HD);
  if (conv_fwd(r, a.qkv, a.xn, nullptr, nullptr, a.qkvb)) return -1;
  if (!a.mid && (la_fused_ok(N, heads, C, C) && la_fused_pays(B, N))) {
    // attention output x to_out projection in one kernel: the HD-channel attention output is neither stored nor re-read
    // (the fused backward does not need it either)
    a.attn = nullptr;
    a.kstat = act_alloc(r, (size_t)B * HD * 2);
    a.ctx = act_alloc(r, (size_t)B * heads * 1024);
    a.qstat = act_alloc(r, npix * heads * 2);
    RUN(launch_la_forward_fused(a.qkvb, a.kstat, a.ctx, a.qstat, U->P[a.out.w], a.out.b >= 0 ? U->P[a.out.b] : nullptr, x, out, C, B,
                                N, heads, r.scratch, r.st));
    r.tmp.release(mk);
    *out_p = out;
    return 0;
  }
  a.attn = act_alloc(r, npix * HD);
  if (a.mid) {
    RUN(launch_mid_attn(a.qkvb, nullptr, a.attn, B, N, heads, false, r.st));
  } else {
    a.kstat = act_alloc(r, (size_t)B * HD * 2);
    a.ctx = act_alloc(r, (size_t)B * heads * 1024);
    a.qstat = act_alloc(r, npix * heads * 2);
    RUN(launch_la_forward(a.qkvb, a.kstat, a.ctx, a.attn, a.qstat, B, N, heads, r.scratch, r.st));
  }
  if (conv_fwd(r, a.out, a.attn, nullptr, x, out)) return -1;
  r.tmp.release(mk);
  *out_p = out;
  return 0;
}

static int forward_impl(Run& r, const float* x_nhwc, const int64_t* t, float* out_nchw, const float* cond_nhwc) {
  pidm_unet* U = r.U;
  const int B = r.B, P = U->cfg.image_size, dim = U->cfg.dim, n = U->n_lv, td = U->tdim;
  U->tape_B = B;
  U->x_in = x_nhwc;
  U->out_nchw = out_nchw;
  // time path: emb -> lin1 -> GELU -> lin2 -> SiLU -> concatenated FiLM linears (bias added by the consumers)
  U->emb = act_alloc(r, (size_t)B * dim);
  RUN(launch_sinusoid(t, U->emb, B, dim, r.st));
  U->h1 = act_alloc(r, (size_t)B * td);
  if (conv_fwd(r, U->lin1, U->emb, nullptr, nullptr, U->h1)) return -1;
  U->h1g = act_alloc(r, (size_t)B * td);
  RUN(launch_act_fwd(U->h1, U->h1g, (size_t)B * td, 1, r.st));
  U->temb = act_alloc(r, (size_t)B * td);
  if (conv_fwd(r, U->lin2, U->h1g, nullptr, nullptr, U->temb)) return -1;
  U->st = act_alloc(r, (size_t)B * td);
  RUN(launch_act_fwd(U->temb, U->st, (size_t)B * td, 0, r.st));
  U->ss = act_alloc(r, (size_t)B * U->ss_total);
  if (conv_fwd(r, U->lincat, U->st, nullptr, nullptr, U->ss)) return -1;

  U->h0 = act_alloc(r, (size_t)B * P * P * dim);
  if (conv_fwd(r, U->init_conv, x_nhwc, nullptr, nullptr, U->h0)) return -1;
  U->tape_cond = cond_nhwc != nullptr;
  if (cond_nhwc) {
    // x = combine_conv(cat(x, emb_conv(cond)));  emb_conv = 1x1 conv, GELU, 3x3 conv.  The concatenation is never
    // materialised (two-source conv); everything downstream (first block AND the final skip) sees the combined field.
    const size_t nh = (size_t)B * P * P * dim;
    U->cond_in = cond_nhwc;
    U->h0pre = U->h0;
    U->e1 = act_alloc(r, nh);
    if (conv_fwd(r, U->emb1, cond_nhwc, nullptr, nullptr, U->e1)) return -1;
    U->e1g = act_alloc(r, nh);
    RUN(launch_act_fwd(U->e1, U->e1g, nh, 1, r.st));
    U->e2 = act_alloc(r, nh);
    if (conv_fwd(r, U->emb2, U->e1g, nullptr, nullptr, U->e2)) return -1;
    float* hc = act_alloc(r, nh);
    if (conv_fwd(r, U->comb, U->h0pre, U->e2, nullptr, hc)) return -1;
    U->h0 = hc;
  }
  float* x = U->h0;
  int irb = 0, iat = 0;
  for (int i = 0; i < n; ++i) {
    if (resblock_fwd(r, U->rb[irb++], x, nullptr, &x)) return -1;
    if (resblock_fwd(r, U->rb[irb++], x, nullptr, &x, &U->attn[iat])) return -1;
    if (attn_fwd(r, U->attn[iat++], x, &x)) return -1;
    U->skip[i] = x;
    if (i < n - 1) {
      U->down_in[i] = x;
      const int Ho = out_h(U->down[i]);
      float* y = act_alloc(r, (size_t)B * Ho * Ho * U->down[i].Cout);
      if (conv_fwd(r, U->down[i], x, nullptr, nullptr, y)) return -1;
      x = y;
    }
  }
  if (resblock_fwd(r, U->rb[irb++], x, nullptr, &x, &U->attn[iat])) return -1;
  if (attn_fwd(r, U->attn[iat++], x, &x)) return -1;
  if (resblock_fwd(r, U->rb[irb++], x, nullptr, &x)) return -1;
  for (int j = 0; j < n; ++j) {
    const float* sk = U->skip[n - 1 - j];
    if (resblock_fwd(r, U->rb[irb++], x, sk, &x)) return -1;
    if (resblock_fwd(r, U->rb[irb++], x, nullptr, &x, &U->attn[iat])) return -1;
    if (attn_fwd(r, U->attn[iat++], x, &x)) return -1;
    if (j < n - 1) {
      U->up_in[j] = x;
      const int Ho = out_h(U->up[j]);
      float* y = act_alloc(r, (size_t)B * Ho * Ho * U->up[j].Cout);
      if (conv_fwd(r, U->up[j], x, nullptr, nullptr, y)) return -1;
      x = y;
    }
  }
  if (resblock_fwd(r, U->rb[irb++], x, U->h0, &x)) return -1;
  U->xfinal = x;
  if (conv_fwd(r, U->final_conv, x, nullptr, nullptr, out_nchw, 1, U->cfg.sigmoid_last_channel)) return -1;
  return 0;
}

// ------------------------------------------------------------------------------------------------------
// backward building blocks
// ------------------------------------------------------------------------------------------------------
// side stream waits for everything issued so far on the caller's stream (the operands of the next weight-gradient kernel)
static hipStream_t fork_side(Run& r) {
  if (!r.overlap) return r.st;
  pidm_unet* U = r.U;
  if (hipEventRecord(U->ev_fork, r.st) != hipSuccess || hipStreamWaitEvent(U->side, U->ev_fork, 0) != hipSuccess) {
    r.overlap = false;   // degrade to in-order execution
    return r.st;
  }
  r.side_pending = true;
  return U->side;
}
// caller's stream waits for the side stream: required before anything consumes the weight-gradient partials (the deferred
// reduction) - and it orders all later work on the caller's stream after the side stream's reads
static int join_side(Run& r) {
  if (!r.side_pending) return 0;
  pidm_unet* U = r.U;
  if (hipEventRecord(U->ev_join, U->side) != hipSuccess || hipStreamWaitEvent(r.st, U->ev_join, 0) != hipSuccess)
    return fail("backward: stream join failed");
  r.side_pending = false;
  return 0;
}

// Launches the weight-gradient problems queued since the previous flush as ONE grouped launch (per 64 table rows).  The table lives
// behind the reduction table in the deferred arena; as there, only rows that differ from what the device holds are uploaded (steady
// state: nothing - a captured pass must find the table unchanged).
static int flush_wgrads(Run& r) {
  if (r.dry || !r.group_on) return 0;
  pidm_unet* U = r.U;
  for (int fam = 0; fam < kWgFams; ++fam) {
    WgradQueue::Fam& q = r.wq.f[fam];
    const size_t n = q.v.size(), first = q.done;
    if (n <= first) continue;
    if (n > kMaxWgradItems) return fail("backward: weight-gradient table overflow (%zu)", n);
    WgradItem* dev = r.wq_dev + (size_t)fam * kMaxWgradItems;
    std::vector<WgradItem>& host = U->wq_table[fam];
    if (host.capacity() < kMaxWgradItems) host.reserve(kMaxWgradItems);     // never reallocates: async uploads read it
    const bool same = U->wq_table_dev == r.wq_dev && host.size() >= n &&
                      memcmp(host.data() + first, q.v.data() + first, (n - first) * sizeof(WgradItem)) == 0;
    if (!same && r.cap) {
      r.cap->failed = true;        // never captured: the pass is re-run eagerly, which uploads
    } else if (!same) {
      if (U->wq_table_dev != r.wq_dev) {
        for (auto& h : U->wq_table) h.clear();
        U->wq_table_dev = r.wq_dev;
      }
      host.resize(n > host.size() ? n : host.size());
      memcpy(host.data() + first, q.v.data() + first, (n - first) * sizeof(WgradItem));
      if (hipMemcpyAsync(dev + first, host.data() + first, (n - first) * sizeof(WgradItem), hipMemcpyHostToDevice, r.st) != hipSuccess)
        return fail("backward: weight-gradient table upload failed");
      ++g_red_table_uploads;       // (counted with the reduction table's: "steady state uploads nothing" covers both)
    }
    for (size_t a = first; a < n; a += 64) {
      const size_t b = a + 64 < n ? a + 64 : n;
      const unsigned blk0 = q.v[a].blk0, blk1 = (b < n) ? q.v[b].blk0 : q.nblocks;
      if (prof_enabled()) {
        double fl = 0.0;
        for (size_t i = a; i < b; ++i) fl += q.fl[i];
        prof_begin_launch(fam == kWgFam1x1 ? 1 : 3, fl, r.st);      // (the 1x1 streams run on the fp32 MFMA)
      }
      const int rc = launch_wgrad_multi(fam, dev, (int)a, (int)(b - a), blk0, blk1 - blk0, r.st);
      if (prof_enabled()) prof_end_launch(r.st);
      if (rc) return rc;
    }
    q.done = n;
  }
  return 0;
}

// ld_dy: channel stride of dy (0 = L.Cout, contiguous): the two halves of a concatenation's gradient are read in place
static int conv_wgrad(Run& r, const ConvLayer& L, const float* x0, const float* x1, const float* dy, int ld_dy = 0) {
  pidm_unet* U = r.U;
  if (!U->have_grads && !r.dry) return 0;
  if (!ld_dy) ld_dy = L.Cout;
  ConvGeom g;
  const int Ho = out_h(L);
  const hipStream_t wst = r.dry ? r.st : fork_side(r);
  if (L.transposed) {
    if (make_geom(&g, 0, r.B, 2 * L.H, 2 * L.H, L.Cout, 0, ld_dy, 0, L.C0, 4, 4, 2, 1, 0, 4, 4)) return -1;
    float* part = r.part_alloc(wgrad_ws_bytes(g));
    RUN(launch_wgrad(g, dy, nullptr, x0, L.C0, U->G[L.w], nullptr, part, wst, r.q(), r.wgq()));
    if (L.b >= 0) {
      float* cpart = r.part_alloc(colsum_ws_bytes((size_t)r.B * Ho * Ho, L.Cout));
      RUN(launch_colsum(dy, (size_t)r.B * Ho * Ho, L.Cout, ld_dy, U->G[L.b], cpart, wst, r.q()));
    }
  } else {
    if (geom_fwd_layer(L, r.B, 0, &g)) return -1;
    float* part = r.part_alloc(wgrad_ws_bytes(g));
    RUN(launch_wgrad(g, x0, x1, dy, ld_dy, U->G[L.w], L.b >= 0 ? U->G[L.b] : nullptr, part, wst, r.q(), r.wgq()));
    if (r.wgq() && r.wq.pending() >= (size_t)wgrad_group_limit() && flush_wgrads(r)) return -1;
  }
  return 0;
}

// ld_dy / ld_res: channel strides of dy and of the residual (0 = contiguous)
static int conv_dgrad(Run& r, const ConvLayer& L, const float* dy, const float* residual, float* dx, int ld_dy = 0, int ld_res = 0) {
  pidm_conv_desc d = desc_of(L, r.B);
  ConvGeom g;
  int kind;
  if (geom_dgrad(&d, ld_dy ? ld_dy : L.Cout, L.C0 + L.C1, &g, &kind)) return -1;
  if (ld_res) g.ldr = ld_res;
  RUN(launch_conv(g, dy, nullptr, r.wpack + L.off_d, nullptr, residual, dx, 0, r.st));
  return 0;
}

// g_out [B,HW,Co] -> g_x [B,HW,C0+C1]
// pc2_in > 0: the kernel that produced g_out already left the per-channel sums GroupNorm 2's backward starts with in r.scratch
// (pc2_in 32-pixel chunks per image) - its own reduction pass is skipped.  next / pc_next: the block whose GroupNorm 2 will consume
// THIS block's g_x as its g_out (the previous block of the level in forward order): the input-gradient convolution that finally
// writes g_x (conv 1's dgrad, which adds the skip / res_conv share as its residual) leaves those sums from its epilogue, and
// *pc_next says how many chunks it wrote (0: not produced - ineligible shape or a kernel without the epilogue).
static int resblock_bwd(Run& r, ResBlock& m, const float* g_out, float* g_x, float* dss, int pc2_in = 0, const ResBlock* next = nullptr,
                        int* pc_next = nullptr) {
  pidm_unet* U = r.U;
  const int B = r.B, HW = m.H * m.H, Co = m.Co, G = U->groups;
  const size_t n = (size_t)B * HW * Co;
  const size_t mk = r.tmp.mark();
  if (pc_next) *pc_next = 0;
  float* g_c = r.tmp.alloc(n);
  float* dgb2 = r.defer_on ? r.defer.alloc((size_t)B * 2 * Co) : nullptr;   // allocated outside RUN: dry runs size the arena
  float* dgb1 = r.defer_on ? r.defer.alloc((size_t)B * 2 * Co) : nullptr;
  RUN(launch_gn_bwd(m.c, g_out, m.st2, U->P[m.gn2w], U->P[m.gn2b], nullptr, nullptr, 0, nullptr, g_c, U->G[m.gn2w],
                    U->G[m.gn2b], B, HW, Co, G, r.scratch, r.st, dgb2, r.q(), pc2_in));
  if (conv_wgrad(r, m.c2, m.bact, nullptr, g_c)) return -1;
  float* g_b = r.tmp.alloc(n);
  const float* ss = m.has_mlp ? U->ss + m.ss_off : nullptr;
  const float* ssb = m.has_mlp ? U->P[m.mlpb] : nullptr;
  // the dgrad of conv2 produces dy of GroupNorm 1: its epilogue also leaves the per-channel sums the GroupNorm backward starts
  // with (the first of its two passes over x and dy).  Needs the deferred-reduction arena: otherwise the weight-gradient
  // partials of conv2 live in r.scratch, where the sums go.
  int pcb = 0;
  {
    const bool off = knob("PIDM_NO_GN_EPILOGUE") != nullptr || knob("PIDM_NO_BN_EPILOGUE") != nullptr;
    pidm_conv_desc d = desc_of(m.c2, r.B);
    ConvGeom g;
    int kind;
    if (geom_dgrad(&d, m.c2.Cout, m.c2.C0 + m.c2.C1, &g, &kind)) return -1;
    const int cpg = (G > 0 && Co % G == 0) ? Co / G : 0;
    int chunks = HW / 32;
    const bool ok = !off && r.defer_on && g.Cout == Co && Co % 32 == 0 && cpg >= 1 && HW % 32 == 0 && g.nz == 1 && g.nph == 1 && g.os == 1 &&
                    g.soc == 1 && g.KH == 3 && g.Ho * g.Wo == HW;
    if (ok) {
      g.bn_part = reinterpret_cast<double*>(r.scratch);
      g.bn_x = m.a; g.bn_stats = m.st1; g.bn_gamma = U->P[m.gn1w]; g.bn_beta = U->P[m.gn1b];
      g.bn_ss = ss; g.bn_ssb = ssb; g.bn_ldss = U->ss_total;
      g.bn_cpg = cpg; g.bn_G = G; g.bn_nchunk = HW / 32;
      g.part_chunks_out = &chunks;
    }
    if (!r.dry) {
      const int rc = launch_conv(g, g_c, nullptr, r.wpack + m.c2.off_d, nullptr, nullptr, g_b, 0, r.st);
      if (rc < 0) return rc;
      if (ok && rc == 0) pcb = chunks;
    }
  }
  // g_c is dead on THIS stream after the two uses above, but the side-stream wgrad of c2 may still be reading it
  float* g_a = r.keep_frames() ? r.tmp.alloc(n) : g_c;
  RUN(launch_gn_bwd(m.a, g_b, m.st1, U->P[m.gn1w], U->P[m.gn1b], ss, ssb, U->ss_total, m.has_mlp ? dss + m.ss_off : nullptr,
                    g_a, U->G[m.gn1w], U->G[m.gn1b], B, HW, Co, G, r.scratch, r.st, dgb1, r.q(), pcb));
  if (conv_wgrad(r, m.c1, m.x0, m.x1, g_a)) return -1;
  const float* res1 = g_out;
  if (m.has_res) {
    if (conv_wgrad(r, m.cr, m.x0, m.x1, g_out)) return -1;
    if (conv_dgrad(r, m.cr, g_out, nullptr, g_x)) return -1;
    res1 = g_x;
  }
  {
    // conv 1's input gradient (+ the skip share as residual) = g_x; with `next` its epilogue also sums for next's GroupNorm 2
    const bool off = knob("PIDM_NO_GN_EPILOGUE") != nullptr || knob("PIDM_NO_BN2_EPILOGUE") != nullptr;
    pidm_conv_desc d = desc_of(m.c1, r.B);
    ConvGeom g;
    int kind;
    if (geom_dgrad(&d, m.c1.Cout, m.c1.C0 + m.c1.C1, &g, &kind)) return -1;
    bool ok = false;
    int chunks = HW / 32;
    if (next && !off && r.defer_on) {
      const int Cn = next->Co, cpg = (G > 0 && Cn % G == 0) ? Cn / G : 0;
      // (res1 != g_x: with a res_conv the residual is the output buffer itself, and the row-streaming kernel re-reads the residual
      // for the GroupNorm-backward sums AFTER it stored the output row - in place it would sum result + residual + result)
      ok = next->H == m.H && g.Cout == Cn && Cn % 32 == 0 && cpg >= 1 && HW % 32 == 0 && g.nz == 1 && g.nph == 1 && g.os == 1 && g.soc == 1 &&
           g.KH == 3 && g.Ho * g.Wo == HW && res1 != g_x;
      if (ok) {
        g.bn_part = reinterpret_cast<double*>(r.scratch);
        g.bn_x = next->c; g.bn_stats = next->st2; g.bn_gamma = U->P[next->gn2w]; g.bn_beta = U->P[next->gn2b];
        g.bn_ss = nullptr; g.bn_ssb = nullptr; g.bn_ldss = 0;
        g.bn_cpg = cpg; g.bn_G = G; g.bn_nchunk = HW / 32;
        g.bn_res = 1;
        g.part_chunks_out = &chunks;
      }
    }
    if (!r.dry) {
      const int rc = launch_conv(g, g_a, nullptr, r.wpack + m.c1.off_d, nullptr, res1, g_x, 0, r.st);
      if (rc < 0) return rc;
      if (ok && rc == 0 && pc_next) *pc_next = chunks;
    }
  }
  // side-stream weight gradients may still be reading buffers of this arena frame: with the overlap on, the frame is simply
  // kept until the end of backward (every gradient buffer is then unique; one join before the deferred reduction)
  if (!r.keep_frames()) r.tmp.release(mk);
  return 0;
}

// g_out [B,N,C] -> g_x [B,N,C]
// prev / pc_prev: the ResnetBlock whose output this attention block consumed (the next one the backward walks).  The LayerNorm
// backward that ends this function then also leaves the first pass of that block's GroupNorm-2 backward - the per-(image, chunk,
// channel) sums - in r.scratch and reports the chunk count (resblock_bwd's pc2_in; 0: not produced).  OFF by default
// (PIDM_LN_GN_SUMS=1 turns it on): 9 launches and 9 reads of a gradient tensor fewer per step, but the LayerNorm kernel then walks one
// chunk of one image per block instead of striding over the batch and loses more than the separate pass cost - batch 64 +1.0 % per
// step, batch 256 +-0, mechanics +1.4 % (profiles/r06_ln_gn_sums_ab.txt).
static int attn_bwd(Run& r, AttnBlock& a, const float* g_out, float* g_x, const ResBlock* prev = nullptr, int* pc_prev = nullptr) {
  pidm_unet* U = r.U;
  const int B = r.B, N = a.H * a.H, C = a.C, heads = U->heads, HD = heads * 32;
  const size_t npix = (size_t)B * N;
  const size_t mk = r.tmp.mark();
  if (pc_prev) *pc_prev = 0;
  // (the sums live in r.scratch like the dgrad epilogue's; the deferred-reduction arena keeps the weight-gradient partials out of it)
  const int gn_pc = (prev && pc_prev && r.defer_on && !prev->has_res && prev->Co == C && prev->H == a.H && knob_on("PIDM_LN_GN_SUMS"))
                        ? layernorm_bwd_gn_chunks(B, N) : 0;
  const float* gn_x = gn_pc ? prev->c : nullptr;
  const float* gn_st = gn_pc ? prev->st2 : nullptr;
  const float* gn_gm = gn_pc ? U->P[prev->gn2w] : nullptr;
  const float* gn_bt = gn_pc ? U->P[prev->gn2b] : nullptr;
  void* gn_part = gn_pc ? (void*)r.scratch : nullptr;
  float* g_qkv = nullptr;
  float* g_xn = nullptr;
  bool out_bias_with_ln = false;
  if (r.dry ? attn_projected(a, heads) : a.projected) {
    // no qkv tensor, no dqkv tensor: d_xn and the to_qkv / to_out weight-gradient shares come straight from (xn, dY)
    const size_t nr = (size_t)B * lap_dw_ranges(N, C);
    const size_t n_qk = nr * 2 * HD * C, n_v = (size_t)B * HD * C, n_o = (size_t)B * C * HD;
    float* dwqk = r.defer_on ? r.defer.alloc(n_qk) : r.tmp.alloc(n_qk);
    float* dwv = r.defer_on ? r.defer.alloc(n_v) : r.tmp.alloc(n_v);
    float* dwo = r.defer_on ? r.defer.alloc(n_o) : r.tmp.alloc(n_o);
    float* ltmp = r.tmp.alloc(lap_bwd_tmp_floats(B, heads, C));
    g_xn = r.tmp.alloc(npix * C);
    RUN(launch_lap_backward(a.xn, g_out, U->P[a.qkv.w], U->P[a.out.w], a.lsaved, a.qstat, g_xn, dwqk, dwv, dwo, ltmp, C, B, N, heads, r.scratch,
                            r.st));
    if (U->have_grads && !r.dry) {
      float* gw = U->G[a.qkv.w];
      if (r.q()) {
        r.q()->push(dwqk, gw, nullptr, nullptr, (size_t)2 * HD * C, (int)nr, 2 * HD, C, 1, 2 * HD, C);
        r.q()->push(dwv, gw + (size_t)2 * HD * C, nullptr, nullptr, (size_t)HD * C, B, HD, C, 1, HD, C);
        r.q()->push(dwo, U->G[a.out.w], nullptr, nullptr, (size_t)C * HD, B, C, HD, 1, C, HD);
      } else {
        RUN(launch_split_reduce(dwqk, gw, nullptr, nullptr, (int)nr, 2 * HD, C, 1, 2 * HD, C, r.st));
        RUN(launch_split_reduce(dwv, gw + (size_t)2 * HD * C, nullptr, nullptr, B, HD, C, 1, HD, C, r.st));
        RUN(launch_split_reduce(dwo, U->G[a.out.w], nullptr, nullptr, B, C, HD, 1, C, HD, r.st));
      }
    }
    // the to_out bias gradient (column sums of g_out) rides with the LayerNorm backward, which reads g_out as its residual share
    float* ln_part = r.part_alloc(layernorm_bwd_ws_bytes(C) + colsum_ws_bytes(1024, C));
    RUN(launch_layernorm_bwd(a.x, U->P[a.gamma], g_xn, g_out, g_x, U->G[a.gamma], npix, C, ln_part, r.st, r.q(),
                             (U->have_grads && !r.dry) ? U->G[a.out.b] : nullptr, gn_x, gn_st, gn_gm, gn_bt, U->groups, B, N, gn_part));
    if (pc_prev) *pc_prev = gn_pc;
    if (!r.keep_frames()) r.tmp.release(mk);
    return 0;
  }
  if (!a.mid && (la_fused_ok(N, heads, C, C) && la_fused_pays(B, N))) {
    // attention backward fused with the to_out projection: the gradient of the attention output (npix*HD floats), the
    // projection's dgrad and its wgrad over the materialised attention output are all replaced (k_attn.hip)
    g_qkv = r.tmp.alloc(npix * 3 * HD);
    float* dctx = r.tmp.alloc((size_t)B * heads * 1024);
    float* rowdot = r.tmp.alloc((size_t)B * heads * 32);
    const size_t dwn = (size_t)B * C * HD;
    float* dwpart = r.defer_on ? r.defer.alloc(dwn) : r.tmp.alloc(dwn);
    RUN(launch_la_backward_fused(a.qkvb, a.kstat, a.qstat, a.ctx, g_out, C, U->P[a.out.w], C, dctx, rowdot, g_qkv, dwpart, B, N,
                                 heads, r.scratch, r.st));
    if (U->have_grads && !r.dry) {
      if (r.q()) r.q()->push(dwpart, U->G[a.out.w], nullptr, nullptr, (size_t)C * HD, B, C, HD, 1, C, HD);
      else RUN(launch_split_reduce(dwpart, U->G[a.out.w], nullptr, nullptr, B, C, HD, 1, C, HD, r.st));
      if (a.out.b >= 0) out_bias_with_ln = true;      // column sums of g_out: with the LayerNorm backward below
    }
    g_xn = r.tmp.alloc(npix * C);
  } else {
    if (conv_wgrad(r, a.out, a.attn, nullptr, g_out)) return -1;
    float* g_attn = r.tmp.alloc(npix * HD);
    if (conv_dgrad(r, a.out, g_out, nullptr, g_attn)) return -1;
    g_qkv = r.tmp.alloc(npix * 3 * HD);
    if (a.mid) {
      RUN(launch_mid_attn(a.qkvb, g_attn, g_qkv, B, N, heads, true, r.st));
    } else {
      float* dctx = r.tmp.alloc((size_t)B * heads * 1024);
      float* rowdot = r.tmp.alloc((size_t)B * heads * 32);
      RUN(launch_la_backward(a.qkvb, a.kstat, a.qstat, a.ctx, g_attn, dctx, rowdot, g_qkv, B, N, heads, r.scratch, r.st));
    }
    g_xn = g_attn;  // reuse (npix*HD >= npix*C is not guaranteed) -> allocate when C > HD
    if (C > HD) g_xn = r.tmp.alloc(npix * C);
  }
  if (conv_wgrad(r, a.qkv, a.xn, nullptr, g_qkv)) return -1;
  if (conv_dgrad(r, a.qkv, g_qkv, nullptr, g_xn)) return -1;
  float* ln_part = r.part_alloc(layernorm_bwd_ws_bytes(C) + colsum_ws_bytes(1024, C));
  RUN(launch_layernorm_bwd(a.x, U->P[a.gamma], g_xn, g_out, g_x, U->G[a.gamma], npix, C, ln_part, r.st, r.q(),
                           out_bias_with_ln ? U->G[a.out.b] : nullptr, gn_x, gn_st, gn_gm, gn_bt, U->groups, B, N, gn_part));
  if (pc_prev) *pc_prev = gn_pc;
  // side-stream weight gradients may still be reading buffers of this arena frame: with the overlap on, the frame is simply
  // kept until the end of backward (every gradient buffer is then unique; one join before the deferred reduction)
  if (!r.keep_frames()) r.tmp.release(mk);
  return 0;
}

static size_t scratch_floats_needed(pidm_unet* U, int B) {
  size_t mx = 1 << 16;
  auto upd = [&](size_t bytes) { if (bytes / 4 + 64 > mx) mx = bytes / 4 + 64; };
  auto conv_ws = [&](const ConvLayer& L) {
    ConvGeom g;
    if (L.transposed) {
      if (make_geom(&g, 0, B, 2 * L.H, 2 * L.H, L.Cout, 0, L.Cout, 0, L.C0, 4, 4, 2, 1, 0, 4, 4)) return;
    } else if (geom_fwd_layer(L, B, 0, &g)) {
      return;
    }
    upd(wgrad_ws_bytes(g));
    const int Ho = out_h(L);
    upd(colsum_ws_bytes((size_t)B * Ho * Ho, L.Cout));
  };
  conv_ws(U->init_conv); conv_ws(U->lin1); conv_ws(U->lin2); conv_ws(U->lincat); conv_ws(U->final_conv);
  for (auto& m : U->rb) {
    conv_ws(m.c1); conv_ws(m.c2);
    if (m.has_res) conv_ws(m.cr);
    upd(gn_ws_bytes(B, m.H * m.H, m.Co, U->groups));
  }
  for (auto& a : U->attn) {
    conv_ws(a.qkv); conv_ws(a.out);
    upd(layernorm_bwd_ws_bytes(a.C) + colsum_ws_bytes(1024, a.C));
    upd(la_scratch_floats(B, a.H * a.H, U->heads) * sizeof(float));
    // both attention forms, whatever the knobs say right now (the backward replays the forward's latched decision)
    if (attn_shape_projectable(a, U->heads)) upd(lap_scratch_floats(B, a.H * a.H, U->heads, a.C) * sizeof(float));
    if (!a.mid && la_fused_ok(a.H * a.H, U->heads, a.C, a.C) && la_fused_pays(B, a.H * a.H)) upd(la_fused_scratch_floats(B, a.H * a.H, U->heads, a.C) * sizeof(float));
  }
  for (int i = 0; i < U->n_lv - 1; ++i) { conv_ws(U->down[i]); conv_ws(U->up[i]); }
  return mx;
}

// ---- stream capture of a pass into graph segments ---------------------------------------------------------------------------
static int cap_begin(Run& r) {
  GraphCapture* c = r.cap;
  if (hipStreamBeginCapture(r.st, hipStreamCaptureModeThreadLocal) != hipSuccess) {
    (void)hipGetLastError();
    c->failed = true;
    return fail("graph capture: hipStreamBeginCapture failed");
  }
  c->open = true;
  c->kernels0 = g_kernel_enqueues;
  return 0;
}
// closes the open segment, instantiates and LAUNCHES it on the caller's stream (a capture records, it does not execute), records
// `ev_after` there, and opens the next segment when `reopen`.  After a failure nothing more is launched: the caller discards the
// entry and runs the whole pass eagerly (every pass is idempotent: it only writes its own outputs and temporaries).
static int cap_end_segment(Run& r, hipEvent_t ev_after, bool reopen) {
  GraphCapture* c = r.cap;
  hipGraph_t g = nullptr;
  const hipError_t e = hipStreamEndCapture(r.st, &g);
  c->open = false;
  if (e != hipSuccess || !g) {
    (void)hipGetLastError();
    c->failed = true;
    if (g) (void)hipGraphDestroy(g);
    return fail("graph capture: hipStreamEndCapture failed (%s)", hipGetErrorString(e));
  }
  GraphSeg seg;
  seg.graph = g;
  seg.ev_after = ev_after;
  seg.kernels = g_kernel_enqueues - c->kernels0;
  g_kernel_enqueues = c->kernels0;     // recorded, not enqueued: counted per replay instead
  if (!c->failed && hipGraphInstantiate(&seg.exec, g, nullptr, nullptr, 0) != hipSuccess) {
    (void)hipGetLastError();
    seg.exec = nullptr;
    c->failed = true;
  }
  c->entry->segs.push_back(seg);
  if (!c->failed) {
    if (hipGraphLaunch(seg.exec, c->user_st) != hipSuccess) {
      (void)hipGetLastError();
      c->failed = true;
    } else {
      ++g_graph_launches;
      g_graph_kernels += seg.kernels;
      if (ev_after && hipEventRecord(ev_after, c->user_st) != hipSuccess) return fail("backward: phase event record failed");
    }
  }
  if (reopen) return cap_begin(r);
  return 0;
}

// Runs the fixed-order reductions queued since the previous flush (weight / bias / norm-parameter gradients) in one launch.
// `phase` >= 0: also record that phase's event (data-parallel overlap).  The descriptor table lives at the head of the deferred
// arena; only the part that differs from what the device already holds is uploaded (steady state: nothing).
static int flush_reductions(Run& r, ReduceDesc* red_dev, size_t* done, int phase, bool final_flush = false) {
  pidm_unet* U = r.U;
  if (r.dry) return 0;
  if (flush_wgrads(r)) return -1;          // the queued weight-gradient problems write the partial slabs this reduction reads
  if (join_side(r)) return -1;
  const size_t n = r.rq.v.size(), first = *done;
  if (n > kMaxReduceDesc) return fail("backward: reduction table overflow (%zu)", n);
  if (n > first) {
    if (U->red_table.capacity() < kMaxReduceDesc) U->red_table.reserve(kMaxReduceDesc);   // never reallocates afterwards: async uploads read it
    const bool same = U->red_table_dev == red_dev && U->red_table.size() >= n &&
                      memcmp(U->red_table.data() + first, r.rq.v.data() + first, (n - first) * sizeof(ReduceDesc)) == 0;
    if (!same && r.cap) {
      r.cap->failed = true;      // a table upload is never captured: host and device table stay as they are, the pass is re-run eagerly
    } else if (!same) {
      if (knob("PIDM_REDUCE_STATS")) {   // one line per table change: what the deferred reduction reads
        double bytes = 0, outs = 0;
        for (size_t i = first; i < n; ++i) {
          const ReduceDesc& d = r.rq.v[i];
          bytes += (double)d.nsplit * d.sstride * 4;
          outs += (double)d.M * d.N * d.T;
          if ((double)d.nsplit * d.sstride * 4 > 8e6)
            fprintf(stderr, "[pidm]   nsplit=%d M=%d N=%d T=%d: %.1f MB\n", d.nsplit, d.M, d.N, d.T, (double)d.nsplit * d.sstride * 4 / 1e6);
        }
        fprintf(stderr, "[pidm] deferred reduction (phase %d): %zu tensors, %.1f MB of partials -> %.2f M outputs\n", phase, n - first,
                bytes / 1e6, outs / 1e6);
      }
      if (U->red_table_dev != red_dev) U->red_table.clear();
      U->red_table.resize(n > U->red_table.size() ? n : U->red_table.size());
      memcpy(U->red_table.data() + first, r.rq.v.data() + first, (n - first) * sizeof(ReduceDesc));
      if (hipMemcpyAsync(red_dev + first, U->red_table.data() + first, (n - first) * sizeof(ReduceDesc), hipMemcpyHostToDevice, r.st) != hipSuccess)
        return fail("backward: reduction table upload failed");
      ++g_red_table_uploads;
      U->red_table_dev = red_dev;
      U->red_table_owner = U->arena_sig;     // the layout whose defer arena holds the table
    }
    const unsigned blk0 = r.rq.v[first].blk0;
    RUN(launch_reduce_multi(red_dev + first, (int)(n - first), r.rq.nblocks - blk0, r.st, blk0));
    *done = n;
  }
  hipEvent_t ev = (phase >= 0 && phase < 3) ? U->phase_ev[phase] : nullptr;
  if (r.cap) {
    // an external event cannot be recorded from inside a capture: the segment ends here, is launched, and the event is recorded
    // on the caller's stream behind it
    if (ev || final_flush) return cap_end_segment(r, ev, !final_flush);
    return 0;
  }
  if (ev && hipEventRecord(ev, r.st) != hipSuccess) return fail("backward: phase event record failed");
  return 0;
}

static int backward_impl(Run& r, const float* grad_out_nchw, float* grad_x_nhwc) {
  pidm_unet* U = r.U;
  const int B = r.B, P = U->cfg.image_size, dim = U->cfg.dim, n = U->n_lv, td = U->tdim, od = U->cfg.out_dim;
  const size_t HW = (size_t)P * P;
  r.defer_on = true;
  r.rq.v.clear();
  r.rq.nblocks = 0;
  ReduceDesc* red_dev = reinterpret_cast<ReduceDesc*>(r.defer.alloc(kMaxReduceDesc * sizeof(ReduceDesc) / 4));
  size_t red_done = 0;
  r.wq.clear();
  r.wq_dev = reinterpret_cast<WgradItem*>(r.defer.alloc(kWgFams * kMaxWgradItems * sizeof(WgradItem) / 4));
  const int n_phases = U->n_phases;
  float* dss = r.tmp.alloc((size_t)B * U->ss_total);
  float* g_o = r.tmp.alloc((size_t)B * HW * od);
  RUN(launch_nchw_to_nhwc(grad_out_nchw, g_o, B, od, (int)HW, U->cfg.sigmoid_last_channel ? U->out_nchw : nullptr, r.st));
  // final 1x1 conv (the NHWC gradient has channel stride od, which may not be a multiple of 4: scalar staging)
  {
    const ConvLayer& L = U->final_conv;
    if (U->have_grads || r.dry) {
      ConvGeom g;
      if (geom_fwd_layer(L, B, 0, &g)) return -1;
      float* part = r.part_alloc(wgrad_ws_bytes(g));
      RUN(launch_wgrad(g, U->xfinal, nullptr, g_o, od, U->G[L.w], U->G[L.b], part, r.st, r.q()));
    }
  }
  float* g_x = r.tmp.alloc((size_t)B * HW * dim);
  if (conv_dgrad(r, U->final_conv, g_o, nullptr, g_x)) return -1;
  int irb = (int)U->rb.size() - 1, iat = (int)U->attn.size() - 1;
  // final resblock: input cat(x, h0)
  float* g_cat = r.tmp.alloc((size_t)B * HW * 2 * dim);
  if (resblock_bwd(r, U->rb[irb--], g_x, g_cat, dss)) return -1;
  const float* g_r = g_cat + dim;      // gradient wrt h0 through the final concat: read in place (stride 2 dim) where it is added to g_h0
  g_x = r.tmp.alloc((size_t)B * HW * dim);
  RUN(launch_copy_add(g_x, dim, g_cat, 2 * dim, nullptr, 0, (size_t)B * HW, dim, r.st));
  // The gradient of a concatenation cat(x, skip) is one [pixels][2 dout] tensor; its two halves are consumed IN PLACE through a
  // channel stride of 2 dout wherever the consumer is a convolution (dy of the next up-sampling layer, residual of the
  // down-sampling layer's input gradient) - only the halves that feed normalisation / attention kernels are copied out
  std::vector<const float*> g_skip(n, nullptr);
  std::vector<int> g_skip_ld(n, 0);
  int ld_gx = 0;                   // channel stride of g_x (0 = contiguous)
  for (int j = n - 1; j >= 0; --j) {
    const int lvl = n - 1 - j;
    const int din = U->dims[lvl], dout = U->dims[lvl + 1], H = P >> lvl;
    const size_t npix = (size_t)B * H * H;
    if (j < n - 1) {
      const ConvLayer& L = U->up[j];
      if (conv_wgrad(r, L, U->up_in[j], nullptr, g_x, ld_gx)) return -1;
      float* g = r.tmp.alloc(npix * din);
      if (conv_dgrad(r, L, g_x, nullptr, g, ld_gx)) return -1;
      g_x = g;
      ld_gx = 0;
    }
    float* g1 = r.tmp.alloc(npix * din);
    int pc_a = 0;              // the attention block's LayerNorm backward sums for block 2's GroupNorm 2
    if (attn_bwd(r, U->attn[iat--], g_x, g1, &U->rb[irb], &pc_a)) return -1;
    float* g2 = r.tmp.alloc(npix * din);
    int pc_b1 = 0;             // block 2's final input-gradient convolution sums for block 1's GroupNorm 2
    if (resblock_bwd(r, U->rb[irb], g1, g2, dss, pc_a, &U->rb[irb - 1], &pc_b1)) return -1;
    --irb;
    float* gc = r.tmp.alloc(npix * 2 * dout);
    if (resblock_bwd(r, U->rb[irb--], g2, gc, dss, pc_b1)) return -1;
    g_skip[lvl] = gc + dout;
    g_skip_ld[lvl] = 2 * dout;
    if (j > 0) {
      g_x = gc;                    // read by the convolutions of up[j-1] with stride 2 dout
      ld_gx = 2 * dout;
    } else {
      g_x = r.tmp.alloc(npix * dout);   // feeds the bottleneck's GroupNorm backward: contiguous
      RUN(launch_copy_add(g_x, dout, gc, 2 * dout, nullptr, 0, npix, dout, r.st));
      ld_gx = 0;
    }
  }
  // decoder half done: every gradient of ups.* / final_conv.* is final after this flush
  if (n_phases >= 2 && flush_reductions(r, red_dev, &red_done, 0)) return -1;
  {
    const int C = U->dims[n], H = P >> (n - 1);
    const size_t nn = (size_t)B * H * H * C;
    float* g1 = r.tmp.alloc(nn);
    if (resblock_bwd(r, U->rb[irb--], g_x, g1, dss)) return -1;
    float* g2 = r.tmp.alloc(nn);
    int pc_a = 0;
    if (attn_bwd(r, U->attn[iat--], g1, g2, &U->rb[irb], &pc_a)) return -1;
    float* g3 = r.tmp.alloc(nn);
    if (resblock_bwd(r, U->rb[irb--], g2, g3, dss, pc_a)) return -1;
    g_x = g3;
  }
  for (int i = n - 1; i >= 0; --i) {
    const int din = U->dims[i], dout = U->dims[i + 1], H = P >> i;
    const size_t npix = (size_t)B * H * H;
    float* g_la = r.tmp.alloc(npix * dout);
    if (i < n - 1) {
      const ConvLayer& L = U->down[i];
      if (conv_wgrad(r, L, U->down_in[i], nullptr, g_x)) return -1;
      if (conv_dgrad(r, L, g_x, g_skip[i], g_la, 0, g_skip_ld[i])) return -1;
    } else {
      RUN(launch_copy_add(g_la, dout, g_x, dout, g_skip[i], g_skip_ld[i], npix, dout, r.st));
    }
    float* g1 = r.tmp.alloc(npix * dout);
    int pc_a = 0;
    if (attn_bwd(r, U->attn[iat--], g_la, g1, &U->rb[irb], &pc_a)) return -1;
    float* g2 = r.tmp.alloc(npix * dout);
    int pc_b1 = 0;
    if (resblock_bwd(r, U->rb[irb], g1, g2, dss, pc_a, &U->rb[irb - 1], &pc_b1)) return -1;
    --irb;
    float* g3 = r.tmp.alloc(npix * din);
    if (resblock_bwd(r, U->rb[irb--], g2, g3, dss, pc_b1)) return -1;
    g_x = g3;
  }
  // encoder half done: downs.* / mid_* gradients are final after this flush
  if (n_phases >= 3 && flush_reductions(r, red_dev, &red_done, 1)) return -1;
  // h0 feeds both the first resblock and the final concat
  float* g_h0 = r.tmp.alloc((size_t)B * HW * dim);
  RUN(launch_copy_add(g_h0, dim, g_x, dim, g_r, 2 * dim, (size_t)B * HW, dim, r.st));
  if (U->tape_cond) {
    const size_t nh = (size_t)B * HW * dim;
    if (conv_wgrad(r, U->comb, U->h0pre, U->e2, g_h0)) return -1;
    float* g_cc = r.tmp.alloc(2 * nh);
    if (conv_dgrad(r, U->comb, g_h0, nullptr, g_cc)) return -1;
    float* g_h0p = r.tmp.alloc(nh);
    float* g_e2 = r.tmp.alloc(nh);
    RUN(launch_copy_add(g_h0p, dim, g_cc, 2 * dim, nullptr, 0, (size_t)B * HW, dim, r.st));
    RUN(launch_copy_add(g_e2, dim, g_cc + dim, 2 * dim, nullptr, 0, (size_t)B * HW, dim, r.st));
    if (conv_wgrad(r, U->emb2, U->e1g, nullptr, g_e2)) return -1;
    float* g_e1g = r.tmp.alloc(nh);
    if (conv_dgrad(r, U->emb2, g_e2, nullptr, g_e1g)) return -1;
    float* g_e1 = r.tmp.alloc(nh);
    RUN(launch_act_bwd(U->e1, g_e1g, g_e1, nh, 1, r.st));
    if (conv_wgrad(r, U->emb1, U->cond_in, nullptr, g_e1)) return -1;
    g_h0 = g_h0p;
    if (!r.dry) U->cond_grads_dirty = true;
  }
  if (conv_wgrad(r, U->init_conv, U->x_in, nullptr, g_h0)) return -1;
  if (grad_x_nhwc) {
    if (conv_dgrad(r, U->init_conv, g_h0, nullptr, grad_x_nhwc)) return -1;
  }
  // ---- time path ----
  if (U->have_grads || r.dry) {
    const int nf = 4 * n + 2;
    ConvGeom g;
    if (geom_fwd_layer(U->lincat, B, 0, &g)) return -1;
    // all FiLM weights and biases at once (contiguous in the flat gradient buffer)
    float* part = r.part_alloc(wgrad_ws_bytes(g));
    RUN(launch_wgrad(g, U->st, nullptr, dss, U->ss_total, U->G[0], U->G[nf], part, r.st, r.q()));
    float* d_st = r.tmp.alloc((size_t)B * td);
    {
      // d_st[b][n] = sum_k dss[b][k] W[n][k] with k over all FiLM outputs (3968 / 15872) and only B rows: split-K kernel
      pidm_conv_desc dd = desc_of(U->lincat, B);
      ConvGeom gd;
      int kind;
      if (geom_dgrad(&dd, U->lincat.Cout, U->lincat.C0, &gd, &kind)) return -1;
      const int Kp = packed_kp(gd);
      if (smallm_splitk_ok(B, td, U->ss_total, U->ss_total, Kp)) {
        float* sc = r.tmp.alloc(smallm_splitk_ws_floats(B, td, U->ss_total));
        RUN(launch_smallm_splitk(dss, U->ss_total, r.wpack + U->lincat.off_d, Kp, d_st, B, td, U->ss_total, sc, r.st));
      } else if (conv_dgrad(r, U->lincat, dss, nullptr, d_st)) {
        return -1;
      }
    }
    float* d_temb = r.tmp.alloc((size_t)B * td);
    RUN(launch_act_bwd(U->temb, d_st, d_temb, (size_t)B * td, 0, r.st));
    if (conv_wgrad(r, U->lin2, U->h1g, nullptr, d_temb)) return -1;
    float* d_h1g = r.tmp.alloc((size_t)B * td);
    if (conv_dgrad(r, U->lin2, d_temb, nullptr, d_h1g)) return -1;
    float* d_h1 = r.tmp.alloc((size_t)B * td);
    RUN(launch_act_bwd(U->h1, d_h1g, d_h1, (size_t)B * td, 1, r.st));
    if (conv_wgrad(r, U->lin1, U->emb, nullptr, d_h1)) return -1;
  }
  // ---- every remaining queued fixed-order reduction (weight/bias/norm-parameter gradients) in one launch ----
  if (flush_reductions(r, red_dev, &red_done, n_phases - 1, /*final_flush=*/true)) return -1;
  return 0;
}

static int widest_level(const pidm_unet* U) {
  int w = 0;
  for (int d : U->dims) w = d > w ? d : w;
  return w;
}

static int plan_sizes(pidm_unet* U, int B, int training, size_t* tape_bytes, size_t* tmp_bytes) {
  const long sig = lap_knob_signature();
  if (sig != U->knob_sig) {    // PIDM_NO_LAP / PIDM_LAP_MIN_N changed on a live handle: the cached plans describe the other form
    U->ws_cache[0].clear();
    U->ws_cache[1].clear();
    U->defer_cache.clear();
    U->knob_sig = sig;
    U->wq_table_dev = nullptr;   // (the layout the grouped tables were uploaded under is gone with the plans)
    for (auto& t : U->wq_table) t.clear();
  }
  auto& cache = U->ws_cache[training ? 1 : 0];
  auto it = cache.find(B);
  if (it != cache.end()) {
    *tape_bytes = it->second.first;
    *tmp_bytes = it->second.second;
    return 0;
  }
  auto keep_defer = U->defer_cache;
  Run r;
  r.U = U; r.B = B; r.train = training != 0; r.dry = true; r.st = nullptr; r.wpack = nullptr;
  r.tape.dry = r.tmp.dry = r.defer.dry = true;
  r.side_allowed = backward_eager(widest_level(U));
  r.group_on = wgrad_group_on(B, U->cfg.image_size, r.side_allowed);      // as backward_body decides it
  // state touched by a dry run is restored afterwards
  pidm_unet saved_ptrs = *U;
  r.scratch_floats = scratch_floats_needed(U, B);
  r.scratch = r.tmp.alloc(r.scratch_floats);
  int rc = forward_impl(r, nullptr, nullptr, nullptr, U->cond_enabled ? reinterpret_cast<const float*>(16) : nullptr);
  if (!rc && training) {
    r.tmp.release(align_up(r.scratch_floats * sizeof(float), 256));
    rc = backward_impl(r, nullptr, reinterpret_cast<float*>(16));
  }
  const size_t tb = r.tape.hwm, mb = r.tmp.hwm, db = align_up(r.defer.hwm + 4096, 4096);
  auto keep_cache0 = U->ws_cache[0];
  auto keep_cache1 = U->ws_cache[1];
  *U = saved_ptrs;
  U->ws_cache[0] = keep_cache0;
  U->ws_cache[1] = keep_cache1;
  U->defer_cache = keep_defer;
  if (rc) return rc;
  // the deferred-reduction arena is the tail of the temporaries region
  const size_t mb_al = align_up(mb + 4096, 4096);
  if (training) U->defer_cache[B] = db;
  cache[B] = {tb + 4096, mb_al + (training ? db : 0)};
  *tape_bytes = tb + 4096;
  *tmp_bytes = mb_al + (training ? db : 0);
  return 0;
}

// Static input / output buffers of a pass, between the packed weights and the tape: the kernels of a pass only ever see these
// addresses (a captured graph is then valid for ANY caller buffers: the wrappers copy x / t / cond / grad_out in and out /
// grad_x out with plain device copies around the graph launch).
struct IoRegion {
  float *x = nullptr, *cond = nullptr, *out = nullptr, *gout = nullptr, *gx = nullptr;
  int64_t* t = nullptr;
  size_t x_bytes = 0, cond_bytes = 0, out_bytes = 0, t_bytes = 0, total = 0;
};
static IoRegion io_region(const pidm_unet* h, int B, char* base) {
  IoRegion io;
  const size_t HW = (size_t)h->cfg.image_size * h->cfg.image_size;
  io.x_bytes = (size_t)B * HW * h->init_conv.C0 * sizeof(float);
  io.cond_bytes = h->cond_enabled ? (size_t)B * HW * h->cfg.channels * sizeof(float) : 0;
  io.out_bytes = (size_t)B * HW * h->cfg.out_dim * sizeof(float);
  io.t_bytes = (size_t)B * sizeof(int64_t);
  size_t off = 0;
  auto take = [&](size_t bytes) { char* p = base ? base + off : nullptr; off += align_up(bytes, 256); return p; };
  io.x = reinterpret_cast<float*>(take(io.x_bytes));
  io.cond = reinterpret_cast<float*>(take(io.cond_bytes));
  io.out = reinterpret_cast<float*>(take(io.out_bytes));
  io.gout = reinterpret_cast<float*>(take(io.out_bytes));
  io.gx = reinterpret_cast<float*>(take(io.x_bytes));
  io.t = reinterpret_cast<int64_t*>(take(io.t_bytes));
  io.total = align_up(off, 4096);
  return io;
}
static size_t packed_region_bytes(const pidm_unet* h) {
  return align_up(h->packed_floats_total * sizeof(float) + kMaxPackDesc * sizeof(PackDesc), 4096);
}

static int setup_run(Run& r, pidm_unet* h, int B, bool train, void* workspace, size_t workspace_bytes, void* stream, bool replay_plan = false) {
  size_t tape_b, tmp_b, defer_b;
  if (replay_plan && h->plan_B == B) {
    tape_b = h->plan_tape_b; tmp_b = h->plan_tmp_b; defer_b = h->plan_defer_b;
  } else {
    if (plan_sizes(h, B, train ? 1 : 0, &tape_b, &tmp_b)) return -1;
    defer_b = train ? h->defer_cache[B] : 0;
    if (train) { h->plan_B = B; h->plan_tape_b = tape_b; h->plan_tmp_b = tmp_b; h->plan_defer_b = defer_b; }
  }
  const size_t packed_b = packed_region_bytes(h) + io_region(h, B, nullptr).total;   // the static I/O buffers follow the packed weights
  if (workspace_bytes < packed_b + tape_b + tmp_b)
    return fail("unet: workspace too small (%zu < %zu bytes)", workspace_bytes, packed_b + tape_b + tmp_b);
  if (reinterpret_cast<uintptr_t>(workspace) & 255) return fail("unet: workspace must be 256-byte aligned");
  char* w = reinterpret_cast<char*>(workspace);
  r.U = h; r.B = B; r.train = train; r.dry = false; r.st = as_stream(stream);
  r.wpack = reinterpret_cast<float*>(w);
  r.tape.base = w + packed_b; r.tape.cap = tape_b;
  r.tmp.base = w + packed_b + tape_b; r.tmp.cap = tmp_b - defer_b;
  r.defer.base = r.tmp.base + r.tmp.cap; r.defer.cap = defer_b;
  r.scratch_floats = scratch_floats_needed(h, B);
  r.scratch = r.tmp.alloc(r.scratch_floats);
  return 0;
}

}  // namespace pidm

extern "C" size_t pidm_unet_workspace_bytes(const pidm_unet* h, int B, int training) {
  size_t tape_b, tmp_b;
  if (plan_sizes(const_cast<pidm_unet*>(h), B, training, &tape_b, &tmp_b)) return 0;
  return packed_region_bytes(h) + io_region(h, B, nullptr).total + tape_b + tmp_b + 256;
}

namespace pidm {

// ---- graph cache ---------------------------------------------------------------------------------------------------------------
static uint64_t hash_words(const uint64_t* w, size_t n, uint64_t h = 1469598103934665603ull) {
  for (size_t i = 0; i < n; ++i) {
    h ^= w[i];
    h *= 1099511628211ull;
    h ^= h >> 29;
  }
  return h;
}
// every PIDM_* environment variable as the knob snapshot saw it (pidm_api.cpp: one read per process until pidm_reload_knobs())
static uint64_t env_signature() { return knob_signature(); }
static bool graphs_enabled() {
  const char* e = knob("PIDM_GRAPH");        // 0: every pass is enqueued launch by launch (A/B measurements)
  return !(e && !atoi(e)) && !prof_enabled();  // per-launch event timing needs individual launches
}
static GraphEntry* graph_find(pidm_unet* h, int kind, const std::vector<uint64_t>& key) {
  for (auto& e : h->graphs[kind])
    if (e.key == key) {
      e.stamp = ++h->graph_stamp;
      return &e;
    }
  return nullptr;
}
// true when `key` was seen before (and forgets it); remembers it otherwise: a pass is captured on its SECOND sighting, so callers
// whose buffers move every call (no stable key) simply keep the eager path
static bool graph_second_sighting(pidm_unet* h, int kind, const std::vector<uint64_t>& key) {
  auto& seen = h->seen[kind];
  for (size_t i = 0; i < seen.size(); ++i)
    if (seen[i] == key) {
      seen.erase(seen.begin() + i);
      return true;
    }
  if (seen.size() >= kMaxSeenKeys) seen.erase(seen.begin());
  seen.push_back(key);
  return false;
}
static GraphEntry* graph_new_entry(pidm_unet* h, int kind, const std::vector<uint64_t>& key) {
  auto& v = h->graphs[kind];
  if (v.size() >= kMaxGraphs) {
    size_t lru = 0;
    for (size_t i = 1; i < v.size(); ++i)
      if (v[i].stamp < v[lru].stamp) lru = i;
    graph_entry_free(v[lru]);
    v.erase(v.begin() + lru);
  }
  v.emplace_back();
  v.back().key = key;
  v.back().stamp = ++h->graph_stamp;
  return &v.back();
}
static void graph_drop_entry(pidm_unet* h, int kind, GraphEntry* e) {
  auto& v = h->graphs[kind];
  graph_entry_free(*e);
  v.erase(v.begin() + (e - v.data()));
}
static int graph_replay(GraphEntry* e, hipStream_t st) {
  for (auto& sg : e->segs) {
    if (hipGraphLaunch(sg.exec, st) != hipSuccess) return fail("graph replay: hipGraphLaunch failed (%s)", hipGetErrorString(hipGetLastError()));
    ++g_graph_launches;
    g_graph_kernels += sg.kernels;
    if (sg.ev_after && hipEventRecord(sg.ev_after, st) != hipSuccess) return fail("graph replay: phase event record failed");
  }
  return 0;
}
static bool ensure_cap_stream(pidm_unet* h) {
  if (!h->cap_stream_ok && !h->cap_stream) {
    h->cap_stream_ok = hipStreamCreateWithFlags(&h->cap_stream, hipStreamNonBlocking) == hipSuccess;
    if (!h->cap_stream_ok) {
      (void)hipGetLastError();
      h->cap_stream = reinterpret_cast<hipStream_t>(1);   // marks "decided: no capture stream"
    }
  }
  return h->cap_stream_ok;
}
static void tape_save(const pidm_unet* U, TapeState* t) {
  t->rb = U->rb; t->attn = U->attn; t->skip = U->skip; t->down_in = U->down_in; t->up_in = U->up_in;
  t->tape_B = U->tape_B; t->x_in = U->x_in; t->emb = U->emb; t->h1 = U->h1; t->h1g = U->h1g; t->temb = U->temb; t->st = U->st;
  t->ss = U->ss; t->h0 = U->h0; t->xfinal = U->xfinal; t->out_nchw = U->out_nchw; t->tape_cond = U->tape_cond; t->cond_in = U->cond_in;
  t->e1 = U->e1; t->e1g = U->e1g; t->e2 = U->e2; t->h0pre = U->h0pre;
  t->plan_B = U->plan_B; t->plan_tape_b = U->plan_tape_b; t->plan_tmp_b = U->plan_tmp_b; t->plan_defer_b = U->plan_defer_b;
}
static void tape_restore(pidm_unet* U, const TapeState& t) {
  U->rb = t.rb; U->attn = t.attn; U->skip = t.skip; U->down_in = t.down_in; U->up_in = t.up_in;
  U->tape_B = t.tape_B; U->x_in = t.x_in; U->emb = t.emb; U->h1 = t.h1; U->h1g = t.h1g; U->temb = t.temb; U->st = t.st;
  U->ss = t.ss; U->h0 = t.h0; U->xfinal = t.xfinal; U->out_nchw = t.out_nchw; U->tape_cond = t.tape_cond; U->cond_in = t.cond_in;
  U->e1 = t.e1; U->e1g = t.e1g; U->e2 = t.e2; U->h0pre = t.h0pre;
  U->plan_B = t.plan_B; U->plan_tape_b = t.plan_tape_b; U->plan_tmp_b = t.plan_tmp_b; U->plan_defer_b = t.plan_defer_b;
}
// A pass with another arena layout may overwrite the reduction descriptor table a training layout keeps on the device (an
// inference forward with a larger batch in the same workspace, ...): if its arena reaches the table, the table has to be uploaded
// again by the next backward (and no backward graph may run before that)
static void touch_arena(pidm_unet* h, int B, bool train, const void* workspace, size_t workspace_bytes) {
  const uint64_t w[4] = {(uint64_t)B, (uint64_t)train, (uint64_t)reinterpret_cast<uintptr_t>(workspace), (uint64_t)workspace_bytes};
  const uint64_t sig = hash_words(w, 4);
  if (sig != h->arena_sig && sig != h->red_table_owner && h->red_table_dev) {
    size_t tape_b = 0, tmp_b = 0;
    const char* ws = reinterpret_cast<const char*>(workspace);
    const char* tab = reinterpret_cast<const char*>(h->red_table_dev);
    const size_t packed_b = packed_region_bytes(h) + io_region(h, B, nullptr).total;
    const bool planned = plan_sizes(h, B, train ? 1 : 0, &tape_b, &tmp_b) == 0;
    const bool inside = tab >= ws && tab < ws + workspace_bytes;
    if (knob("PIDM_DEBUG_ARENA")) fprintf(stderr, "[pidm] arena touch B=%d train=%d: extent %zu, table at %zu\n", B, (int)train, packed_b + tape_b + tmp_b, (size_t)(tab - ws));
    if (!planned || !inside || ws + packed_b + tape_b + tmp_b > tab) h->red_table_dev = h->wq_table_dev = nullptr;
  }
  h->arena_sig = sig;
}

static int forward_body(pidm_unet* h, const float* x_nhwc, const int64_t* t, float* out_nchw, int B, bool train, bool repack,
                        const float* cond, void* workspace, size_t workspace_bytes, hipStream_t st, GraphCapture* cap) {
  Run r;
  r.cap = cap;
  if (setup_run(r, h, B, train, workspace, workspace_bytes, st)) return -1;
  if (cap && cap_begin(r)) return -1;
  if (repack || h->packed_zeroed_for != r.wpack || !h->pack_table_valid) {
    if (pack_all(r)) return -1;
  }
  if (forward_impl(r, x_nhwc, t, out_nchw, cond)) return -1;
  if (r.tape.overflow() || r.tmp.overflow()) return fail("unet_forward: internal arena overflow");
  if (cap && cap->open && cap_end_segment(r, nullptr, false)) return -1;
  return 0;
}

static int backward_body(pidm_unet* h, const float* grad_out_nchw, float* grad_x_nhwc, int B, void* workspace, size_t workspace_bytes,
                         hipStream_t st, GraphCapture* cap) {
  Run r;
  r.cap = cap;
  if (setup_run(r, h, B, true, workspace, workspace_bytes, st, /*replay_plan=*/true)) return -1;
  // per-kernel HIP-event timing (bench.py's roofline leg) needs kernels that own the chip: concurrent kernels share it and
  // their individual durations stop being a property of the kernel - the overlap is off while the profiler hooks are on.
  // A captured pass is linear as well: a graph with ~40 fork / join pairs replays SLOWER than the launch-by-launch form it
  // replaces (11.68 vs 11.15 ms per step, profiles/r03_graph_ab.txt; HIP maps the branches onto internal streams and
  // synchronises them with events), while the linear graph equals the eager step with the overlap (11.21 ms).
  r.side_allowed = backward_eager(widest_level(h));
  r.overlap = h->side_ok && !prof_enabled() && !cap && r.side_allowed;
  r.group_on = wgrad_group_on(B, h->cfg.image_size, r.side_allowed);
  // A pass WITHOUT grouping recycles the arena's frames and lays its partial slabs out differently: they may overwrite the grouped
  // tables a previous pass left on the device (grouping toggled on a live handle through pidm_reload_knobs) - the next grouped pass
  // must upload them again instead of trusting the host copies.
  if (!r.group_on && h->wq_table_dev) {
    h->wq_table_dev = nullptr;
    for (auto& t : h->wq_table) t.clear();
  }
  if (cap && cap_begin(r)) return -1;
  if (backward_impl(r, grad_out_nchw, grad_x_nhwc)) return -1;
  if (r.tmp.overflow() || r.defer.overflow()) return fail("unet_backward: internal arena overflow");
  if (cap && cap->open && cap_end_segment(r, nullptr, false)) return -1;   // (the final flush normally closed the last segment)
  return 0;
}

// closes a capture that an error left open, so that the stream can be used again
static void cap_abort(pidm_unet* h, GraphCapture* cap) {
  if (cap->open) {
    hipGraph_t g = nullptr;
    (void)hipStreamEndCapture(h->cap_stream, &g);
    if (g) (void)hipGraphDestroy(g);
    (void)hipGetLastError();
    cap->open = false;
  }
}

}  // namespace pidm

extern "C" int pidm_unet_forward(pidm_unet* h, const float* x_nhwc, const int64_t* t, float* out_nchw, int B,
                                 int save_for_backward, int repack_weights, void* workspace, size_t workspace_bytes,
                                 void* stream) {
  if (!h || !x_nhwc || !t || !out_nchw || !workspace) return fail("unet_forward: null argument");
  if (B <= 0) return fail("unet_forward: B must be positive");
  for (size_t i = 0; i < h->P.size(); ++i)
    if (!h->P[i]) return fail("unet_forward: parameters not bound (pidm_unet_bind)");
  const float* cond = h->cond_next;
  h->cond_next = nullptr;
  if (cond && !h->cond_enabled) return fail("unet_forward: conditioning input given but pidm_unet_enable_cond was not called");
  const bool train = save_for_backward != 0;
  const hipStream_t st = as_stream(stream);
  touch_arena(h, B, train, workspace, workspace_bytes);
  if (reinterpret_cast<uintptr_t>(workspace) & 255) return fail("unet: workspace must be 256-byte aligned");
  if (workspace_bytes < packed_region_bytes(h) + io_region(h, B, nullptr).total) return fail("unet: workspace too small");
  // the pass runs on the static I/O buffers of the workspace (see IoRegion): inputs in, output out, by plain device copies
  const IoRegion io = io_region(h, B, reinterpret_cast<char*>(workspace) + packed_region_bytes(h));
  if (x_nhwc != io.x && hipMemcpyAsync(io.x, x_nhwc, io.x_bytes, hipMemcpyDeviceToDevice, st) != hipSuccess) return fail("unet_forward: input copy failed");
  if (hipMemcpyAsync(io.t, t, io.t_bytes, hipMemcpyDeviceToDevice, st) != hipSuccess) return fail("unet_forward: time-step copy failed");
  if (cond && hipMemcpyAsync(io.cond, cond, io.cond_bytes, hipMemcpyDeviceToDevice, st) != hipSuccess) return fail("unet_forward: condition copy failed");
  const float* cond_s = cond ? io.cond : nullptr;
  const uint64_t kw[11] = {0xF0, (uint64_t)B, (uint64_t)train, (uint64_t)(repack_weights != 0), (uint64_t)(cond != nullptr),
                           (uint64_t)reinterpret_cast<uintptr_t>(workspace), (uint64_t)workspace_bytes,
                           h->param_sig, env_signature(), (uint64_t)h->cond_enabled, 0};
  const std::vector<uint64_t> key(kw, kw + 11);
  if (train) h->fwd_key_hash = hash_words(kw, 11);
  int rc = -1;
  bool done = false;
  // a graph may run only while the device-side pack descriptor table in this workspace describes the bound parameters
  const bool tables_ok = h->packed_zeroed_for == workspace && h->pack_table_valid;
  if (graphs_enabled() && tables_ok) {
    if (GraphEntry* e = graph_find(h, 0, key)) {
      rc = graph_replay(e, st);
      if (rc == 0 && train) tape_restore(h, e->tape);
      done = true;
    } else if (graph_second_sighting(h, 0, key) && ensure_cap_stream(h)) {
      e = graph_new_entry(h, 0, key);
      GraphCapture cap;
      cap.entry = e;
      cap.user_st = st;
      rc = forward_body(h, io.x, io.t, io.out, B, train, repack_weights != 0, cond_s, workspace, workspace_bytes, h->cap_stream, &cap);
      cap_abort(h, &cap);
      if (rc == 0 && !cap.failed) {
        ++g_graph_captures;
        if (train) tape_save(h, &e->tape);
        done = true;
      } else {
        graph_drop_entry(h, 0, e);     // not capturable right now: the eager path below redoes the whole pass
      }
    }
  }
  if (!done) rc = forward_body(h, io.x, io.t, io.out, B, train, repack_weights != 0, cond_s, workspace, workspace_bytes, st, nullptr);
  if (rc) return rc;
  if (out_nchw != io.out && hipMemcpyAsync(out_nchw, io.out, io.out_bytes, hipMemcpyDeviceToDevice, st) != hipSuccess) return fail("unet_forward: output copy failed");
  return 0;
}

extern "C" int pidm_unet_backward(pidm_unet* h, const float* grad_out_nchw, float* grad_x_nhwc, int B, void* workspace,
                                  size_t workspace_bytes, void* stream) {
  if (!h || !grad_out_nchw || !workspace) return fail("unet_backward: null argument");
  if (h->tape_B != B) return fail("unet_backward: no matching forward (tape holds B=%d)", h->tape_B);
  if (!h->have_grads) return fail("unet_backward: gradient buffers not bound");
  const hipStream_t st = as_stream(stream);
  if (!h->side_ok && !h->side) {
    // created once per handle; PIDM_NO_OVERLAP=1 keeps the whole backward on the caller's stream (A/B measurements)
    const char* e = knob("PIDM_NO_OVERLAP");
    if (!(e && atoi(e))) {
      h->side_ok = hipStreamCreateWithFlags(&h->side, hipStreamNonBlocking) == hipSuccess &&
                   hipEventCreateWithFlags(&h->ev_fork, hipEventDisableTiming) == hipSuccess &&
                   hipEventCreateWithFlags(&h->ev_join, hipEventDisableTiming) == hipSuccess;
    } else {
      h->side = reinterpret_cast<hipStream_t>(1);   // marks "decided: off"
    }
  }
  touch_arena(h, B, true, workspace, workspace_bytes);
  if (!h->tape_cond && h->cond_grads_dirty) {
    // the conditioning parameters were not used by this forward: their slots of the (flat) gradient buffer must not keep
    // an earlier step's values (the wrapper leaves p.grad = None for them, as the reference does).  Freshly bound
    // buffers count as dirty; after one zero-fill nothing needs to be done until the branch is used again.
    for (size_t i = (size_t)h->cond_first_param; i < h->names.size(); ++i)
      if (hipMemsetAsync(h->G[i], 0, h->numels[i] * sizeof(float), st) != hipSuccess) return fail("backward: memset failed");
    h->cond_grads_dirty = false;
  }
  const IoRegion io = io_region(h, B, reinterpret_cast<char*>(workspace) + packed_region_bytes(h));
  if (grad_out_nchw != io.gout && hipMemcpyAsync(io.gout, grad_out_nchw, io.out_bytes, hipMemcpyDeviceToDevice, st) != hipSuccess)
    return fail("unet_backward: gradient copy failed");
  float* gx_s = grad_x_nhwc ? io.gx : nullptr;
  const uint64_t kw[14] = {0xB0, (uint64_t)B, (uint64_t)(grad_x_nhwc != nullptr),
                           (uint64_t)reinterpret_cast<uintptr_t>(workspace), (uint64_t)workspace_bytes, h->bind_sig, env_signature(),
                           h->fwd_key_hash, (uint64_t)h->n_phases, (uint64_t)reinterpret_cast<uintptr_t>(h->phase_ev[0]),
                           (uint64_t)reinterpret_cast<uintptr_t>(h->phase_ev[1]), (uint64_t)reinterpret_cast<uintptr_t>(h->phase_ev[2]),
                           (uint64_t)h->tape_cond, (uint64_t)h->side_ok};
  const std::vector<uint64_t> key(kw, kw + 14);
  int rc = -1;
  bool done = false;
  if (graphs_enabled() && !backward_eager(widest_level(h))) {
    GraphEntry* e = graph_find(h, 1, key);
    // the reduction descriptor table the graph's reduce launches read must still be the one this layout uploaded
    if (e && e->red_dev == h->red_table_dev && h->red_table_dev) {
      rc = graph_replay(e, st);
      done = true;
    } else if (!e && h->red_table_dev && graph_second_sighting(h, 1, key) && ensure_cap_stream(h)) {
      e = graph_new_entry(h, 1, key);
      GraphCapture cap;
      cap.entry = e;
      cap.user_st = st;
      rc = backward_body(h, io.gout, gx_s, B, workspace, workspace_bytes, h->cap_stream, &cap);
      cap_abort(h, &cap);
      if (rc == 0 && !cap.failed) {
        ++g_graph_captures;
        e->red_dev = h->red_table_dev;
        done = true;
      } else {
        graph_drop_entry(h, 1, e);   // the eager path below redoes the whole pass
      }
    }
  }
  if (!done) rc = backward_body(h, io.gout, gx_s, B, workspace, workspace_bytes, st, nullptr);
  if (rc == 0 && h->tape_cond) h->cond_grads_dirty = true;
  if (rc == 0 && grad_x_nhwc && grad_x_nhwc != io.gx &&
      hipMemcpyAsync(grad_x_nhwc, io.gx, io.x_bytes, hipMemcpyDeviceToDevice, st) != hipSuccess)
    return fail("unet_backward: input-gradient copy failed");
  return rc;
}

// eager kernel enqueues, graph launches, kernels inside the launched graphs, captures - since the library was loaded
extern "C" int pidm_debug_launch_counts(long long* out4) {
  if (!out4) return fail("debug_launch_counts: null argument");
  out4[0] = pidm::g_kernel_enqueues;
  out4[1] = pidm::g_graph_launches;
  out4[2] = pidm::g_graph_kernels;
  out4[3] = pidm::g_graph_captures;
  return 0;
}

extern "C" int pidm_unet_set_grad_events(pidm_unet* h, int n_phases, void* const* events) {
  if (!h) return fail("unet_set_grad_events: null handle");
  if (n_phases < 1 || n_phases > 3) return fail("unet_set_grad_events: n_phases must be 1..3 (got %d)", n_phases);
  h->n_phases = n_phases;
  for (int i = 0; i < 3; ++i) h->phase_ev[i] = (events && i < n_phases) ? reinterpret_cast<hipEvent_t>(events[i]) : nullptr;
  return 0;
}

extern "C" int pidm_unet_grad_phase_range(const pidm_unet* h, int n_phases, int phase, int* first_param, int* end_param) {
  if (!h || !first_param || !end_param) return fail("unet_grad_phase_range: null argument");
  if (n_phases < 1 || n_phases > 3 || phase < 0 || phase >= n_phases) return fail("unet_grad_phase_range: bad phase %d of %d", phase, n_phases);
  const int n = (int)h->names.size();
  // canonical order: [FiLM + init + time | downs + mid | ups + final | conditioning]; the last phase finalises whatever the
  // earlier ones did not (head and conditioning tail): it is reported as ONE range only when it is contiguous (n_phases == 1),
  // otherwise as the head range - the caller treats "everything not covered by earlier phases" as the last bucket
  if (n_phases == 1) { *first_param = 0; *end_param = n; return 0; }
  if (phase == 0) { *first_param = h->idx_ups_first; *end_param = h->cond_first_param; return 0; }
  if (n_phases == 3 && phase == 1) { *first_param = h->idx_downs_first; *end_param = h->idx_ups_first; return 0; }
  *first_param = 0;
  *end_param = (n_phases == 3) ? h->idx_downs_first : h->idx_ups_first;
  return 0;
}

extern "C" long long pidm_debug_reduce_table_uploads(void) { return pidm::g_red_table_uploads; }

extern "C" int pidm_unet_num_cond_params(const pidm_unet* h) { return h ? (int)h->names.size() - h->cond_first_param : 0; }

extern "C" int pidm_unet_enable_cond(pidm_unet* h, int on) {
  if (!h) return fail("unet_enable_cond: null handle");
  const bool v = on != 0;
  if (v != h->cond_enabled) {
    h->cond_enabled = v;
    h->ws_cache[0].clear();
    h->ws_cache[1].clear();
    h->defer_cache.clear();
  }
  return 0;
}

extern "C" int pidm_unet_set_condition(pidm_unet* h, const float* cond_nhwc) {
  if (!h) return fail("unet_set_condition: null handle");
  h->cond_next = cond_nhwc;
  return 0;
}
