// forward_eval.hip — SparseCADGCN.forward (eval mode) as one host call: the complete kernel sequence
// of cad_recognition/architecture3cc_rpn_gp_iter2.py:44-71,106-137 is enqueued from C++ with no
// Python between launches (the cfg-2 forward is ~100-300 us of GPU time, i.e. launch-latency
// territory: the host side has to be a tight loop of hipLaunchKernel calls).
#include "common.hpp"

namespace {
struct Carver {
  char* base; size_t off, cap;
  template <class T> T* take(size_t n) {
    off = (off + 255) & ~(size_t)255;
    T* p = reinterpret_cast<T*>(base + off);
    off += n * sizeof(T);
    return p;
  }
};

struct Plan {
  int* row_ptr; int* perm; int* src; int* dst; float* attr; int* work; int* seg_ptr; int* node_seg;
  float* H1; float* H2; float* UV; float* UV2; float* f_tmp[YOLAT_MAX_LAYERS]; float* s_tmp[YOLAT_MAX_LAYERS];
  float* feats; float* fsup; float* Z; float* c1; float* c2;
  float* gx_work;                                      // split-K partials of cls1 on yolat_gemm_x6 (few proposals)
  uint16_t* Zs;                                        // Z pre-split for the skinny bf16x6 classifier (few proposals)
  size_t bytes;
};

Plan carve(const yolat_model_eval* m, long N, long E, long P, void* ws) {
  Carver c; c.base = reinterpret_cast<char*>(ws); c.off = 0; c.cap = 0;
  Plan p;
  const long C = m->C, F = m->F, D = C * m->n_blocks_out, Ee = E > 0 ? E : 1;
  p.row_ptr = c.take<int>(N + 1); p.perm = c.take<int>(Ee); p.src = c.take<int>(Ee); p.dst = c.take<int>(Ee);
  p.attr = c.take<float>(Ee * 4); p.work = c.take<int>(yolat_graph_work_elems(N, E));
  p.seg_ptr = c.take<int>(P + 1); p.node_seg = c.take<int>(N);
  p.H1 = c.take<float>(Ee * C); p.H2 = c.take<float>(Ee * C); p.UV = c.take<float>(N * 2 * C);
  // second UV buffer: layer l + 1's node side is written by layer l's edge launch while that launch still gathers from UV
  p.UV2 = (m->n_blocks >= 2 && yl_edge_tile_groups(N, E) == 1) ? c.take<float>(N * 2 * C) : nullptr;
  const int lo = m->n_blocks - m->n_blocks_out;
  for (int l = 0; l < m->n_blocks; ++l) {
    p.f_tmp[l] = (l < lo) ? c.take<float>(N * C) : nullptr;
    p.s_tmp[l] = (l < lo) ? c.take<float>(N * C) : nullptr;
  }
  p.feats = c.take<float>(N * D); p.fsup = c.take<float>(N * D);
  p.Z = c.take<float>(P * 2 * (F + D)); p.c1 = c.take<float>(P * m->H1); p.c2 = c.take<float>(P * m->H2);
  const size_t gxw = (m->Wc1_gx && (2 * (F + D)) % 16 == 0) ? yolat_gemm_x6_work_elems(P, m->H1, 2 * (F + D)) : 0;
  p.gx_work = gxw ? c.take<float>((long)gxw) : nullptr;
  p.Zs = (P <= YOLAT_CLS_X6_MAX_ROWS && m->Wc_x6[0] && (2 * (F + D)) % 16 == 0)
             ? c.take<uint16_t>((long)yolat_split_bf16x3_packed_elems(P, 2 * (F + D))) : nullptr;
  p.bytes = c.off + 256;
  return p;
}
}  // namespace

// yolat_graph_prepare_node_uv with the `primed` promise (graph.hip yl_graph_prepare_impl)
static int prep_node_uv(const int64_t* edge, int64_t stride_e, int64_t stride_c, const float* e_attr,
                        const int64_t* bbox_idx, int64_t E, int64_t N, int64_t P, const Plan& p, int32_t* status,
                        const float* x, int64_t ldx, int64_t Cin, const float* Wuv, const float* uv_bias, const float* Wr,
                        const float* br, const float* Wn, const float* bn, const float* sn, const float* tn, int64_t C,
                        float* f_out, int64_t ld_fo, float* s_out, int64_t ld_so, bool primed, yolat_stream_t stream) {
  NodeUv a;
  const int rc = yl_build_node_uv(&a, x, ldx, x, ldx, N, Cin, Wuv, uv_bias, Wr, br, Wn, bn, sn, tn, C, p.UV, 2 * C, f_out,
                                  ld_fo, s_out, ld_so);
  if (rc != 0) return rc;
  return yl_graph_prepare_impl(edge, stride_e, stride_c, e_attr, bbox_idx, E, N, P, p.row_ptr, p.perm, p.src, p.dst,
                               p.attr, p.seg_ptr, p.node_seg, p.work, status, &a, primed, stream);
}

#define YL_TRY(call)            \
  do {                          \
    int rc__ = (call);          \
    if (rc__ != 0) return rc__; \
  } while (0)

// ---- stage profiler (hipEvent pairs on the launch stream) ------------------------------------------
#include <stdlib.h>
#include <string>
#include <vector>
namespace {
struct StageRec { std::string name; double flops, bytes; std::vector<std::pair<hipEvent_t, hipEvent_t>> ev; };
std::vector<StageRec> g_stages;
std::vector<hipEvent_t> g_pool;
bool g_profile = false;
hipEvent_t new_event() {
  if (!g_pool.empty()) { hipEvent_t e = g_pool.back(); g_pool.pop_back(); return e; }
  hipEvent_t e; (void)hipEventCreate(&e); return e;
}
// flops / bytes are SUMMED over the calls of a stage and reported per call (yolat_profile_get): the conv layers of a forward
// share one stage name but not one size (the head layer's Cin is the raw feature width)
StageRec& stage_rec(const char* name, double flops, double bytes) {
  for (auto& s : g_stages) if (s.name == name) { s.flops += flops; s.bytes += bytes; return s; }
  g_stages.push_back(StageRec{name, flops, bytes, {}});
  return g_stages.back();
}
}  // namespace

#define YL_STAGE(name, flops, bytes, call)                         \
  do {                                                                      \
    if (g_profile) {                                                        \
      StageRec& sr__ = stage_rec(name, (double)(flops), (double)(bytes));   \
      hipEvent_t a__ = new_event(), b__ = new_event();                      \
      (void)hipEventRecord(a__, (hipStream_t)stream);                       \
      int rc__ = (call);                                                    \
      (void)hipEventRecord(b__, (hipStream_t)stream);                       \
      sr__.ev.push_back({a__, b__});                                        \
      if (rc__ != 0) return rc__;                                           \
    } else {                                                                \
      YL_TRY(call);                                                         \
    }                                                                       \
  } while (0)

// stage hooks for the other whole-forward entry points (bf16_eval.hip)
namespace { int g_open_stage = -1; }
bool yl_profile_on() { return g_profile; }
void yl_stage_begin(const char* name, double flops, double bytes, yolat_stream_t stream) {
  stage_rec(name, flops, bytes);
  for (size_t i = 0; i < g_stages.size(); ++i) if (g_stages[i].name == name) g_open_stage = (int)i;
  hipEvent_t a = new_event(), b = new_event();
  (void)hipEventRecord(a, (hipStream_t)stream);
  g_stages[g_open_stage].ev.push_back({a, b});
}
void yl_stage_end(yolat_stream_t stream) {
  if (g_open_stage >= 0) (void)hipEventRecord(g_stages[g_open_stage].ev.back().second, (hipStream_t)stream);
  g_open_stage = -1;
}

extern "C" int yolat_profile_enable(int on) { g_profile = on != 0; return 0; }
extern "C" int yolat_profile_enabled(void) { return g_profile ? 1 : 0; }
extern "C" int yolat_profile_reset(void) {
  for (auto& s : g_stages) for (auto& e : s.ev) { g_pool.push_back(e.first); g_pool.push_back(e.second); }
  g_stages.clear();
  return 0;
}
extern "C" int yolat_profile_count(void) { return (int)g_stages.size(); }
extern "C" int yolat_profile_get(int index, char* name, int cap, float* total_ms, int* calls, double* flops,
                                 double* bytes) {
  if (index < 0 || index >= (int)g_stages.size() || !name || cap <= 0) return YOLAT_E_INVALID;
  StageRec& s = g_stages[index];
  snprintf(name, cap, "%s", s.name.c_str());
  float tot = 0.f;
  for (auto& e : s.ev) { float ms = 0.f; if (hipEventElapsedTime(&ms, e.first, e.second) == hipSuccess) tot += ms; }
  if (total_ms) *total_ms = tot;
  if (calls) *calls = (int)s.ev.size();
  const double n = s.ev.empty() ? 1.0 : (double)s.ev.size();
  if (flops) *flops = s.flops / n;
  if (bytes) *bytes = s.bytes / n;
  return 0;
}

// cls1 on the skinny bf16x6 kernel: K = 2304 is long enough that splitting Z once (a small launch) beats splitting it
// on the fly in each of the H1/32 column tiles
static int cls1_x6(const float* Z, long ZW, long P, uint16_t* Zs, const uint16_t* Wp, const float* shift, long H1,
                   float* c1, yolat_stream_t stream) {
  YL_TRY(yolat_split_bf16x3_packed(Z, ZW, P, ZW, nullptr, Zs, stream));
  return yolat_linear_x6_pre(Zs, P, ZW, Wp, shift, 1, H1, c1, H1, stream);
}

extern "C" size_t yolat_forward_eval_workspace_bytes(const yolat_model_eval* m, int64_t N, int64_t E,
                                                     int64_t P) {
  if (!m || N <= 0 || E < 0 || P <= 0) return 0;
  return carve(m, N, E, P, nullptr).bytes;
}

static int forward_eval_impl(const yolat_model_eval* m, const float* x, int64_t ldx, const int64_t* edge,
                             int64_t stride_e, int64_t stride_c, const float* e_attr, const int64_t* bbox_idx,
                             int64_t N, int64_t E, int64_t P, float* logits, int64_t ld_logits, void* workspace,
                             size_t workspace_bytes, int32_t* status, bool primed, yolat_stream_t stream,
                             const yolat_graph_csr* g = nullptr);

// The forward on a prepared device graph (collate.hip: the batch's CSR merged from the items' cached CSRs, one H2D copy):
// no COO -> CSR conversion, the first layer's node side as its own launch.
extern "C" int yolat_forward_eval_csr(const yolat_model_eval* m, const float* x, int64_t ldx, const yolat_graph_csr* g,
                                      int64_t N, int64_t E, int64_t P, float* logits, int64_t ld_logits, void* workspace,
                                      size_t workspace_bytes, yolat_stream_t stream) {
  if (!g || !g->row_ptr || !g->seg_ptr || !g->node_seg || (E > 0 && (!g->src || !g->dst || !g->attr)))
    return YOLAT_E_INVALID;
  int32_t unused_status = 0;
  return forward_eval_impl(m, x, ldx, nullptr, 0, 0, nullptr, reinterpret_cast<const int64_t*>(g->node_seg), N, E, P, logits,
                           ld_logits, workspace, workspace_bytes, &unused_status, false, stream, g);
}

extern "C" int yolat_forward_eval(const yolat_model_eval* m, const float* x, int64_t ldx,
                                  const int64_t* edge, int64_t stride_e, int64_t stride_c,
                                  const float* e_attr, const int64_t* bbox_idx, int64_t N, int64_t E,
                                  int64_t P, float* logits, int64_t ld_logits, void* workspace,
                                  size_t workspace_bytes, int32_t* status, yolat_stream_t stream) {
  return forward_eval_impl(m, x, ldx, edge, stride_e, stride_c, e_attr, bbox_idx, N, E, P, logits, ld_logits, workspace,
                           workspace_bytes, status, false, stream);
}

// The same forward for a caller that keeps `workspace` to itself: the previous call on this workspace was
// yolat_forward_eval / yolat_forward_eval_primed with the SAME m (layout), N, E, P and has been enqueued on the same
// stream (or completed).  Every forward leaves the CSR-build counters zero, so this one skips their memset launch
// (3.8 us of the 120 us cfg-2 forward).  Anything else written into the workspace in between voids the promise.
extern "C" int yolat_forward_eval_primed(const yolat_model_eval* m, const float* x, int64_t ldx,
                                         const int64_t* edge, int64_t stride_e, int64_t stride_c,
                                         const float* e_attr, const int64_t* bbox_idx, int64_t N, int64_t E,
                                         int64_t P, float* logits, int64_t ld_logits, void* workspace,
                                         size_t workspace_bytes, int32_t* status, yolat_stream_t stream) {
  return forward_eval_impl(m, x, ldx, edge, stride_e, stride_c, e_attr, bbox_idx, N, E, P, logits, ld_logits, workspace,
                           workspace_bytes, status, true, stream);
}

static int forward_eval_impl(const yolat_model_eval* m, const float* x, int64_t ldx, const int64_t* edge,
                             int64_t stride_e, int64_t stride_c, const float* e_attr, const int64_t* bbox_idx,
                             int64_t N, int64_t E, int64_t P, float* logits, int64_t ld_logits, void* workspace,
                             size_t workspace_bytes, int32_t* status, bool primed, yolat_stream_t stream,
                             const yolat_graph_csr* g) {
  if (!m || !x || !bbox_idx || !logits || !workspace || !status || N <= 0 || E < 0 || P <= 0)
    return YOLAT_E_INVALID;
  if (m->n_blocks < 1 || m->n_blocks > YOLAT_MAX_LAYERS || m->n_blocks_out < 1 ||
      m->n_blocks_out > m->n_blocks)
    return YOLAT_E_INVALID;
  Plan p = carve(m, N, E, P, workspace);
  if (p.bytes > workspace_bytes) return YOLAT_E_INVALID;
  if (g != nullptr) {                 // prepared graph: the kernels read the caller's arrays (never written here)
    p.row_ptr = const_cast<int*>(g->row_ptr); p.src = const_cast<int*>(g->src); p.dst = const_cast<int*>(g->dst);
    p.attr = const_cast<float*>(g->attr); p.seg_ptr = const_cast<int*>(g->seg_ptr);
    p.node_seg = const_cast<int*>(g->node_seg);
  }
  const long C = m->C, F = m->F, D = C * m->n_blocks_out, ZW = 2 * (F + D);
  const int lo = m->n_blocks - m->n_blocks_out;

  // ---- graph structure (CSR by destination, e_attr in CSR order, proposal segments)
  char nm[128];
  // The node side of layer 0 (UV products, root Linear, node-branch Linear) reads only x: its GEMM tiles are
  // co-scheduled with the last, latency-bound pre-processing launch instead of being a launch of their own.
  const yolat_conv_eval& cv0 = m->conv[0];
  const bool node0_ok = C == 64 && cv0.Wuv != nullptr && cv0.Wc4 != nullptr;
  const bool node0_in_prep = node0_ok && g == nullptr;
  const bool fold0 = cv0.Wuvf && cv0.uvb && cv0.Wc4f && cv0.t2f;
  // One launch per conv layer (small graphs on the <= 16-node tiles, every layer factorised + folded, packed next-layer
  // weights present): layer l's edge launch computes layer l + 1's UV / root rows for its own nodes and, in extra
  // workgroups, layer l + 1's node branch (EdgeNext, common.hpp) — k_gemm_nt_node3 is not launched at all.
  // YOLAT_NODE_CHAIN=0: node side as a launch per layer.
  static const bool chain_on = []() { const char* v = getenv("YOLAT_NODE_CHAIN"); return !(v && v[0] == '0'); }();
  bool chain_mode = chain_on && node0_ok && fold0 && E > 0 && m->n_blocks >= 2 && p.UV2 != nullptr &&
                    yl_edge_tile_groups(N, E) == 1;
  for (int l = 1; l < m->n_blocks && chain_mode; ++l) {
    const yolat_conv_eval& c = m->conv[l];
    chain_mode = c.Cin == 64 && c.Wuv && c.Wc4 && c.Wuvf && c.uvb && c.Wc4f && c.t2f && c.Wnx && c.tnx && c.Wn &&
                 c.sn && c.tn && yl_aligned16(c.Wn);
  }
  bool next_done = false;             // the previous edge launch computed this layer's UV / root ...
  float* uv_next = nullptr;           // ... into this buffer
  if (node0_in_prep) {
    const int slot0 = 0 - lo;
    float* f0 = slot0 >= 0 ? p.feats + slot0 * C : p.f_tmp[0];
    float* s0 = slot0 >= 0 ? p.fsup + slot0 * C : p.s_tmp[0];
    const long ld0 = slot0 >= 0 ? D : C;
    YL_STAGE("graph_prep[csr+attr+segments] + node_uv[layer 0]", 8.0 * N * cv0.Cin * C,
             16.0 * E + 12.0 * E + 32.0 * E + 12.0 * N + 4.0 * (2.0 * N * cv0.Cin + 4.0 * N * C),
             prep_node_uv(edge, stride_e, stride_c, e_attr, bbox_idx, E, N, P, p, status, x, ldx, cv0.Cin,
                          fold0 ? cv0.Wuvf : cv0.Wuv, fold0 ? cv0.uvb : nullptr, cv0.Wr, cv0.br, cv0.Wn, cv0.bn, cv0.sn,
                          cv0.tn, C, f0, ld0, s0, ld0, primed, stream));
  } else if (g == nullptr) {
  YL_STAGE("graph_prep[csr+attr+segments]", 0, 16.0 * E + 12.0 * E + 32.0 * E + 12.0 * N,
           yl_graph_prepare_impl(edge, stride_e, stride_c, e_attr, bbox_idx, E, N, P, p.row_ptr, p.perm, p.src,
                                 p.dst, p.attr, p.seg_ptr, p.node_seg, p.work, status, nullptr, primed, stream));
  }

  // ---- pooling prologue (segment.hip k_pool_prepare): a launch of its own for large graphs; for small ones
  // (YOLAT_POOL_RIDERS != 0, P * (F + 2 D) below ~1 M items) its parts ride in the last edge launch and the fusion
  // launch (common.hpp PoolRider) — at cfg 2 that launch is 5 us of pure latency
  static const bool riders_on = []() { const char* v = getenv("YOLAT_POOL_RIDERS"); return !(v && v[0] == '0'); }();
  const bool fusion_x6 = m->Wf_hi && m->Wf_mid && m->Wf_lo && m->tf_fold && m->Wfs_hi && m->Wfs_mid && m->Wfs_lo &&
                         m->tfs_fold && (D == 64 || D == 128) && F % 64 == 0 && (long)P * ZW < (1LL << 32);
  const bool small_pool = riders_on && (long)P * (F + 2 * D) <= (1L << 20);
  PoolRider ride_a{}, ride_b{};
  ride_a.feats = ride_b.feats = p.feats; ride_a.fsup = ride_b.fsup = p.fsup; ride_a.ld = ride_b.ld = D;
  ride_a.D = ride_b.D = (int)D; ride_a.F = ride_b.F = (int)F; ride_a.P = ride_b.P = (int)P;
  ride_a.seg_ptr = ride_b.seg_ptr = p.seg_ptr; ride_a.Z = ride_b.Z = p.Z; ride_a.ldz = ride_b.ldz = ZW;
  ride_a.parts = YL_POOL_ZERO | YL_POOL_MEAN; ride_a.blocks = small_pool && fusion_x6 ? 256 : 0;
  ride_b.parts = YL_POOL_MAX; ride_b.blocks = small_pool && fusion_x6 ? 64 : 0;
  int pool_done = 0;

  // ---- conv layers (torch_vertex.py:319-337), outputs written into their concat slots
  const float* f_in = x; long ld_f = ldx;
  const float* s_in = x; long ld_s = ldx;
  for (int l = 0; l < m->n_blocks; ++l) {
    const yolat_conv_eval& cv = m->conv[l];
    const int slot = l - lo;
    float* f_out = slot >= 0 ? p.feats + slot * C : p.f_tmp[l];
    float* s_out = slot >= 0 ? p.fsup + slot * C : p.s_tmp[l];
    const long ld_out = slot >= 0 ? D : C;
    const double K1 = 2.0 * cv.Cin + 4;
    if (C == 64) {
      if (cv.Wuv != nullptr && cv.Wc4 != nullptr) {
        // factorised layer, two launches: (1) node side — UV = f_in.[W1a-W1b | W1b]^T, root Linear and
        // node-branch Linear in one launch; (2) per destination-node tile: gather-add of U[dst] + V[src] +
        // W1c.attr, BN+ReLU, second edge Linear, BN+ReLU and the CSR mean, accumulated into the root output
        // (neither [E,64] activation reaches HBM).  The K = 2*Cin GEMM runs once per node instead of once
        // per edge (E = 4..6 N).
        // folded form (yolat_conv_eval.{Wuvf, uvb, Wc4f, t2f}): layer 1's bias / BatchNorm ride in the node-side
        // epilogue and the scaled weights, layer 2's bias in t2f
        const bool fold = cv.Wuvf && cv.uvb && cv.Wc4f && cv.t2f;
        float* uv_l = next_done ? uv_next : p.UV;
        if (!(l == 0 && node0_in_prep) && !next_done) {
          uv_l = p.UV;
          snprintf(nm, sizeof nm, "node_uv[UV | lin_r | mlp_node, N x %ld -> %ld+%ld+%ld]", (long)cv.Cin, 2 * C, C, C);
          // large graphs: the three GEMMs as bf16x6-emulated products on the rows kernel (A read once per 256 rows);
          // small ones stay on the fp32 64x64 tiles (more workgroups: latency).  Measured per launch: N = 200 k 94 -> 89 us,
          // N = 43.5 k 26 = 26 us, N = 10 k 12.4 -> 18 us
          const bool node_x6 = fold && cv.Cin == 64 && N >= YOLAT_NODE_X6_MIN_ROWS && cv.Wfr_x6[0] && cv.Wfr_x6[1] &&
                               cv.Wfr_x6[2] && cv.tfr && cv.Wn_x6[0] && cv.Wn_x6[1] && cv.Wn_x6[2] && cv.tn_fold &&
                               ld_f % 4 == 0 && ld_s % 4 == 0;
          YL_STAGE(nm, 8.0 * N * cv.Cin * C, 4.0 * (2.0 * N * cv.Cin + 4.0 * N * C),
                   node_x6 ? yolat_node_uv_eval_x6(f_in, ld_f, s_in, ld_s, N, cv.Wfr_x6[0], cv.Wfr_x6[1], cv.Wfr_x6[2],
                                                   cv.tfr, cv.Wn_x6[0], cv.Wn_x6[1], cv.Wn_x6[2], cv.tn_fold, p.UV, 2 * C,
                                                   f_out, ld_out, s_out, ld_out, stream)
                           : yolat_node_uv_eval(f_in, ld_f, s_in, ld_s, N, cv.Cin, fold ? cv.Wuvf : cv.Wuv,
                                                fold ? cv.uvb : nullptr, cv.Wr, cv.br, cv.Wn, cv.bn, cv.sn, cv.tn, C, p.UV,
                                                2 * C, f_out, ld_out, s_out, ld_out, stream));
        }
        if (E > 0) {
          // last layer of a small graph: the parts of the pooling prologue that do not need this layer's messages
          // (zero the pooled maxima, per-proposal mean of the node branch) ride in this launch
          const PoolRider* rd = (l == m->n_blocks - 1 && ride_a.blocks > 0) ? &ride_a : nullptr;
          int rode = 0, did_next = 0;
          EdgeNext nx{};
          if (chain_mode && l + 1 < m->n_blocks) {
            const int sn = l + 1 - lo;
            nx.Wp = m->conv[l + 1].Wnx; nx.bias = m->conv[l + 1].tnx;
            nx.UV = (uv_l == p.UV) ? p.UV2 : p.UV; nx.ld_uv = 2 * C;
            nx.root = sn >= 0 ? p.feats + sn * C : p.f_tmp[l + 1]; nx.ld_root = sn >= 0 ? D : C;
            const yolat_conv_eval& cn = m->conv[l + 1];
            nx.s_in = s_out; nx.ld_si = ld_out;                 // this layer's node branch: written one launch earlier
            nx.Wn = cn.Wn; nx.bn = cn.bn; nx.sn = cn.sn; nx.tn = cn.tn;
            nx.s_out = sn >= 0 ? p.fsup + sn * C : p.s_tmp[l + 1]; nx.ld_so = sn >= 0 ? D : C;
            nx.s_tiles = yl_cdiv(N, 64);
          }
          // a launch that also carries the next layer's node side is a stage of its own (its algorithmic work includes
          // that GEMM set: reads s, writes UV' / root' / s')
          const bool with_next = nx.Wp != nullptr;
          if (with_next)
            snprintf(nm, sizeof nm, "edge_uv_mlp2_mean+node_uv_next[E x (U+V+attr) -> %ld -> %ld -> mean; N x %ld -> %ld+%ld+%ld]",
                     C, C, C, 2 * C, C, C);
          else
            snprintf(nm, sizeof nm, "edge_uv_mlp2_mean[E x (U+V+attr) -> %ld -> %ld -> mean]", C, C);
          // bytes: SURVEY.md 8(d) B_agg(l) of the UNFACTORISED layer, E ((2 Cin + 4) 4 + 2 * 4) + N C 4 — the credit figure
          // (what the kernel has to move is priced in bench.py executed_pricing)
          YL_STAGE(nm, 2.0 * E * (4.0 * C + C * C) + (with_next ? 8.0 * N * C * C : 0.0),
                   E * ((2.0 * cv.Cin + 4.0) * 4.0 + 8.0) + 4.0 * N * C + (with_next ? 4.0 * (N * C + 4.0 * N * C) : 0.0),
                   fold ? yl_edge_uv_mlp2_mean_eval_impl(uv_l, 2 * C, p.src, p.dst, p.attr, p.row_ptr, N, E, cv.Wc4f,
                                                         nullptr, nullptr, nullptr, cv.W2, nullptr, cv.s2, cv.t2f, C,
                                                         f_out, ld_out, YOLAT_EDGE_AUTO, rd, &rode,
                                                         nx.Wp ? &nx : nullptr, &did_next, stream)
                        : yl_edge_uv_mlp2_mean_eval_impl(uv_l, 2 * C, p.src, p.dst, p.attr, p.row_ptr, N, E, cv.Wc4,
                                                         cv.b1, cv.s1, cv.t1, cv.W2, cv.b2, cv.s2, cv.t2, C, f_out,
                                                         ld_out, YOLAT_EDGE_AUTO, rd, &rode, nullptr, nullptr, stream));
          if (rode) pool_done |= ride_a.parts;
          // (had the launch not taken `next`, the following layer's node side runs as its own launch into p.UV,
          // recomputing the node branch the chain already wrote — same values)
          next_done = did_next != 0;
          uv_next = nx.UV;
        }
      } else {
      // three launches per layer: edge MLP (hidden activation in LDS); root Linear | node-branch Linear as one
      // paired GEMM launch; CSR mean accumulated into the root output.
      if (E > 0) {
        snprintf(nm, sizeof nm, "edge_mlp2[E x %ld -> %ld -> %ld, gathered]", (long)K1, C, C);
        YL_STAGE(nm, 2.0 * E * (K1 * C + C * C), E * (K1 * 4.0 + 8.0) + 4.0 * E * C,
                 yolat_edge_mlp2_eval(f_in, ld_f, N, cv.Cin, p.src, p.dst, p.attr, E, cv.W1, cv.b1, cv.s1, cv.t1,
                                      cv.W2, cv.b2, cv.s2, cv.t2, C, p.H2, C, stream));
      }
      snprintf(nm, sizeof nm, "node_pair[lin_r | mlp_node, N x %ld -> %ld]", (long)cv.Cin, C);
      YL_STAGE(nm, 4.0 * N * cv.Cin * C, 8.0 * (N * cv.Cin + N * C),
               yolat_node_side_eval(f_in, ld_f, s_in, ld_s, N, cv.Cin, cv.Wr, cv.br, cv.Wn, cv.bn, cv.sn, cv.tn,
                                    nullptr, C, p.row_ptr, 0, C, f_out, ld_out, s_out, ld_out, stream));
      if (E > 0) {
        YL_STAGE("csr_mean[E x C -> N x C]", 1.0 * E * C, 4.0 * (E * C + 2.0 * N * C) + 4.0 * N,
                 yolat_csr_mean_fwd(p.H2, C, C, nullptr, nullptr, 0, p.row_ptr, N, f_out, ld_out, 1, stream));
      }
      }
    } else {
    // out = lin_r(x)
    snprintf(nm, sizeof nm, "lin_r[N x %ld -> %ld]", (long)cv.Cin, C);
    YL_STAGE(nm, 2.0 * N * cv.Cin * C, 4.0 * (N * cv.Cin + N * C + C * cv.Cin),
             yolat_linear_fwd(f_in, ld_f, N, cv.Cin, nullptr, nullptr, 0, cv.Wr, cv.Cin, cv.br, C, nullptr,
                              nullptr, 0, f_out, ld_out, 0, nullptr, stream));
    if (E > 0) {
      snprintf(nm, sizeof nm, "edge_lin1[E x %ld -> %ld, gathered]", (long)K1, C);
      YL_STAGE(nm, 2.0 * E * K1 * C, E * (K1 * 4.0 + 8.0) + 4.0 * E * C,
               yolat_edge_lin1_fwd(f_in, ld_f, N, cv.Cin, p.src, p.dst, p.attr, E, cv.W1, 2 * cv.Cin + 4,
                                   cv.b1, C, cv.s1, cv.t1, 1, p.H1, C, nullptr, stream));
      snprintf(nm, sizeof nm, "edge_lin2[E x %ld -> %ld]", C, C);
      YL_STAGE(nm, 2.0 * E * C * C, 8.0 * E * C,
               yolat_linear_fwd(p.H1, C, E, C, nullptr, nullptr, 0, cv.W2, C, cv.b2, C, cv.s2, cv.t2, 1, p.H2,
                                C, 0, nullptr, stream));
      YL_STAGE("csr_mean[E x C -> N x C]", 1.0 * E * C, 4.0 * (E * C + 2.0 * N * C) + 4.0 * N,
               yolat_csr_mean_fwd(p.H2, C, C, nullptr, nullptr, 0, p.row_ptr, N, f_out, ld_out, 1, stream));
    }
    // node branch
    snprintf(nm, sizeof nm, "mlp_node[N x %ld -> %ld]", (long)cv.Cin, C);
    YL_STAGE(nm, 2.0 * N * cv.Cin * C, 4.0 * (N * cv.Cin + N * C + C * cv.Cin),
             yolat_linear_fwd(s_in, ld_s, N, cv.Cin, nullptr, nullptr, 0, cv.Wn, cv.Cin, cv.bn, C, cv.sn, cv.tn, 1,
                              s_out, ld_out, 0, nullptr, stream));
    }
    f_in = f_out; ld_f = ld_out; s_in = s_out; ld_s = ld_out;
  }

  // ---- fusion over nodes + per-proposal max (arch:61-63,122)
  float* sup = p.Z + 2 * F + D;
  const bool ride_max = ride_b.blocks > 0 && (pool_done & YL_POOL_ZERO);       // riders all the way, or not at all
  const int pool_left = (YL_POOL_ZERO | YL_POOL_MAX | YL_POOL_MEAN) & ~pool_done & ~(ride_max ? YL_POOL_MAX : 0);
  if (pool_left) {
    YL_STAGE("pool_prepare[max(feats), mean(fsup), zero]", 2.0 * N * D, 8.0 * N * D + 4.0 * P * (F + 2 * D),
             yl_pool_prepare_parts(p.feats, p.fsup, D, D, F, p.seg_ptr, P, p.Z, ZW, pool_left, stream));
  }
  // fusion block over the nodes + per-proposal max, and fusion_block_super over the per-proposal means
  // (arch:61-63,65-69,122): two independent GEMMs in one flattened launch
  snprintf(nm, sizeof nm, "fusion_gemm+segmax[N x %ld -> %ld -> P] | super[P x %ld -> %ld]", D, F, D, F);
  YL_STAGE(nm, 2.0 * (N + P) * D * F, 4.0 * (N * D + 2.0 * D * F + 2.0 * P * F + N + P * D),
           fusion_x6 ? yl_fusion_pair_eval_x6_impl(p.feats, D, N, D, m->Wf_hi, m->Wf_mid, m->Wf_lo, m->tf_fold, F,
                                                   p.node_seg, p.Z, ZW, sup, ZW, P, m->Wfs_hi, m->Wfs_mid, m->Wfs_lo,
                                                   m->tfs_fold, p.Z + F + D, ZW, ride_max ? &ride_b : nullptr, stream)
                     : yolat_fusion_pair_eval(p.feats, D, N, D, m->Wf, m->bf, m->sf, m->tf, F, p.node_seg, p.Z, ZW, sup,
                                              ZW, P, m->Wfs, m->bfs, m->sfs, m->tfs, p.Z + F + D, ZW, stream));
  // ---- classifier (arch:91-93,127-128)
  // few proposals: the skinny bf16x6 kernel (one workgroup per 32 x 32 outputs, weights re-read per 32 rows)
  bool cls_x6 = p.Zs != nullptr && m->H1 % 16 == 0 && m->H2 % 16 == 0;
  for (int i = 0; i < 3; ++i) cls_x6 = cls_x6 && m->Wc_x6[i] != nullptr && m->tc_fold[i] != nullptr;
  // the bf16x6 GEMM wins once there are enough 128-row tiles to run without a deep K split (measured: P = 2000
  // 41 vs 58 us, P = 8000 129 vs 187 us; P = 400: 21 us either way — profiles/r02_gemm_x6_cls1.txt)
  const bool cls1_gx = m->Wc1_gx && m->tc1_gx && ZW % 16 == 0 && P >= YOLAT_CLS1_X6_MIN_ROWS;
  snprintf(nm, sizeof nm, "cls1[P x %ld -> %ld]", ZW, (long)m->H1);
  YL_STAGE(nm, 2.0 * P * ZW * m->H1, 4.0 * (P * ZW + ZW * m->H1 + P * m->H1),
           cls1_gx ? yolat_gemm_x6(p.Z, ZW, P, ZW, m->Wc1_gx, m->tc1_gx, 1, m->H1, p.c1, m->H1, p.gx_work, stream)
           : cls_x6 ? cls1_x6(p.Z, ZW, P, p.Zs, m->Wc_x6[0], m->tc_fold[0], m->H1, p.c1, stream)
                  : yolat_linear_fwd(p.Z, ZW, P, ZW, nullptr, nullptr, 0, m->Wc1, ZW, m->bc1, m->H1, m->sc1, m->tc1, 1,
                                     p.c1, m->H1, 0, nullptr, stream));
  snprintf(nm, sizeof nm, "cls2[P x %ld -> %ld]", (long)m->H1, (long)m->H2);
  YL_STAGE(nm, 2.0 * P * m->H1 * m->H2, 4.0 * (P * m->H1 + m->H1 * m->H2 + P * m->H2),
           cls_x6 ? yolat_linear_x6(p.c1, m->H1, P, m->H1, m->Wc_x6[1], m->tc_fold[1], 1, m->H2, p.c2, m->H2, stream)
                  : yolat_linear_fwd(p.c1, m->H1, P, m->H1, nullptr, nullptr, 0, m->Wc2, m->H1, m->bc2, m->H2, m->sc2,
                                     m->tc2, 1, p.c2, m->H2, 0, nullptr, stream));
  snprintf(nm, sizeof nm, "cls3[P x %ld -> %d]", (long)m->H2, (int)m->n_classes);
  YL_STAGE(nm, 2.0 * P * m->H2 * m->n_classes, 4.0 * (P * m->H2 + m->H2 * m->n_classes + P * m->n_classes),
           cls_x6 ? yolat_linear_x6(p.c2, m->H2, P, m->H2, m->Wc_x6[2], m->tc_fold[2], 0, m->n_classes, logits,
                                    ld_logits, stream)
                  : yolat_linear_fwd(p.c2, m->H2, P, m->H2, nullptr, nullptr, 0, m->Wc3, m->H2, m->bc3, m->n_classes,
                                     nullptr, nullptr, 0, logits, ld_logits, 0, nullptr, stream));
  return 0;
}
