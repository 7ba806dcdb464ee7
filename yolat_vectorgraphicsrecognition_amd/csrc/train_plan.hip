// train_plan.hip — the training step as ONE native call (round 6): graph preparation + forward + CrossEntropy + backward
// (+ Adam) enqueued from C on two streams, exactly the schedule engine.py issues from Python (same kernels, same operands,
// same order per stream -> bit-identical losses, gradients and parameters), without ~150-230 ctypes calls, tensor
// allocations and record_stream marks per step on the host.
//
// Reference: the loop body of cad_recognition/train.py:263-284 (forward, loss, backward, optimizer.step) over
// SparseCADGCN.forward (architecture3cc_rpn_gp_iter2.py:44-71,106-137), AttrRelativeEdgeConvGlobalPool2
// (gcn_lib/sparse/torch_vertex.py:288-341), MLP (torch_nn.py:50-71), DetectionLoss (arch:358-379), torch.optim.Adam
// (train.py:212).  Every launch below goes through the library's own C entry points (include/yolat_hip.h) — this file is
// the HOST schedule only; the kernels are where they were.
//
// Covered: the reference recipe's shapes on the default schedule of engine.py — n_filters 64, n_blocks_out 2 (fusion dims
// 128: the fused fusion block), Linear biases and BatchNorm everywhere, no dropout, softmax classifier, E >= N (the
// factorised backward of the first edge Linear) — fp32 and bf16 storage of the per-edge tensors.  Anything else returns
// YOLAT_E_UNSUPPORTED and the caller keeps the Python schedule (trainer.Trainer does).
//
// Streams.  `side` != NULL: the weight gradients and the node branches run on it beside the dX chain, forked behind an
// event on `stream` at every hand-over and joined in front of the classifier, before the head bucket is declared complete
// and at the end of the backward — engine._on_side / _join_side.  The side stream also takes what the Python schedule
// leaves on the main one although nothing on the critical path waits for it: weight-only / graph-only preparation at the
// start of the step, the pooling of feats / the node branches and fusion_block_super beside the fusion block (forward),
// fusion_block_super's backward and the per-proposal mean's beside the fusion block's backward.  Same kernels, same
// operands, so the results stay bit-identical; only WHERE a launch waits changes.  Every temporary has its own range of
// the workspace (nothing is recycled inside a step), so the two streams never share scratch.
//
// Phases (for the data-parallel exchange, which stays with the caller's process group): 1 = everything up to the point
// where the gradients of the fusion blocks and the classifier (the "head" bucket, 93 % of the bytes) are final on `stream`;
// 2 = the conv layers' backward; 4 = Adam.  The caller issues its all-reduces between the calls.
#include "common.hpp"
#include <stdlib.h>
#include <string.h>

#include <mutex>

namespace {

struct Carve {
  char* base; size_t off;
  template <class T> T* take(size_t n) {
    off = (off + 255) & ~(size_t)255;
    T* p = base ? reinterpret_cast<T*>(base + off) : nullptr;
    off += (n > 0 ? n : 1) * sizeof(T);
    return p;
  }
};

constexpr int TP_MAXL = YOLAT_MAX_LAYERS;

struct ConvBuf {
  bool fact_fwd, half;
  float* f_tmp; float* s_tmp;              // outputs of layers below the concat (slot < 0)
  void* H1; void* H2; float* st1; float* st2; float* c1; float* c2;
  float* wuv; float* wc4; float* uv; uint16_t* wwork_f;
  float* st_n; float* cn;                  // node branch: statistics, coef [4C] (scale / shift may live in sup_coef)
  float* cn_scale; float* cn_shift;
  // backward
  void* dA1; float* coef2; float* coef1; float* w_stats; float* w_l2; float* dUV; float* dwc4; float* w_apply;
  float* wuv_b; float* wc4_b; float* dwuv; float* w_dwuv; float* w_root; float* w_node_bn; float* w_node_w;
  float* dx_tmp; float* dxn_tmp;
};

struct TrainBuf {
  // graph
  int* row_ptr; int* perm; int* src; int* dst; float* attr; int* zblock; int* seg_ptr; int* node_seg; int* gwork;
  int* col_ptr; int* slots; int* cwork; float* inv_deg;
  float* feats; float* fsup; float* sup_coef;
  ConvBuf cv[TP_MAXL];
  float* Z; int* arg_feat; float* fus_coef; float* fus_saved; float* fus_work;
  float* fs_y; float* fs_st; float* fs_c;
  float* c1y; float* c1st; float* c1c; uint16_t* c1pack;
  float* c2y; float* c2st; float* c2c;
  float* dl; float* ce_work;
  // backward
  float* d2; float* d1; float* dZ; float* d_fsup; float* d_feats;
  float* w3; float* w2bn; float* w2w; uint16_t* p2; float* x2w; float* w1bn; float* w1w; uint16_t* p1; float* x1w;
  float* wfsbn; float* wfsw;
  size_t bytes;
};

bool strict_fp32() {
  const char* e = getenv("YOLAT_STRICT_FP32");
  return e && e[0] == '1';
}

// ops.linear_fwd's choice of the bf16x6 LDS-tiled GEMM (many rows x long K in front of a training BatchNorm)
bool x6_fwd(const float* A, long lda, long M, long K, long Nout, bool a_pro, bool stats, bool bias) {
  return !strict_fp32() && stats && !a_pro && bias && M >= 1024 && K >= 256 && K % 16 == 0 && Nout >= 128 && lda % 4 == 0 &&
         (((uintptr_t)A) & 15) == 0 && yolat_gemm_x6_work_elems(M, Nout, K) == 0;
}
// ops.linear_fwd_wt's
bool x6_wt(const float* A, long lda, long M, long K, long Nout, bool accumulate) {
  return !strict_fp32() && !accumulate && M >= 1024 && K >= 256 && K % 16 == 0 && Nout >= 512 && lda % 4 == 0 &&
         (((uintptr_t)A) & 15) == 0;
}

TrainBuf carve(const yolat_train_model* m, long N, long E, long P, void* ws) {
  Carve c{reinterpret_cast<char*>(ws), 0};
  TrainBuf b;
  memset(&b, 0, sizeof b);
  const long C = m->C, F = m->F, D = C * m->n_blocks_out, ZW = 2 * (F + D), L = m->n_blocks, lo = L - m->n_blocks_out;
  const long Ee = E > 0 ? E : 1, K = m->n_classes;
  b.row_ptr = c.take<int>(N + 1); b.perm = c.take<int>(Ee); b.src = c.take<int>(Ee); b.dst = c.take<int>(Ee);
  b.attr = c.take<float>(Ee * 4);
  const long n_seg = (P + 1 + 3) / 4 * 4;
  b.zblock = c.take<int>(n_seg + N);
  b.seg_ptr = b.zblock; b.node_seg = b.zblock ? b.zblock + n_seg : nullptr;
  b.gwork = c.take<int>(yolat_graph_work_elems(N, E));
  b.col_ptr = c.take<int>(N + 1); b.slots = c.take<int>(Ee); b.cwork = c.take<int>(yolat_csc_work_elems(N));
  b.inv_deg = c.take<float>(N);
  b.feats = c.take<float>(N * D); b.fsup = c.take<float>(N * D); b.sup_coef = c.take<float>(2 * D);
  for (long l = 0; l < L; ++l) {
    ConvBuf& v = b.cv[l];
    const long Cin = m->conv[l].Cin;
    v.fact_fwd = C == 64 && (double)E >= 2.0 * (double)N;
    v.half = m->half != 0 && v.fact_fwd;
    const size_t es = v.half ? 2 : 4;
    v.f_tmp = (l < lo) ? c.take<float>(N * C) : nullptr;
    v.s_tmp = (l < lo) ? c.take<float>(N * C) : nullptr;
    v.H1 = c.take<char>(Ee * C * es); v.H2 = c.take<char>(Ee * C * es);
    v.st1 = c.take<float>(yolat_bn_stats_elems(Ee, C)); v.st2 = c.take<float>(yolat_bn_stats_elems(Ee, C));
    v.c1 = c.take<float>(4 * C); v.c2 = c.take<float>(4 * C);
    v.wuv = c.take<float>(2 * C * Cin); v.wc4 = c.take<float>(C * 4); v.uv = c.take<float>(N * 2 * C);
    v.wwork_f = c.take<uint16_t>(C * C);
    v.st_n = c.take<float>(yolat_bn_stats_elems(N, C)); v.cn = c.take<float>(4 * C);
    v.dA1 = c.take<char>(Ee * C * es); v.coef2 = c.take<float>(2 * C); v.coef1 = c.take<float>(2 * C);
    v.w_stats = c.take<float>(yolat_bn_csr_work_elems(Ee, C)); v.w_l2 = c.take<float>(yolat_bn_csr_l2_bwd_work_elems());
    v.dUV = c.take<float>(N * 2 * C); v.dwc4 = c.take<float>(C * 4);
    v.w_apply = c.take<float>(yolat_bn_apply_edge_sums_work_elems(N));
    v.wuv_b = c.take<float>(2 * C * Cin); v.wc4_b = c.take<float>(C * 4);
    v.dwuv = c.take<float>(2 * C * Cin); v.w_dwuv = c.take<float>(yolat_linear_bwd_w_work_elems(N, 2 * C, Cin));
    v.w_root = c.take<float>(yolat_linear_bwd_w_work_elems(N, C, Cin));
    v.w_node_bn = c.take<float>(yolat_bn_bwd_work_elems(N, C));
    v.w_node_w = c.take<float>(yolat_linear_bwd_w_work_elems(N, C, Cin));
    // gradients flowing into a layer below the concat: one buffer each (d_f_next / d_s_next of engine.model_bwd)
    v.dx_tmp = (l > 0 && l - 1 < lo) ? c.take<float>(N * Cin) : nullptr;
    v.dxn_tmp = (l > 0 && l - 1 < lo) ? c.take<float>(N * Cin) : nullptr;
  }
  b.Z = c.take<float>(P * ZW); b.arg_feat = c.take<int>(P * D);
  b.fus_coef = c.take<float>(4 * F); b.fus_saved = c.take<float>(yolat_fusion_pool_train_saved_elems(D, F, P));
  b.fus_work = c.take<float>(yolat_fusion_pool_train_work_elems(N, D, F, P));
  b.fs_y = c.take<float>(P * F); b.fs_st = c.take<float>(yolat_bn_stats_elems(P, F)); b.fs_c = c.take<float>(4 * F);
  b.c1y = c.take<float>(P * m->H1); b.c1st = c.take<float>(yolat_bn_stats_elems(P, m->H1)); b.c1c = c.take<float>(4 * m->H1);
  b.c1pack = c.take<uint16_t>(yolat_gemm_x6_packed_elems(m->H1, ZW));
  b.c2y = c.take<float>(P * m->H2); b.c2st = c.take<float>(yolat_bn_stats_elems(P, m->H2)); b.c2c = c.take<float>(4 * m->H2);
  b.dl = c.take<float>(P * K); b.ce_work = c.take<float>(yolat_softmax_ce_work_elems(P));
  b.d2 = c.take<float>(P * m->H2); b.d1 = c.take<float>(P * m->H1); b.dZ = c.take<float>(P * ZW);
  b.d_fsup = c.take<float>(N * D); b.d_feats = c.take<float>(N * D);
  b.w3 = c.take<float>(yolat_linear_bwd_w_work_elems(P, K, m->H2));
  b.w2bn = c.take<float>(yolat_bn_bwd_work_elems(P, m->H2)); b.w2w = c.take<float>(yolat_linear_bwd_w_work_elems(P, m->H2, m->H1));
  b.p2 = c.take<uint16_t>(yolat_gemm_x6_packed_elems(m->H1, m->H2));
  b.x2w = c.take<float>(yolat_gemm_x6_work_elems(P, m->H1, m->H2) + 1);
  b.w1bn = c.take<float>(yolat_bn_bwd_work_elems(P, m->H1)); b.w1w = c.take<float>(yolat_linear_bwd_w_work_elems(P, m->H1, ZW));
  b.p1 = c.take<uint16_t>(yolat_gemm_x6_packed_elems(ZW, m->H1));
  b.x1w = c.take<float>(yolat_gemm_x6_work_elems(P, ZW, m->H1) + 1);
  b.wfsbn = c.take<float>(yolat_bn_bwd_work_elems(P, F)); b.wfsw = c.take<float>(yolat_linear_bwd_w_work_elems(P, F, D));
  b.bytes = c.off + 256;
  return b;
}

int model_ok(const yolat_train_model* m) {
  if (!m || !m->param_base || !m->grad_base) return YOLAT_E_INVALID;
  if (m->n_blocks < 1 || m->n_blocks > TP_MAXL || m->n_blocks_out < 1 || m->n_blocks_out > m->n_blocks || m->n_classes < 1)
    return YOLAT_E_INVALID;
  if (m->C != 64 || m->n_blocks_out != 2 || m->F <= 0 || m->F % 4 != 0 || m->H1 <= 0 || m->H2 <= 0) return YOLAT_E_UNSUPPORTED;
  auto lin_ok = [](const yolat_train_lin& l) { return l.W != nullptr && l.b != nullptr; };
  auto bn_ok = [](const yolat_train_bn& b) { return b.gamma != nullptr && b.beta != nullptr; };
  for (int l = 0; l < m->n_blocks; ++l) {
    const yolat_train_conv& cv = m->conv[l];
    if (cv.Cin < 1 || (l > 0 && cv.Cin != 64)) return YOLAT_E_UNSUPPORTED;
    if (!lin_ok(cv.nn0) || !bn_ok(cv.bn1) || !lin_ok(cv.nn3) || !bn_ok(cv.bn4) || !lin_ok(cv.lin_r) || !lin_ok(cv.node) ||
        !bn_ok(cv.bn_node))
      return YOLAT_E_UNSUPPORTED;
  }
  if (!lin_ok(m->fus) || !bn_ok(m->fus_bn) || !lin_ok(m->fus_s) || !bn_ok(m->fus_s_bn) || !lin_ok(m->c1) || !bn_ok(m->c1_bn) ||
      !lin_ok(m->c2) || !bn_ok(m->c2_bn) || !lin_ok(m->c3))
    return YOLAT_E_UNSUPPORTED;
  return 0;
}

__global__ void k_tp_zero(int* p, long n, int* status) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i < n) p[i] = 0;
  if (i == 0 && status) *status = 0;
}
struct NbtList { long long* p[4 * TP_MAXL + 8]; int n; };
__global__ void k_tp_nbt(NbtList l) {
  if ((int)threadIdx.x < l.n && l.p[threadIdx.x] != nullptr) *l.p[threadIdx.x] += 1;
}

// two streams, forked / joined through events (engine._on_side / _join_side)
struct Streams {
  hipStream_t main, side;
  hipEvent_t ev_fork, ev_join, ev_mark;
  bool dirty;
  void mark() {                        // a point of the side stream the main stream can wait for WITHOUT joining everything
    if (side) (void)hipEventRecord(ev_mark, side);
  }
  void wait_mark() {
    if (side) (void)hipStreamWaitEvent(main, ev_mark, 0);
  }
  hipStream_t fork() {                 // the side stream, ordered behind everything issued on main so far
    if (!side) return main;
    (void)hipEventRecord(ev_fork, main);
    (void)hipStreamWaitEvent(side, ev_fork, 0);
    dirty = true;
    return side;
  }
  void join() {                        // main waits for the side-stream work issued so far
    if (!side || !dirty) return;
    (void)hipEventRecord(ev_join, side);
    (void)hipStreamWaitEvent(main, ev_join, 0);
    dirty = false;
  }
};

// the two events of a (main, side) pair, created once per process and pair of streams
struct EvCache { hipStream_t m, s; hipEvent_t f, j, k; };
EvCache g_ev[8];
int g_nev = 0;
std::mutex g_ev_mu;
bool events_for(hipStream_t m, hipStream_t s, hipEvent_t* f, hipEvent_t* j, hipEvent_t* k) {
  std::lock_guard<std::mutex> lock(g_ev_mu);       // (trainers on several devices / host threads share the table)
  for (int i = 0; i < g_nev; ++i)
    if (g_ev[i].m == m && g_ev[i].s == s) { *f = g_ev[i].f; *j = g_ev[i].j; *k = g_ev[i].k; return true; }
  EvCache e{m, s, nullptr, nullptr, nullptr};
  if (hipEventCreateWithFlags(&e.f, hipEventDisableTiming) != hipSuccess) return false;
  if (hipEventCreateWithFlags(&e.j, hipEventDisableTiming) != hipSuccess) return false;
  if (hipEventCreateWithFlags(&e.k, hipEventDisableTiming) != hipSuccess) return false;
  const int slot = g_nev < 8 ? g_nev++ : 7;      // (a process uses one or two pairs; the last slot is recycled beyond eight)
  if (slot == 7 && g_nev == 8 && g_ev[7].f) {
    (void)hipEventDestroy(g_ev[7].f); (void)hipEventDestroy(g_ev[7].j); (void)hipEventDestroy(g_ev[7].k);
  }
  g_ev[slot] = e;
  *f = e.f; *j = e.j; *k = e.k;
  return true;
}

#define TP_TRY(call)            \
  do {                          \
    int rc__ = (call);          \
    if (rc__ != 0) return rc__; \
  } while (0)

inline float* grad_of(const yolat_train_model* m, const float* p) {
  return p ? m->grad_base + (p - m->param_base) : nullptr;
}

// ops.linear_fwd (fp32 operands).  pack != NULL: the bf16x6 path may be taken; prepacked: its weight image was produced
// earlier in this step (the side stream's weight-only preparation) — the same kernel on the same weights, issued sooner
int lin_fwd(const float* A, long lda, long M, long K, const float* asc, const float* ash, int arelu, const float* W,
            const float* bias, long Nout, float* Y, long ldy, float* stats, uint16_t* pack, bool prepacked, hipStream_t st) {
  if (pack && x6_fwd(A, lda, M, K, Nout, asc != nullptr, stats != nullptr, bias != nullptr)) {
    if (!prepacked) TP_TRY(yolat_gemm_x6_pack(W, K, Nout, K, nullptr, pack, st));
    return yolat_gemm_x6_stats(A, lda, M, K, pack, bias, Nout, Y, ldy, stats, st);
  }
  return yolat_linear_fwd(A, lda, M, K, asc, ash, arelu, W, K, bias, Nout, nullptr, nullptr, 0, Y, ldy, 0, stats, st);
}

// ops.linear_fwd_wt (fp32): Y (+)= A . Wt, Wt [K, Nout] row-major
int lin_wt(const float* A, long lda, long M, long K, const float* Wt, long Nout, float* Y, long ldy, int accumulate,
           uint16_t* pack, bool prepacked, float* work, hipStream_t st) {
  if (pack && x6_wt(A, lda, M, K, Nout, accumulate != 0)) {
    if (!prepacked) TP_TRY(yolat_gemm_x6_pack_t(Wt, Nout, Nout, K, pack, st));
    return yolat_gemm_x6(A, lda, M, K, pack, nullptr, 0, Nout, Y, ldy, work, st);
  }
  return yolat_linear_fwd_wt(A, lda, M, K, Wt, Nout, Nout, Y, ldy, accumulate, st);
}

struct Lazy { const float* t; long ld; const float* scale; const float* shift; int relu; };

}  // namespace

extern "C" size_t yolat_train_step_workspace_bytes(const yolat_train_model* m, int64_t N, int64_t E, int64_t P) {
  if (model_ok(m) != 0 || N <= 0 || E < 0 || P <= 0) return 0;
  return carve(m, N, E, P, nullptr).bytes;
}

extern "C" int yolat_train_step(const yolat_train_model* m, const float* x, int64_t ldx, const int64_t* edge,
                                int64_t stride_e, int64_t stride_c, const float* e_attr, const int64_t* bbox_idx,
                                const yolat_graph_csr* g, const int64_t* labels, int64_t N, int64_t E, int64_t P,
                                float* logits, int64_t ld_logits, float* loss, void* workspace, size_t workspace_bytes,
                                int32_t* status, const yolat_adam_args* adam, int phases, yolat_stream_t stream,
                                yolat_stream_t side_stream) {
  TP_TRY(model_ok(m));
  if (!x || !labels || !logits || !loss || !workspace || !status || N <= 0 || E <= 0 || P <= 0 || (phases & 7) == 0)
    return YOLAT_E_INVALID;
  if (!g && (!edge || !e_attr || !bbox_idx)) return YOLAT_E_INVALID;
  if ((phases & 4) && (!adam || !adam->exp_avg || !adam->exp_avg_sq || adam->n <= 0 || adam->step < 1)) return YOLAT_E_INVALID;
  if (E < N || N >= (1LL << 30) || E >= (1LL << 30)) return YOLAT_E_UNSUPPORTED;   // (E >= N: the factorised backward)
  const long C = m->C, F = m->F, D = C * m->n_blocks_out, ZW = 2 * (F + D), L = m->n_blocks, lo = L - m->n_blocks_out;
  const long K = m->n_classes, H1 = m->H1, H2 = m->H2;
  if (ld_logits < K) return YOLAT_E_INVALID;
  TrainBuf b = carve(m, N, E, P, workspace);
  if (b.bytes > workspace_bytes || (((uintptr_t)workspace) & 255) != 0) return YOLAT_E_INVALID;
  // the fused backward kernels read parameter / coefficient vectors with 16-byte loads (engine.conv_bwd's `aligned` gate)
  for (long l = 0; l < L; ++l)
    if ((((uintptr_t)m->conv[l].nn3.W) & 15) != 0) return YOLAT_E_UNSUPPORTED;
  if ((((uintptr_t)m->fus.W) & 15) != 0) return YOLAT_E_UNSUPPORTED;

  Streams S;
  S.main = (hipStream_t)stream;
  S.side = (side_stream && side_stream != stream) ? (hipStream_t)side_stream : nullptr;
  S.dirty = false;
  S.ev_fork = S.ev_join = S.ev_mark = nullptr;
  if (S.side && !events_for(S.main, S.side, &S.ev_fork, &S.ev_join, &S.ev_mark)) return YOLAT_E_INVALID;
  hipStream_t st = S.main;

  const int* row_ptr = b.row_ptr; const int* src = b.src; const int* dst = b.dst; const float* attr = b.attr;
  const int* seg_ptr = b.seg_ptr; const int* node_seg = b.node_seg;
  if (g) {
    if (!g->row_ptr || !g->src || !g->dst || !g->attr || !g->seg_ptr || !g->node_seg) return YOLAT_E_INVALID;
    row_ptr = g->row_ptr; src = g->src; dst = g->dst; attr = g->attr; seg_ptr = g->seg_ptr; node_seg = g->node_seg;
  }
  if ((((uintptr_t)attr) & 15) != 0) return YOLAT_E_UNSUPPORTED;

  auto f_out = [&](long l) { return l - lo >= 0 ? b.feats + (l - lo) * C : b.cv[l].f_tmp; };
  auto s_out = [&](long l) { return l - lo >= 0 ? b.fsup + (l - lo) * C : b.cv[l].s_tmp; };
  auto ld_out = [&](long l) { return l - lo >= 0 ? D : C; };
  auto finalize = [&](const float* stats, long M, long Cc, const yolat_train_bn& bn, float* scale, float* shift, float* mean,
                      float* invstd, hipStream_t s) {
    return yolat_bn_finalize(stats, M, Cc, bn.gamma, bn.beta, bn.running_mean, bn.running_var, bn.momentum, bn.eps, mean, invstd,
                             scale, shift, s);
  };

  if (phases & 1) {
    // ================================ graph ================================
    if (!g) {
      const long n_seg = (P + 1 + 3) / 4 * 4;
      hipLaunchKernelGGL(k_tp_zero, dim3(yl_cdiv(n_seg + N, 256)), dim3(256), 0, st, b.zblock, n_seg + N, status);
      YL_LAUNCH_CHECK();
      TP_TRY(yolat_graph_prepare(edge, stride_e, stride_c, e_attr, bbox_idx, E, N, P, b.row_ptr, b.perm, b.src, b.dst, b.attr,
                                 b.seg_ptr, b.node_seg, b.gwork, status, st));
    }
    // ---- work that depends on the weights or the graph only, off the critical path: the side stream takes it now, the
    // forward's join (before the per-proposal mean) is long past it.  (The Python schedule issues the same launches where
    // their results are first needed, on the main stream; same kernels, same operands.)
    {
      hipStream_t ss = S.fork();
      if (x6_fwd(b.Z, ZW, P, ZW, H1, false, true, true)) TP_TRY(yolat_gemm_x6_pack(m->c1.W, ZW, H1, ZW, nullptr, b.c1pack, ss));
      if (x6_wt(b.d2, H2, P, H2, H1, false)) TP_TRY(yolat_gemm_x6_pack_t(m->c2.W, H1, H1, H2, b.p2, ss));
      if (x6_wt(b.d1, H1, P, H1, ZW, false)) TP_TRY(yolat_gemm_x6_pack_t(m->c1.W, ZW, ZW, H1, b.p1, ss));
      for (long l = 0; l < L; ++l)
        if (!b.cv[l].fact_fwd) TP_TRY(yolat_conv_split_w1(m->conv[l].nn0.W, m->conv[l].Cin, C, b.cv[l].wuv_b, b.cv[l].wc4_b, ss));
      // CSC by source + 1 / deg for the backward (ops.Graph.ensure_csc / inv_deg)
      TP_TRY(yolat_inv_degree(row_ptr, N, b.inv_deg, ss));
      TP_TRY(yolat_csc_by_source(src, E, N, b.col_ptr, b.slots, b.cwork, ss));
    }
    // ================================ forward ================================
    Lazy s{x, ldx, nullptr, nullptr, 0};
    const float* f = x;
    long ldf = ldx;
    for (long l = 0; l < L; ++l) {
      const yolat_train_conv& cv = m->conv[l];
      ConvBuf& v = b.cv[l];
      const long Cin = cv.Cin;
      float* of = f_out(l);
      float* os = s_out(l);
      const long ldo = ld_out(l);
      // root term first, the aggregation accumulates onto it (torch_vertex.py:325)
      TP_TRY(yolat_linear_fwd(f, ldf, N, Cin, nullptr, nullptr, 0, cv.lin_r.W, Cin, cv.lin_r.b, C, nullptr, nullptr, 0, of, ldo, 0,
                              nullptr, st));
      if (v.fact_fwd) {
        TP_TRY(yolat_conv_split_w1(cv.nn0.W, Cin, C, v.wuv, v.wc4, st));
        TP_TRY(yolat_linear_fwd(f, ldf, N, Cin, nullptr, nullptr, 0, v.wuv, Cin, nullptr, 2 * C, nullptr, nullptr, 0, v.uv,
                                2 * C, 0, nullptr, st));
        if (v.half)
          TP_TRY(yolat_edge_uv_lin1_fwd_h(v.uv, 2 * C, src, dst, attr, E, v.wc4, cv.nn0.b, C, (uint16_t*)v.H1, C, v.st1, st));
        else
          TP_TRY(yolat_edge_uv_lin1_fwd(v.uv, 2 * C, src, dst, attr, E, v.wc4, cv.nn0.b, C, (float*)v.H1, C, v.st1, st));
      } else {
        TP_TRY(yolat_edge_lin1_fwd(f, ldf, N, Cin, src, dst, attr, E, cv.nn0.W, 2 * Cin + 4, cv.nn0.b, C, nullptr, nullptr, 0,
                                   (float*)v.H1, C, v.st1, st));
      }
      TP_TRY(finalize(v.st1, E, C, cv.bn1, v.c1, v.c1 + C, v.c1 + 2 * C, v.c1 + 3 * C, st));
      if (v.half)
        TP_TRY(yolat_linear_fwd_h((const uint16_t*)v.H1, C, E, C, v.c1, v.c1 + C, 1, cv.nn3.W, C, cv.nn3.b, C, (uint16_t*)v.H2, C,
                                  v.st2, v.wwork_f, st));
      else
        TP_TRY(yolat_linear_fwd((const float*)v.H1, C, E, C, v.c1, v.c1 + C, 1, cv.nn3.W, C, cv.nn3.b, C, nullptr, nullptr, 0,
                                (float*)v.H2, C, 0, v.st2, st));
      TP_TRY(finalize(v.st2, E, C, cv.bn4, v.c2, v.c2 + C, v.c2 + 2 * C, v.c2 + 3 * C, st));
      if (v.half)
        TP_TRY(yolat_csr_mean_fwd_h((const uint16_t*)v.H2, C, C, v.c2, v.c2 + C, 1, row_ptr, N, of, ldo, 1, st));
      else
        TP_TRY(yolat_csr_mean_fwd((const float*)v.H2, C, C, v.c2, v.c2 + C, 1, row_ptr, N, of, ldo, 1, st));
      // node branch (mlp_node) on the side stream: read again only by the next layer's node branch and the per-proposal mean
      {
        hipStream_t ss = S.fork();
        v.cn_scale = (l - lo >= 0) ? b.sup_coef + (l - lo) * C : v.cn;
        v.cn_shift = (l - lo >= 0) ? b.sup_coef + D + (l - lo) * C : v.cn + C;
        TP_TRY(yolat_linear_fwd(s.t, s.ld, N, Cin, s.scale, s.shift, s.relu, cv.node.W, Cin, cv.node.b, C, nullptr, nullptr, 0, os,
                                ldo, 0, v.st_n, ss));
        TP_TRY(finalize(v.st_n, N, C, cv.bn_node, v.cn_scale, v.cn_shift, v.cn + 2 * C, v.cn + 3 * C, ss));
      }
      f = of; ldf = ldo;
      s = Lazy{os, ldo, v.cn_scale, v.cn_shift, 1};
    }
    // Everything between the conv layers and the classifier that does NOT go through the fusion block — the per-proposal max
    // of feats, the per-proposal mean of the node branches (computed on the side stream anyway) and fusion_block_super on its
    // P rows — runs on the side stream BESIDE the fusion block (428 us at cfg 3; disjoint column ranges of Z), joined in
    // front of the classifier.  (The Python schedule issues them on the main stream behind it; same kernels and operands.)
    {
      hipStream_t ss = S.fork();       // (behind the last conv layer's aggregation: feats is complete)
      float* sup = b.Z + 2 * F + D;
      TP_TRY(yolat_segment_max_fwd(b.feats, D, D, nullptr, nullptr, 0, seg_ptr, P, N, b.Z + F, ZW, b.arg_feat, ss));
      TP_TRY(yolat_segment_mean_fwd(b.fsup, D, D, b.sup_coef, b.sup_coef + D, 1, seg_ptr, P, sup, ZW, ss));
      TP_TRY(lin_fwd(sup, ZW, P, D, nullptr, nullptr, 0, m->fus_s.W, m->fus_s.b, F, b.fs_y, F, b.fs_st, nullptr, false, ss));
      TP_TRY(finalize(b.fs_st, P, F, m->fus_s_bn, b.fs_c, b.fs_c + F, b.fs_c + 2 * F, b.fs_c + 3 * F, ss));
      TP_TRY(yolat_scale_shift_relu(b.fs_y, F, P, F, b.fs_c, b.fs_c + F, 1, b.Z + F + D, ZW, ss));
    }
    // fusion block over nodes + per-proposal max (arch:61-63,122): fused, no [N, F] activation
    TP_TRY(yolat_fusion_pool_train_fwd(b.feats, D, N, D, m->fus.W, m->fus.b, F, m->fus_bn.gamma, m->fus_bn.beta,
                                       m->fus_bn.running_mean, m->fus_bn.running_var, m->fus_bn.momentum, m->fus_bn.eps, node_seg,
                                       P, b.Z, ZW, b.fus_coef, b.fus_saved, b.fus_work, st));
    S.join();                          // Z is complete: node branches, pooled rows, fusion_block_super, the weight packs
    // classifier (arch:91-93,128)
    TP_TRY(lin_fwd(b.Z, ZW, P, ZW, nullptr, nullptr, 0, m->c1.W, m->c1.b, H1, b.c1y, H1, b.c1st, b.c1pack, true, st));
    TP_TRY(finalize(b.c1st, P, H1, m->c1_bn, b.c1c, b.c1c + H1, b.c1c + 2 * H1, b.c1c + 3 * H1, st));
    TP_TRY(lin_fwd(b.c1y, H1, P, H1, b.c1c, b.c1c + H1, 1, m->c2.W, m->c2.b, H2, b.c2y, H2, b.c2st, nullptr, false, st));
    TP_TRY(finalize(b.c2st, P, H2, m->c2_bn, b.c2c, b.c2c + H2, b.c2c + 2 * H2, b.c2c + 3 * H2, st));
    TP_TRY(yolat_linear_fwd(b.c2y, H2, P, H2, b.c2c, b.c2c + H2, 1, m->c3.W, H2, m->c3.b, K, nullptr, nullptr, 0, logits, ld_logits,
                            0, nullptr, st));
    {   // BatchNorm1d.num_batches_tracked += 1 for every layer of the forward, one launch
      NbtList nl;
      nl.n = 0;
      auto add = [&](const yolat_train_bn& bn) { if (bn.num_batches_tracked) nl.p[nl.n++] = (long long*)bn.num_batches_tracked; };
      for (long l = 0; l < L; ++l) { add(m->conv[l].bn1); add(m->conv[l].bn4); add(m->conv[l].bn_node); }
      add(m->fus_bn); add(m->fus_s_bn); add(m->c1_bn); add(m->c2_bn);
      if (nl.n > 0) {
        hipLaunchKernelGGL(k_tp_nbt, dim3(1), dim3(64), 0, st, nl);
        YL_LAUNCH_CHECK();
      }
    }
    // ================================ loss ================================
    TP_TRY(yolat_softmax_ce(logits, ld_logits, labels, P, K, loss, b.dl, K, b.ce_work, st));

    // ================================ backward: classifier, fusion blocks ================================
    // prediction_cls.2 (Linear only)
    TP_TRY(yolat_linear_bwd_w(b.dl, K, P, K, b.c2y, H2, H2, b.c2c, b.c2c + H2, 1, grad_of(m, m->c3.W), H2, grad_of(m, m->c3.b), 0,
                              b.w3, S.fork()));
    TP_TRY(yolat_linear_fwd_wt(b.dl, K, P, K, m->c3.W, H2, H2, b.d2, H2, 0, st));
    // prediction_cls.1
    TP_TRY(yolat_bn_relu_bwd(b.d2, H2, b.c2y, H2, P, H2, m->c2_bn.gamma, b.c2c + 2 * H2, b.c2c + 3 * H2, b.c2c, b.c2c + H2, 1,
                             grad_of(m, m->c2_bn.gamma), grad_of(m, m->c2_bn.beta), 0, b.d2, H2, b.w2bn, st));
    TP_TRY(yolat_linear_bwd_w(b.d2, H2, P, H2, b.c1y, H1, H1, b.c1c, b.c1c + H1, 1, grad_of(m, m->c2.W), H1, grad_of(m, m->c2.b), 0,
                              b.w2w, S.fork()));
    TP_TRY(lin_wt(b.d2, H2, P, H2, m->c2.W, H1, b.d1, H1, 0, b.p2, true, b.x2w, st));
    // prediction_cls.0
    TP_TRY(yolat_bn_relu_bwd(b.d1, H1, b.c1y, H1, P, H1, m->c1_bn.gamma, b.c1c + 2 * H1, b.c1c + 3 * H1, b.c1c, b.c1c + H1, 1,
                             grad_of(m, m->c1_bn.gamma), grad_of(m, m->c1_bn.beta), 0, b.d1, H1, b.w1bn, st));
    TP_TRY(yolat_linear_bwd_w(b.d1, H1, P, H1, b.Z, ZW, ZW, nullptr, nullptr, 0, grad_of(m, m->c1.W), ZW, grad_of(m, m->c1.b), 0,
                              b.w1w, S.fork()));
    TP_TRY(lin_wt(b.d1, H1, P, H1, m->c1.W, ZW, b.dZ, ZW, 0, b.p1, true, b.x1w, st));
    // fusion_block_super: input sup = Z[:, 2F+D:], post-activation output Z[:, F+D:2F+D].  Its whole backward and the
    // per-proposal mean's feed nothing but the node branches' backward chain, which lives on the side stream: so do they,
    // beside the fusion block's backward (their columns of dZ are disjoint from the ones the main stream reads)
    {
      hipStream_t ss = S.fork();
      float* d_sup = b.dZ + 2 * F + D;
      float* dz_fs = b.dZ + F + D;
      TP_TRY(yolat_bn_relu_bwd(dz_fs, ZW, b.fs_y, F, P, F, m->fus_s_bn.gamma, b.fs_c + 2 * F, b.fs_c + 3 * F, b.fs_c, b.fs_c + F, 1,
                               grad_of(m, m->fus_s_bn.gamma), grad_of(m, m->fus_s_bn.beta), 0, dz_fs, ZW, b.wfsbn, ss));
      TP_TRY(yolat_linear_bwd_w(dz_fs, ZW, P, F, b.Z + 2 * F + D, ZW, D, nullptr, nullptr, 0, grad_of(m, m->fus_s.W), D,
                                grad_of(m, m->fus_s.b), 0, b.wfsw, ss));
      TP_TRY(yolat_linear_fwd_wt(dz_fs, ZW, P, F, m->fus_s.W, D, D, d_sup, ZW, 1, ss));
      TP_TRY(yolat_segment_mean_bwd(d_sup, ZW, D, seg_ptr, node_seg, N, b.d_fsup, D, ss));
    }
    // fusion_block + max pooling
    TP_TRY(yolat_segment_max_bwd(b.dZ + F, ZW, D, b.arg_feat, node_seg, N, b.d_feats, D, st));
    auto fus_part = [&](int mask, hipStream_t s) {
      return yolat_fusion_pool_train_bwd_parts(b.feats, D, N, D, m->fus.W, m->fus_bn.gamma, F, b.fus_coef, b.fus_saved, node_seg,
                                               seg_ptr, P, b.dZ, ZW, grad_of(m, m->fus.W), grad_of(m, m->fus.b),
                                               grad_of(m, m->fus_bn.gamma), grad_of(m, m->fus_bn.beta), b.d_feats, D, b.fus_work,
                                               mask, s);
    };
    // (column reductions both halves read, then the weight gradient on the side stream beside the input gradient)
    TP_TRY(fus_part(1, st));
    TP_TRY(fus_part(2, S.fork()));
    TP_TRY(fus_part(4, st));
    S.join();                          // the head bucket's gradients are complete on `stream`
  }

  if (phases & 2) {
    // ================================ backward: conv layers, last to first ================================
    // (CSC by source, 1 / deg and the weight splits were prepared by phase 1 on the side stream, joined at its end)
    float* d_f_next = nullptr;
    float* d_s_next = nullptr;
    for (long l = L - 1; l >= 0; --l) {
      const yolat_train_conv& cv = m->conv[l];
      ConvBuf& v = b.cv[l];
      const long Cin = cv.Cin, slot = l - lo;
      float* d_f = slot >= 0 ? b.d_feats + slot * C : d_f_next;
      float* d_s = slot >= 0 ? b.d_fsup + slot * C : d_s_next;
      const long ldd = slot >= 0 ? D : C;
      const bool need_dx = l > 0;
      float* dx = nullptr;
      float* dxn = nullptr;
      long lddx = Cin;
      int acc = 0;
      if (need_dx) {
        const long pslot = l - 1 - lo;
        if (pslot >= 0) { dx = b.d_feats + pslot * C; dxn = b.d_fsup + pslot * C; lddx = D; acc = 1; }
        else { dx = v.dx_tmp; dxn = v.dxn_tmp; lddx = Cin; acc = 0; }
      }
      // the layer's inputs as the forward saw them
      const float* xin = l == 0 ? x : f_out(l - 1);
      const long ldxin = l == 0 ? ldx : ld_out(l - 1);
      Lazy xn = l == 0 ? Lazy{x, ldx, nullptr, nullptr, 0}
                       : Lazy{s_out(l - 1), ld_out(l - 1), b.cv[l - 1].cn_scale, b.cv[l - 1].cn_shift, 1};
      if (l > 0 && !(phases & 1)) {    // (phase 2 in a call of its own: the forward's coefficient slots, recomputed)
        const long ps = l - 1 - lo;
        xn.scale = ps >= 0 ? b.sup_coef + ps * C : b.cv[l - 1].cn;
        xn.shift = ps >= 0 ? b.sup_coef + D + ps * C : b.cv[l - 1].cn + C;
      }
      const float* cn_scale = slot >= 0 ? b.sup_coef + slot * C : v.cn;
      const float* cn_shift = slot >= 0 ? b.sup_coef + D + slot * C : v.cn + C;
      // node branch: a chain of its own through the layers -> side stream
      {
        hipStream_t ss = S.fork();
        TP_TRY(yolat_bn_relu_bwd(d_s, ldd, s_out(l), ld_out(l), N, C, cv.bn_node.gamma, v.cn + 2 * C, v.cn + 3 * C, cn_scale,
                                 cn_shift, 1, grad_of(m, cv.bn_node.gamma), grad_of(m, cv.bn_node.beta), 0, d_s, ldd,
                                 v.w_node_bn, ss));
        TP_TRY(yolat_linear_bwd_w(d_s, ldd, N, C, xn.t, xn.ld, Cin, xn.scale, xn.shift, xn.relu, grad_of(m, cv.node.W), Cin,
                                  grad_of(m, cv.node.b), 0, v.w_node_w, ss));
        if (need_dx) TP_TRY(yolat_linear_fwd_wt(d_s, ldd, N, C, cv.node.W, Cin, Cin, dxn, lddx, acc, ss));
      }
      // root term
      TP_TRY(yolat_linear_bwd_w(d_f, ldd, N, C, xin, ldxin, Cin, nullptr, nullptr, 0, grad_of(m, cv.lin_r.W), Cin,
                                grad_of(m, cv.lin_r.b), 0, v.w_root, S.fork()));
      if (need_dx) TP_TRY(yolat_linear_fwd_wt(d_f, ldd, N, C, cv.lin_r.W, Cin, Cin, dx, lddx, acc, st));
      // edge side: the gradient w.r.t. H2 (mean -> ReLU -> BatchNorm backward) is formed inside its consumers (bn_csr.hip)
      yolat_bn_csr_grad dg;
      dg.d_out = d_f; dg.ld_out = ldd; dg.dst = dst; dg.inv_deg = b.inv_deg; dg.Y = v.H2; dg.ldy = C;
      dg.mean = v.c2 + 2 * C; dg.invstd = v.c2 + 3 * C; dg.scale = v.c2; dg.shift = v.c2 + C; dg.coef = v.coef2; dg.relu = 1;
      dg.half = v.half ? 1 : 0;
      TP_TRY(yolat_bn_csr_bwd_stats(&dg, E, C, grad_of(m, cv.bn4.gamma), grad_of(m, cv.bn4.beta), 0, v.coef2, v.w_stats, st));
      TP_TRY(yolat_bn_csr_l2_bwd(&dg, E, v.H1, C, v.c1, v.c1 + C, 1, cv.nn3.W, C, grad_of(m, cv.nn3.W), C, grad_of(m, cv.nn3.b), 0,
                                 v.dA1, C, v.w_l2, v.c1 + 2 * C, v.c1 + 3 * C, grad_of(m, cv.bn1.gamma), grad_of(m, cv.bn1.beta),
                                 v.coef1, st));
      // BatchNorm-1 backward apply + per-node dU sums + attr weight gradient + db1 in one pass
      TP_TRY(yolat_bn_apply_edge_sums(v.dA1, C, v.H1, C, v.dA1, C, v.half ? 1 : 0, E, v.c1 + 2 * C, v.c1 + 3 * C, v.c1, v.c1 + C, 1,
                                      v.coef1, row_ptr, attr, N, v.dUV, 2 * C, v.dwc4, grad_of(m, cv.nn0.b), v.w_apply, st));
      // first edge Linear through the per-node products (ops.edge_lin1_bwd_factorised with partial = (dUV, dWc4))
      const float* wuv = v.wuv;
      if (!v.fact_fwd) wuv = v.wuv_b;
      TP_TRY(yolat_edge_uv_sums_v(v.dA1, C, v.half ? 1 : 0, b.col_ptr, b.slots, N, C, v.dUV, 2 * C, st));
      {
        hipStream_t ss = S.fork();
        TP_TRY(yolat_linear_bwd_w(v.dUV, 2 * C, N, 2 * C, xin, ldxin, Cin, nullptr, nullptr, 0, v.dwuv, Cin, nullptr, 0, v.w_dwuv,
                                  ss));
        TP_TRY(yolat_conv_merge_dw1(v.dwuv, v.dwc4, Cin, C, grad_of(m, cv.nn0.W), 2 * Cin + 4, 0, ss));
      }
      if (need_dx) TP_TRY(yolat_linear_fwd_wt(v.dUV, 2 * C, N, 2 * C, wuv, Cin, Cin, dx, lddx, 1, st));
      d_f_next = dx;
      d_s_next = dxn;
    }
    S.join();                          // every gradient is complete on `stream`
  }

  if (phases & 4) {
    TP_TRY(yolat_adam_step(const_cast<float*>(m->param_base), m->grad_base, adam->exp_avg, adam->exp_avg_sq, adam->n, adam->lr,
                           adam->beta1, adam->beta2, adam->eps, adam->weight_decay, adam->step, adam->grad_scale, st));
  }
  return 0;
}
