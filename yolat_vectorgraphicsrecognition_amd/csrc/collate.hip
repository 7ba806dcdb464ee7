// collate.hip — HOST side of the batch hand-over: collate + index fix-up + the destination-sorted
// edge structure as native code behind the C ABI, feeding ONE pinned staging buffer and ONE host -> device copy.
//
// Reference: cad_recognition/train.py:123-171 (collate: torch.cat per key + slices), :238-258 (edge / bbox_idx offset
// fix-up loops) and the six synchronous .cuda() copies of architecture3cc_rpn_gp_iter2.py:107-115.
//
//   yolat_collate_pack      memcpy of every item's arrays, field by field, into the staging buffer (one call; ctypes
//                           releases the GIL around it)
//   yolat_item_csr_host     the CSR form of ONE dataset item — what yolat_graph_prepare builds on the device for a batch
//                           (stable counting sort of the edges by destination, e_attr permuted, proposal segment
//                           pointers; same clamping, same status flags).  A dataset item's graph never changes, so this
//                           runs once per item and is cached with it, the way the reference caches its proposals
//                           (Datasets/graph_dict3.py:924-929).
//   yolat_collate_csr_pack  the batch's CSR from the items' CSRs: every edge joins two nodes of the same item
//                           (graph_dict3.py:594-600), so the batch adjacency is block diagonal and its destination-sorted
//                           form is the CONCATENATION of the items' forms with the node / edge / proposal offsets added
//                           — bit-identical to rebuilding it from the collated COO list (tests/test_abi_host.py).
// No device code in this file.
#include "common.hpp"
#include <string.h>
#include <vector>

extern "C" int yolat_collate_pack(void* dst, const int64_t* field_off, const yolat_span* spans, int64_t n_fields,
                                  int64_t n_items) {
  if (!dst || !field_off || !spans || n_fields < 0 || n_items < 0) return YOLAT_E_INVALID;
  char* base = reinterpret_cast<char*>(dst);
  for (int64_t f = 0; f < n_fields; ++f) {
    char* o = base + field_off[f];
    for (int64_t i = 0; i < n_items; ++i) {
      const yolat_span& s = spans[f * n_items + i];
      if (s.bytes < 0 || (s.bytes > 0 && !s.ptr)) return YOLAT_E_INVALID;
      if (s.bytes) memcpy(o, s.ptr, (size_t)s.bytes);
      o += s.bytes;
    }
  }
  return 0;
}

extern "C" int yolat_item_csr_host(const int64_t* edge, int64_t stride_e, int64_t stride_c, const float* e_attr,
                                   const int64_t* bbox_idx, int64_t E, int64_t N, int64_t P, int32_t* row_ptr,
                                   int32_t* perm, int32_t* src, int32_t* dst, float* attr, int32_t* seg_ptr,
                                   int32_t* node_seg, int32_t* status) {
  if (N <= 0 || E < 0 || !row_ptr || !status || (E > 0 && (!edge || !e_attr || !perm || !src || !dst || !attr)))
    return YOLAT_E_INVALID;
  if (N >= (1LL << 31) || E >= (1LL << 31)) return YOLAT_E_UNSUPPORTED;
  int st = 0;
  // destination counts (ids outside [0, N) are clamped and flagged, as k_prep_count does)
  std::vector<int32_t> s32((size_t)E), d32((size_t)E);
  for (int64_t n = 0; n <= N; ++n) row_ptr[n] = 0;
  for (int64_t e = 0; e < E; ++e) {
    int64_t s = edge[e * stride_e], d = edge[e * stride_e + stride_c];
    if (s < 0 || s >= N || d < 0 || d >= N) {
      st |= YOLAT_STATUS_EDGE_RANGE;
      s = s < 0 ? 0 : (s >= N ? N - 1 : s);
      d = d < 0 ? 0 : (d >= N ? N - 1 : d);
    }
    s32[(size_t)e] = (int32_t)s;
    d32[(size_t)e] = (int32_t)d;
    ++row_ptr[d + 1];
  }
  for (int64_t n = 0; n < N; ++n) row_ptr[n + 1] += row_ptr[n];
  // stable placement: ascending edge id inside a destination row (the summation order of torch_scatter's CPU path)
  std::vector<int32_t> fill(row_ptr, row_ptr + N);
  for (int64_t e = 0; e < E; ++e) {
    const int32_t pos = fill[(size_t)d32[(size_t)e]]++;
    perm[pos] = (int32_t)e;
    src[pos] = s32[(size_t)e];
    dst[pos] = d32[(size_t)e];
    memcpy(attr + 4 * (int64_t)pos, e_attr + 4 * e, 16);
  }
  // proposal segments (k_prep_count's second half): seg_ptr[p] = first row whose proposal id is >= p
  if (bbox_idx != nullptr) {
    if (!seg_ptr || !node_seg || P <= 0) return YOLAT_E_INVALID;
    for (int64_t p = 0; p <= P; ++p) seg_ptr[p] = 0;
    for (int64_t t = 0; t <= N; ++t) {
      int64_t prev = (t == 0) ? -1 : bbox_idx[t - 1];
      int64_t cur = (t == N) ? P : bbox_idx[t];
      if (t < N) {
        if (cur < 0 || cur >= P) { st |= YOLAT_STATUS_SEG_RANGE; cur = cur < 0 ? 0 : P - 1; }
        node_seg[t] = (int32_t)cur;
      }
      if (prev >= P) prev = P - 1;
      if (prev < -1) prev = -1;
      if (cur < prev) st |= YOLAT_STATUS_SEG_UNSORTED;
      else for (int64_t p = prev + 1; p <= cur; ++p) seg_ptr[p] = (int32_t)t;
    }
  }
  *status = st;
  return 0;
}

// The locality record of ONE dataset item on the host (the twin of yolat_batch_locality): is its edge list grouped by
// proposal (Datasets/graph_dict3.py:725,752-764), does every edge stay inside its proposal (:582-600,733), how large is the
// largest proposal.  A batch of items inherits the OR of the flags and the maxima: collate adds per-image offsets
// (train.py:238-258) and keeps the order, so the batch's list is grouped iff every item's is.  Computed once per item and
// cached with it (data.item_locality), like the item's CSR.
extern "C" int yolat_item_locality_host(const int64_t* edge, int64_t stride_e, int64_t stride_c, const int64_t* bbox_idx,
                                        int64_t E, int64_t N, int64_t P, yolat_locality* out) {
  if (!out || N <= 0 || E < 0 || P <= 0 || !bbox_idx || (E > 0 && !edge)) return YOLAT_E_INVALID;
  if (N >= (1LL << 30) || E >= (1LL << 30) || P >= (1LL << 30)) return YOLAT_E_UNSUPPORTED;
  int flags = 0;
  std::vector<int32_t> nn((size_t)P, 0), ne((size_t)P, 0);
  int64_t prev = -1;
  for (int64_t n = 0; n < N; ++n) {
    int64_t b = bbox_idx[n];
    if (b < 0 || b >= P) { flags |= YOLAT_LOC_MALFORMED; b = b < 0 ? 0 : P - 1; }
    if (b < prev) flags |= YOLAT_LOC_MALFORMED;
    prev = b;
    ++nn[(size_t)b];
  }
  auto seg = [&](int64_t v) -> int64_t {
    if (v < 0 || v >= N) { flags |= YOLAT_LOC_MALFORMED; v = v < 0 ? 0 : N - 1; }
    int64_t b = bbox_idx[v];
    return b < 0 ? 0 : (b >= P ? P - 1 : b);
  };
  prev = -1;
  for (int64_t e = 0; e < E; ++e) {
    const int64_t ps = seg(edge[e * stride_e]), pd = seg(edge[e * stride_e + stride_c]);
    if (ps != pd) flags |= YOLAT_LOC_CROSSING;
    if (pd < prev) flags |= YOLAT_LOC_UNGROUPED;
    prev = pd;
    ++ne[(size_t)pd];
  }
  int32_t mn = 0, me = 0;
  for (int64_t p = 0; p < P; ++p) { mn = nn[(size_t)p] > mn ? nn[(size_t)p] : mn; me = ne[(size_t)p] > me ? ne[(size_t)p] : me; }
  out->known = 1; out->flags = flags; out->max_nodes = mn; out->max_edges = me;
  return 0;
}

extern "C" int yolat_collate_csr_pack(const yolat_item_csr* items, int64_t B, int32_t* row_ptr, int32_t* src, int32_t* dst,
                                      float* attr, int32_t* seg_ptr, int32_t* node_seg) {
  if (!items || B <= 0 || !row_ptr || !seg_ptr || !node_seg) return YOLAT_E_INVALID;
  int64_t noff = 0, eoff = 0, poff = 0;
  for (int64_t b = 0; b < B; ++b) {
    const yolat_item_csr& it = items[b];
    if (it.N <= 0 || it.E < 0 || it.P <= 0 || !it.row_ptr || !it.seg_ptr || !it.node_seg) return YOLAT_E_INVALID;
    if (it.E > 0 && (!it.src || !it.dst || !it.attr || !src || !dst || !attr)) return YOLAT_E_INVALID;
    if (noff + it.N >= (1LL << 31) || eoff + it.E >= (1LL << 31)) return YOLAT_E_UNSUPPORTED;
    for (int64_t n = 0; n < it.N; ++n) row_ptr[noff + n] = it.row_ptr[n] + (int32_t)eoff;
    for (int64_t e = 0; e < it.E; ++e) {
      src[eoff + e] = it.src[e] + (int32_t)noff;
      dst[eoff + e] = it.dst[e] + (int32_t)noff;
    }
    if (it.E) memcpy(attr + 4 * eoff, it.attr, (size_t)it.E * 16);
    for (int64_t p = 0; p < it.P; ++p) seg_ptr[poff + p] = it.seg_ptr[p] + (int32_t)noff;
    for (int64_t n = 0; n < it.N; ++n) node_seg[noff + n] = it.node_seg[n] + (int32_t)poff;
    noff += it.N; eoff += it.E; poff += it.P;
  }
  row_ptr[noff] = (int32_t)eoff;
  seg_ptr[poff] = (int32_t)noff;
  return 0;
}

// The whole csr-mode hand-over of a batch in ONE call: layout (256-byte aligned fields: the dense keys in order, then
// row_ptr, src, dst, attr, seg_ptr, node_seg), the slices tables (cumulative rows per key, train.py:141-147), the packed
// copy of every key of every item and the merged CSR.  `items` are the per-item descriptors a caller caches with its
// dataset items.  When `dst` is NULL or `cap` is too small only off / total / slices / totals are produced (the caller
// grows its staging buffer and calls again).
extern "C" int yolat_collate_batch(const yolat_item_desc* const* items, int64_t B, void* dst, int64_t cap, int64_t* off,
                                   int64_t* total, int64_t* slices, int64_t* totals) {
  if (!items || B <= 0 || !off || !total || !slices || !totals) return YOLAT_E_INVALID;
  const int64_t nk = items[0]->n_keys;
  if (nk < 0 || nk > YOLAT_MAX_KEYS) return YOLAT_E_INVALID;
  int64_t N = 0, E = 0, P = 0;
  std::vector<int64_t> fbytes((size_t)nk, 0);
  for (int64_t k = 0; k < nk; ++k) slices[k * (B + 1)] = 0;
  for (int64_t b = 0; b < B; ++b) {
    const yolat_item_desc* it = items[b];
    if (!it || it->n_keys != nk) return YOLAT_E_INVALID;
    for (int64_t k = 0; k < nk; ++k) {
      fbytes[(size_t)k] += it->key[k].bytes;
      slices[k * (B + 1) + b + 1] = slices[k * (B + 1) + b] + it->rows[k];
    }
    N += it->csr.N; E += it->csr.E; P += it->csr.P;
  }
  if (N >= (1LL << 31) || E >= (1LL << 31)) return YOLAT_E_UNSUPPORTED;
  totals[0] = N; totals[1] = E; totals[2] = P;
  int64_t o = 0;
  auto place = [&](int64_t f, int64_t bytes) { off[f] = o; o = (o + bytes + 255) / 256 * 256; };
  for (int64_t k = 0; k < nk; ++k) place(k, fbytes[(size_t)k]);
  const int64_t Ee = E > 0 ? E : 1;
  place(nk + 0, (N + 1) * 4); place(nk + 1, Ee * 4); place(nk + 2, Ee * 4); place(nk + 3, Ee * 16);
  place(nk + 4, (P + 1) * 4); place(nk + 5, N * 4);
  *total = o > 256 ? o : 256;
  if (!dst || cap < *total) return 0;
  char* base = reinterpret_cast<char*>(dst);
  for (int64_t k = 0; k < nk; ++k) {
    char* w = base + off[k];
    for (int64_t b = 0; b < B; ++b) {
      const yolat_span& sp = items[b]->key[k];
      if (sp.bytes < 0 || (sp.bytes > 0 && !sp.ptr)) return YOLAT_E_INVALID;
      const int fx = items[b]->fix[k];
      if (fx == 0) {
        if (sp.bytes) memcpy(w, sp.ptr, (size_t)sp.bytes);
      } else {
        // an int64 index tensor with the offset of its item added on the way (train.py:238-258)
        const int64_t ok = fx == 1 ? items[b]->node_key : (fx == 2 ? items[b]->prop_key : -1);
        if (ok < 0 || ok >= nk || sp.bytes % 8 != 0) return YOLAT_E_INVALID;
        const int64_t add = slices[ok * (B + 1) + b];
        const int64_t* in = reinterpret_cast<const int64_t*>(sp.ptr);
        int64_t* out = reinterpret_cast<int64_t*>(w);
        const int64_t n = sp.bytes / 8;
        for (int64_t i = 0; i < n; ++i) out[i] = in[i] + add;
      }
      w += sp.bytes;
    }
  }
  // COO mode: no item carries a prepared graph (all csr members zero) — the six graph fields stay empty
  bool any_csr = false, all_csr = true;
  for (int64_t b = 0; b < B; ++b) {
    const bool has = items[b]->csr.N > 0 || items[b]->csr.row_ptr != nullptr;
    any_csr = any_csr || has;
    all_csr = all_csr && has;
  }
  if (!any_csr) return 0;
  if (!all_csr) return YOLAT_E_INVALID;
  std::vector<yolat_item_csr> cs((size_t)B);
  for (int64_t b = 0; b < B; ++b) cs[(size_t)b] = items[b]->csr;
  return yolat_collate_csr_pack(cs.data(), B, reinterpret_cast<int32_t*>(base + off[nk + 0]),
                                reinterpret_cast<int32_t*>(base + off[nk + 1]), reinterpret_cast<int32_t*>(base + off[nk + 2]),
                                reinterpret_cast<float*>(base + off[nk + 3]), reinterpret_cast<int32_t*>(base + off[nk + 4]),
                                reinterpret_cast<int32_t*>(base + off[nk + 5]));
}
