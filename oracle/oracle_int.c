/* CPU ORACLE (plain C) — TEST INFRASTRUCTURE ONLY; never linked into the product.
 *
 * Integer pre-processing contract of libyolat_hip.so restated in scalar C: stable counting sort of
 * the COO edge list by destination (what PyG's propagate does implicitly with edge_index,
 * gcn_lib/sparse/torch_vertex.py:324, made explicit) and segment pointers of the sorted bbox_idx
 * (Datasets/graph_dict3.py:732; torch_scatter.scatter index at architecture3cc_rpn_gp_iter2.py:67,122).
 * Parity status: see oracle/oracle_torch.py.  Built by __graft_entry__.build() into
 * oracle/_build/liboracle_int.so; checked against oracle_np.py and the golden integer fixtures.
 */
#include <stdint.h>
#include <stdlib.h>

int oracle_coo_to_csr(const int64_t* src, const int64_t* dst, int64_t E, int64_t N, int32_t* row_ptr,
                      int32_t* perm, int32_t* src_csr, int32_t* dst_csr) {
  int64_t e, i;
  int32_t* cursor;
  for (i = 0; i <= N; ++i) row_ptr[i] = 0;
  for (e = 0; e < E; ++e) {
    if (dst[e] < 0 || dst[e] >= N || src[e] < 0 || src[e] >= N) return 1;
    row_ptr[dst[e] + 1] += 1;
  }
  for (i = 0; i < N; ++i) row_ptr[i + 1] += row_ptr[i];
  cursor = (int32_t*)malloc(sizeof(int32_t) * (size_t)(N > 0 ? N : 1));
  if (!cursor) return 2;
  for (i = 0; i < N; ++i) cursor[i] = row_ptr[i];
  for (e = 0; e < E; ++e) { /* ascending e => stable */
    int32_t pos = cursor[dst[e]]++;
    perm[pos] = (int32_t)e;
    src_csr[pos] = (int32_t)src[e];
    dst_csr[pos] = (int32_t)dst[e];
  }
  free(cursor);
  return 0;
}

int oracle_segment_ptr(const int64_t* bbox_idx, int64_t N, int64_t P, int32_t* seg_ptr) {
  int64_t p, r = 0;
  for (p = 0; p <= P; ++p) {
    while (r < N && bbox_idx[r] < p) ++r;
    seg_ptr[p] = (int32_t)r;
  }
  for (r = 1; r < N; ++r)
    if (bbox_idx[r] < bbox_idx[r - 1]) return 1;
  return 0;
}
