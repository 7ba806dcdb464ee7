// graph.hip — integer pre-processing of one batch (bit-exact contract, see oracle/oracle_np.py):
//   COO edge list (int64, as the reference hands it to GraphConv: architecture3cc_rpn_gp_iter2.py:110)
//   -> destination-sorted CSR (stable: ascending edge id inside a row), CSC by source for the
//   backward, proposal segment pointers from the sorted bbox_idx, row gather for e_attr.
// Counting sort = histogram (int atomics: order-independent result) -> single-workgroup exclusive
// scan -> cursor fill (unordered) -> per-row insertion sort by edge id (restores the stable order;
// rows are short: in-degree of a Bezier end point is a handful).
#include "common.hpp"

__global__ void k_edge_count(const int64_t* edge, long se, long sc, int E, int N, int* src32,
                             int* dst32, int* cnt, int* status) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= E) return;
  int64_t s = edge[(long)e * se];
  int64_t d = edge[(long)e * se + sc];
  if (s < 0 || s >= N || d < 0 || d >= N) {
    // flag it and clamp, so that every downstream kernel stays memory-safe; the caller must treat
    // the results as invalid once it has read the status word
    atomicOr(status, YOLAT_STATUS_EDGE_RANGE);
    s = s < 0 ? 0 : (s >= N ? N - 1 : s);
    d = d < 0 ? 0 : (d >= N ? N - 1 : d);
  }
  src32[e] = (int)s; dst32[e] = (int)d;
  atomicAdd(&cnt[d], 1);
}

__global__ void k_count32(const int* key, int n, int* cnt) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int k = key[i];
  if (k >= 0) atomicAdd(&cnt[k], 1);
}

// Exclusive scan of cnt[0..N) into ptr[0..N], ptr[N] = total; cnt is zeroed (it becomes the fill
// cursor).  One 1024-thread workgroup, 4 elements per thread per sweep.
__global__ void __launch_bounds__(1024) k_scan_excl(int* cnt, int N, int* ptr) {
  __shared__ int wsum[16];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  int carry = 0;
  for (int base = 0; base < N; base += 4096) {
    const int idx = base + tid * 4;
    int v[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      v[j] = (idx + j < N) ? cnt[idx + j] : 0;
      if (idx + j < N) cnt[idx + j] = 0;
    }
    const int t = v[0] + v[1] + v[2] + v[3];
    int incl = t;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      const int nb = __shfl_up(incl, off);
      if (lane >= off) incl += nb;
    }
    if (lane == 63) wsum[wave] = incl;
    __syncthreads();
    int woff = 0, total = 0;
#pragma unroll
    for (int w = 0; w < 16; ++w) {
      const int s = wsum[w];
      if (w < wave) woff += s;
      total += s;
    }
    int excl = carry + woff + incl - t;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      if (idx + j < N) ptr[idx + j] = excl;
      excl += v[j];
    }
    carry += total;
    __syncthreads();
  }
  if (tid == 0) ptr[N] = carry;
}

__global__ void k_fill(const int* key, int n, const int* ptr, int* cursor, int* out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int k = key[i];
  if (k < 0) return;
  const int pos = ptr[k] + atomicAdd(&cursor[k], 1);
  out[pos] = i;
}

__global__ void k_sort_rows(const int* ptr, int* items, int N) {
  const int n = blockIdx.x * blockDim.x + threadIdx.x;
  if (n >= N) return;
  const int b = ptr[n], e = ptr[n + 1];
  for (int i = b + 1; i < e; ++i) {
    const int v = items[i];
    int j = i - 1;
    while (j >= b && items[j] > v) { items[j + 1] = items[j]; --j; }
    items[j + 1] = v;
  }
}

__global__ void k_emit_csr(const int* perm_tmp, const int* src32, const int* dst32, int total,
                           int* perm, int* src_csr, int* dst_csr) {
  const int q = blockIdx.x * blockDim.x + threadIdx.x;
  if (q >= total) return;
  const int e = perm_tmp[q];
  perm[q] = e;
  src_csr[q] = src32[e];
  dst_csr[q] = dst32[e];
}

extern "C" size_t yolat_csr_work_elems(int64_t N, int64_t E) { return (size_t)(N + 3 * E + 16); }

extern "C" int yolat_coo_to_csr(const int64_t* edge, int64_t stride_e, int64_t stride_c, int64_t E,
                                int64_t N, int32_t* row_ptr, int32_t* perm, int32_t* src_csr,
                                int32_t* dst_csr, int32_t* work, int32_t* status,
                                yolat_stream_t stream) {
  if (N <= 0 || E < 0 || N >= (1LL << 31) - 4096 || E >= (1LL << 31) - 256) return YOLAT_E_INVALID;
  if (!row_ptr || !work || !status || (E > 0 && (!edge || !perm || !src_csr || !dst_csr)))
    return YOLAT_E_INVALID;
  hipStream_t st = (hipStream_t)stream;
  int* cnt = work;
  int* src32 = work + N;
  int* dst32 = src32 + E;
  int* tmp = dst32 + E;
  hipError_t err = hipMemsetAsync(cnt, 0, sizeof(int) * (size_t)N, st);
  if (err != hipSuccess) return (int)err;
  if (E > 0) {
    hipLaunchKernelGGL(k_edge_count, dim3(yl_cdiv(E, 256)), dim3(256), 0, st, edge, (long)stride_e,
                       (long)stride_c, (int)E, (int)N, src32, dst32, cnt, status);
    YL_LAUNCH_CHECK();
  }
  hipLaunchKernelGGL(k_scan_excl, dim3(1), dim3(1024), 0, st, cnt, (int)N, row_ptr);
  YL_LAUNCH_CHECK();
  if (E > 0) {
    hipLaunchKernelGGL(k_fill, dim3(yl_cdiv(E, 256)), dim3(256), 0, st, dst32, (int)E, row_ptr, cnt,
                       tmp);
    YL_LAUNCH_CHECK();
    hipLaunchKernelGGL(k_sort_rows, dim3(yl_cdiv(N, 256)), dim3(256), 0, st, row_ptr, tmp, (int)N);
    YL_LAUNCH_CHECK();
    hipLaunchKernelGGL(k_emit_csr, dim3(yl_cdiv(E, 256)), dim3(256), 0, st, tmp, src32, dst32,
                       (int)E, perm, src_csr, dst_csr);
    YL_LAUNCH_CHECK();
  }
  return 0;
}

// col_ptr[i] = block-local exclusive scan + prefix of the block totals; the counters are zeroed so they
// can serve as the fill cursors.  n = N + 1 (the last element is the grand total).
__global__ void k_csc_ptr(int* local, const int* btot, int n, int* ptr);
__global__ void __launch_bounds__(1024) k_prep_scan(int* cnt, int n, int* local, int* btot);

extern "C" size_t yolat_csc_work_elems(int64_t N) { return (size_t)(N + 1 + (N + 1 + 4095) / 4096 + 16); }

extern "C" int yolat_csc_by_source(const int32_t* src_csr, int64_t E, int64_t N, int32_t* col_ptr,
                                   int32_t* slots, int32_t* work, yolat_stream_t stream) {
  if (N <= 0 || E < 0 || !col_ptr || !work || (E > 0 && (!src_csr || !slots)))
    return YOLAT_E_INVALID;
  hipStream_t st = (hipStream_t)stream;
  const int n1 = (int)N + 1;
  int* cnt = work;             // [N+1] counters -> block-local scan (in place) -> fill cursors
  int* btot = work + n1;       // [ceil((N+1)/4096)]
  hipError_t err = hipMemsetAsync(cnt, 0, sizeof(int) * (size_t)n1, st);
  if (err != hipSuccess) return (int)err;
  if (E > 0) {
    hipLaunchKernelGGL(k_count32, dim3(yl_cdiv(E, 256)), dim3(256), 0, st, src_csr, (int)E, cnt);
    YL_LAUNCH_CHECK();
  }
  // multi-workgroup scan (a single-workgroup sweep over N = 174k took 121 us)
  hipLaunchKernelGGL(k_prep_scan, dim3(yl_cdiv(n1, 4096)), dim3(1024), 0, st, cnt, n1, cnt, btot);
  YL_LAUNCH_CHECK();
  hipLaunchKernelGGL(k_csc_ptr, dim3(yl_cdiv(n1, 256)), dim3(256), 0, st, cnt, btot, n1, col_ptr);
  YL_LAUNCH_CHECK();
  if (E > 0) {
    hipLaunchKernelGGL(k_fill, dim3(yl_cdiv(E, 256)), dim3(256), 0, st, src_csr, (int)E, col_ptr,
                       cnt, slots);
    YL_LAUNCH_CHECK();
    hipLaunchKernelGGL(k_sort_rows, dim3(yl_cdiv(N, 256)), dim3(256), 0, st, col_ptr, slots, (int)N);
    YL_LAUNCH_CHECK();
  }
  return 0;
}

// seg_ptr[p] = first row r with bbox_idx[r] >= p.  Thread r in [0, N] owns the boundary between
// rows r-1 and r and writes seg_ptr for every p in (bbox[r-1], bbox[r]].
__global__ void k_segment_ptr(const int64_t* bbox, long N, long P, int* seg_ptr, int* node_seg,
                              int* status) {
  const long r = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (r > N) return;
  long prev = (r == 0) ? -1 : bbox[r - 1];
  long cur = (r == N) ? P : bbox[r];
  if (r < N) {
    if (cur < 0 || cur >= P) { atomicOr(status, YOLAT_STATUS_SEG_RANGE); cur = cur < 0 ? 0 : P - 1; }
    node_seg[r] = (int)cur;
  }
  if (prev >= P) prev = P - 1;
  if (prev < -1) prev = -1;
  if (cur < prev) { atomicOr(status, YOLAT_STATUS_SEG_UNSORTED); return; }
  for (long p = prev + 1; p <= cur; ++p) seg_ptr[p] = (int)r;
}

extern "C" int yolat_segment_ptr(const int64_t* bbox_idx, int64_t N, int64_t P, int32_t* seg_ptr,
                                 int32_t* node_seg, int32_t* status, yolat_stream_t stream) {
  if (N < 0 || P <= 0 || N >= (1LL << 31) - 1 || !seg_ptr || !status || (N > 0 && (!bbox_idx || !node_seg)))
    return YOLAT_E_INVALID;
  hipLaunchKernelGGL(k_segment_ptr, dim3(yl_cdiv(N + 1, 256)), dim3(256), 0, (hipStream_t)stream,
                     bbox_idx, (long)N, (long)P, seg_ptr, node_seg, status);
  YL_LAUNCH_CHECK();
  return 0;
}

// ------------------------------------------------------------------------------------------------
// yolat_graph_prepare: the whole per-batch structure in 4 launches + 1 memset.
//   1. k_prep_count_win  edge e: int64 -> int32, range check, rank[e] = a unique position inside its row (arrival
//                     order: only used to place e somewhere inside its row); row r: segment pointers
//   2. k_prep_scan    per-4096-element-block exclusive scan of cnt (multi-workgroup) + block totals
//   3. k_prep_fill    items[local[dst] + blockprefix + rank[e]] = e
//   4. k_prep_rows    slot t: rank of items[t] among its row's items by edge id (restores the stable order),
//                     emit perm / src / dst / attr at row start + rank; row n: the final row_ptr
// ------------------------------------------------------------------------------------------------
#define PREP_BLK 4096
__global__ void __launch_bounds__(256) k_zero_i32(int* p, int n) {      // n a multiple of 4, p 16-byte aligned or not
  const int i = (blockIdx.x * 256 + threadIdx.x) * 4;
#pragma unroll
  for (int j = 0; j < 4; ++j)
    if (i + j < n) p[i + j] = 0;
}
// Counting pre-aggregated per workgroup (round 4).  One RETURNING device-scope atomic per edge (rounds 1-3) executes
// behind the XCDs' L2s, ~32 per ns for the whole device: 38 us at E = 1.2 M, a third of the large-graph preparation.  Edge lists of this domain are grouped by proposal (graph_dict3.py:582-600 builds them
// proposal by proposal; collate concatenates graphs), so the PCW_E consecutive edges of a workgroup point into a narrow
// window of destination rows.  When that window is at most PCW_WIN rows the workgroup ranks its edges with LDS atomics
// and sends ONE returning global atomic per touched row (count added, base returned): rank = base + rank inside the
// workgroup.  Any window wider than that (shuffled edge lists) takes the direct path — same result either way, because
// the rank only places an edge somewhere inside its row (k_prep_rows restores the stable order).
#define PCW_T 512
#define PCW_PER 4
#define PCW_E (PCW_T * PCW_PER)
#define PCW_WIN 4096
__global__ void __launch_bounds__(PCW_T) k_prep_count_win(const int64_t* edge, long se, long sc, int E, int N, int* src32,
                                                          int* dst32, int* rank, int* cnt, const int64_t* bbox, long P,
                                                          int* seg_ptr, int* node_seg, int* status) {
  __shared__ int win[PCW_WIN];
  __shared__ int wlo[PCW_T / 64], whi[PCW_T / 64];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int e0 = blockIdx.x * PCW_E;
  int d[PCW_PER];
  int lo = 0x7fffffff, hi = -1;
#pragma unroll
  for (int j = 0; j < PCW_PER; ++j) {
    const int t = e0 + tid + PCW_T * j;
    d[j] = -1;
    if (t < E) {
      int64_t s = edge[(long)t * se];
      int64_t dd = edge[(long)t * se + sc];
      if (s < 0 || s >= N || dd < 0 || dd >= N) {
        atomicOr(status, YOLAT_STATUS_EDGE_RANGE);
        s = s < 0 ? 0 : (s >= N ? N - 1 : s);
        dd = dd < 0 ? 0 : (dd >= N ? N - 1 : dd);
      }
      src32[t] = (int)s; dst32[t] = (int)dd;
      d[j] = (int)dd;
      lo = yl_min(lo, d[j]); hi = yl_max(hi, d[j]);
    }
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    lo = yl_min(lo, __shfl_xor(lo, off));
    hi = yl_max(hi, __shfl_xor(hi, off));
  }
  if (lane == 0) { wlo[wave] = lo; whi[wave] = hi; }
  __syncthreads();
#pragma unroll
  for (int w = 0; w < PCW_T / 64; ++w) { lo = yl_min(lo, wlo[w]); hi = yl_max(hi, whi[w]); }
  const int span = hi - lo + 1;                           // <= 0: no edge in this workgroup
  if (span > 0 && span <= PCW_WIN) {
    for (int i = tid; i < span; i += PCW_T) win[i] = 0;
    __syncthreads();
    int lr[PCW_PER];
#pragma unroll
    for (int j = 0; j < PCW_PER; ++j) lr[j] = d[j] >= 0 ? atomicAdd(&win[d[j] - lo], 1) : 0;
    __syncthreads();
    for (int i = tid; i < span; i += PCW_T) {
      const int c = win[i];
      if (c > 0) win[i] = atomicAdd(&cnt[lo + i], c);
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < PCW_PER; ++j)
      if (d[j] >= 0) rank[e0 + tid + PCW_T * j] = win[d[j] - lo] + lr[j];
  } else if (span > 0) {
#pragma unroll
    for (int j = 0; j < PCW_PER; ++j)
      if (d[j] >= 0) rank[e0 + tid + PCW_T * j] = atomicAdd(&cnt[d[j]], 1);
  }
  if (bbox != nullptr) {
#pragma unroll
    for (int j = 0; j < PCW_PER; ++j) {
      const int t = e0 + tid + PCW_T * j;
      if (t > N) continue;
      long prev = (t == 0) ? -1 : bbox[t - 1];
      long cur = (t == N) ? P : bbox[t];
      if (t < N) {
        if (cur < 0 || cur >= P) { atomicOr(status, YOLAT_STATUS_SEG_RANGE); cur = cur < 0 ? 0 : P - 1; }
        node_seg[t] = (int)cur;
      }
      if (prev >= P) prev = P - 1;
      if (prev < -1) prev = -1;
      if (cur < prev) atomicOr(status, YOLAT_STATUS_SEG_UNSORTED);
      else for (long p = prev + 1; p <= cur; ++p) seg_ptr[p] = t;
    }
  }
}

// local[i] = exclusive scan of cnt inside its PREP_BLK block; btot[b] = block total.  n = N + 1.
// cnt is left ZERO (its only reader is this kernel): a caller that keeps the work buffer to itself can skip the memset of
// the next call with the same shape (yl_graph_prepare_impl's `primed`).
__global__ void __launch_bounds__(1024) k_prep_scan(int* cnt, int n, int* local, int* btot) {
  __shared__ int wsum[16];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int idx = blockIdx.x * PREP_BLK + tid * 4;
  int v[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    v[j] = (idx + j < n) ? cnt[idx + j] : 0;
    if (idx + j < n) cnt[idx + j] = 0;
  }
  const int t = v[0] + v[1] + v[2] + v[3];
  int incl = t;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const int nb = __shfl_up(incl, off);
    if (lane >= off) incl += nb;
  }
  if (lane == 63) wsum[wave] = incl;
  __syncthreads();
  int woff = 0, total = 0;
#pragma unroll
  for (int w = 0; w < 16; ++w) {
    const int s = wsum[w];
    if (w < wave) woff += s;
    total += s;
  }
  int excl = woff + incl - t;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    if (idx + j < n) local[idx + j] = excl;
    excl += v[j];
  }
  if (tid == 0) btot[blockIdx.x] = total;
}

__device__ __forceinline__ int prep_prefix(const int* btot, int b) {
  int s = 0;
  for (int i = 0; i < b; ++i) s += btot[i];
  return s;
}

// Exclusive scan of the block totals into LDS (sh[b] = sum_{i<b} btot[i], b <= nblk <= PREP_OFFS), built by every
// 256-thread workgroup for itself (a per-thread prep_prefix() is a loop over up to N/4096 block totals).
// Returns false (sh unused) when there are more blocks than the table holds.  Whole workgroup must call.
#define PREP_OFFS 1024
__device__ __forceinline__ bool prep_offsets_lds(const int* btot, int nblk, int* sh, int* wsum) {
  if (nblk > PREP_OFFS) return false;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  int v[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) v[j] = (tid * 4 + j < nblk) ? btot[tid * 4 + j] : 0;
  const int t = v[0] + v[1] + v[2] + v[3];
  int incl = t;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const int nb = __shfl_up(incl, off);
    if (lane >= off) incl += nb;
  }
  if (lane == 63) wsum[wave] = incl;
  __syncthreads();
  int woff = 0;
#pragma unroll
  for (int w = 0; w < 4; ++w) woff += (w < wave) ? wsum[w] : 0;
  int excl = woff + incl - t;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    sh[tid * 4 + j] = excl;
    excl += v[j];
  }
  __syncthreads();
  return true;
}

__global__ void k_csc_ptr(int* local, const int* btot, int n, int* ptr) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  ptr[i] = local[i] + prep_prefix(btot, i / PREP_BLK);
  local[i] = 0;
}

__global__ void k_prep_fill(const int* dst32, const int* rank, int E, const int* local, const int* btot,
                            int* items) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= E) return;
  const int d = dst32[e];
  items[local[d] + prep_prefix(btot, d / PREP_BLK) + rank[e]] = e;
}

// One thread per CSR SLOT (and per row for row_ptr): thread t takes the t-th entry of `items` (edges grouped by
// destination, arbitrary order inside a row), finds its rank among the row's items by edge id — that restores the
// stable order of the reference's index_select / scatter — and emits the edge at row start + rank.  The loads of
// items and the four stores are (nearly) coalesced because neighbouring threads work on the same or adjacent rows.
// (A thread-per-row version with the row sorted in registers was TA-bound on its scattered accesses:
// 66 us for 1.2 M edges.)
__device__ __forceinline__ void prep_rows_body(int t, const int* local, const int* btot, int N, int E, const int* items,
                                               const int* src32, const int* dst32, const float4* attr, int* row_ptr,
                                               int* perm, int* src_csr, int* dst_csr, float4* attr_csr) {
  __shared__ int sh[PREP_OFFS + 4], wsum[4];
  const bool have = prep_offsets_lds(btot, (N + 1 + PREP_BLK - 1) / PREP_BLK, sh, wsum);   // whole workgroup
  auto off = [&](int i) { return local[i] + (have ? sh[i / PREP_BLK] : prep_prefix(btot, i / PREP_BLK)); };
  if (t <= N) row_ptr[t] = off(t);
  if (t >= E) return;
  const int e = items[t];
  const int i = dst32[e];
  const int b = off(i), en = off(i + 1);
  int r = 0;
  for (int j = b; j < en; j += 8) {          // 8 independent (clamped) loads per step
    int v[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) v[k] = items[yl_min(j + k, en - 1)];
#pragma unroll
    for (int k = 0; k < 8; ++k) r += (j + k < en && v[k] < e) ? 1 : 0;
  }
  const int pos = b + r;
  const int sv = src32[e];
  const float4 av = attr[e];
  perm[pos] = e;
  src_csr[pos] = sv;
  dst_csr[pos] = i;
  attr_csr[pos] = av;
}

__global__ void __launch_bounds__(256) k_prep_rows(const int* local, const int* btot, int N, int E, const int* items,
                                                   const int* src32, const int* dst32, const float4* attr, int* row_ptr,
                                                   int* perm, int* src_csr, int* dst_csr, float4* attr_csr) {
  prep_rows_body(blockIdx.x * 256 + threadIdx.x, local, btot, N, E, items, src32, dst32, attr, row_ptr, perm, src_csr,
                 dst_csr, attr_csr);
}

// The last pre-processing kernel co-scheduled with the node side of the FIRST conv layer (which depends only
// on x, not on the graph): workgroups [0, rows_blocks) sort / emit the CSR rows, the rest are the 64x64 GEMM
// tiles of yolat_node_uv_eval.  The row kernel alone is 40 workgroups of integer latency (~10 us at
// N = 10k); the GEMM tiles fill the other CUs meanwhile instead of being a launch of their own.
template <int BK>
__global__ void __launch_bounds__(256) k_prep_rows_node3(const int* local, const int* btot, int N, int E,
                                                         const int* items, const int* src32, const int* dst32,
                                                         const float4* attr, int* row_ptr, int* perm, int* src_csr,
                                                         int* dst_csr, float4* attr_csr, int rows_blocks, NodeUv a) {
  if ((int)blockIdx.x < rows_blocks) {
    prep_rows_body(blockIdx.x * 256 + threadIdx.x, local, btot, N, E, items, src32, dst32, attr, row_ptr, perm, src_csr,
                   dst_csr, attr_csr);
    return;
  }
  const int t = blockIdx.x - rows_blocks;
  node_uv_tile<BK>(a, t >> 2, t & 3);
}

// ------------------------------------------------------------------------------------------------
// One-launch graph preparation for SMALL graphs (k_prep_small).  At E = 40 k the four launches above are ~5 us of
// dependent latency each (24 us of a 117 us forward) and every in-launch hand-off between workgroups costs as much as a
// launch boundary on this part (8 XCDs with private L2s: DESIGN.md Appendix R).  So this kernel has NO communication
// between workgroups at all: workgroup b owns the destination rows [b R, (b+1) R) and finds their edges by reading the
// WHOLE edge list itself (E x 16 bytes through its L1: ~4 us at E = 40 k, the price of the scheme and the reason it is
// for small graphs only):
//   walk   wave w reads the edge range [w E/16, (w+1) E/16), 64 edges per step, eight steps of loads in flight:
//          below += [dst < r0]; an OWNED edge (r0 <= dst < r1) bumps its row's counter (LDS atomic, no return) and is
//          appended to a list in LDS (one slot allocation per eight steps)
//   sums   base = row_ptr[r0] = the workgroup's total of `below`; exclusive scan of the row counters -> row_ptr
//   group  list -> row buckets (LDS cursor per row; any order)
//   rank   every bucket entry counts the entries of its row with a smaller edge id — the stable order of the
//          reference's index_select / scatter (rows are a handful of edges) — and emits perm / src / dst / attr at
//          base + rowstart + rank
// A workgroup whose rows hold more than PS_CAP edges (skewed graphs) takes the ORDERED path instead: two more walks of
// the edge list in ascending edge order per wave.  Rank of an owned edge among the lanes of its step with the same
// destination: every owned lane ORs its lane bit into the wave's 64-bit word of that row (commutative: independent of
// the order the lanes are served in) and reads the word back = the match mask of its key; rank = wave's running row
// counter + popcount(mask below me).  First walk: per-wave row counts; exclusive prefix over the waves; second walk:
// emit at base + rowstart + prefix + rank.  No sort, O(E) per workgroup whatever the degrees.
// Proposal segments (k_prep_count's second half) and the node side of the first conv layer (K = in_channels <= 8:
// node3_smallk_body) in extra workgroups of the same launch.  Bit-exact with the four-launch form (tests/test_gpu_ops.py).
// ------------------------------------------------------------------------------------------------
#define PS_T 1024
#define PS_NW (PS_T / 64)
#define PS_RMAX 256
#define PS_CAP 4096
#define PS_PF 8
#define PS_NODE_ITERS 2
struct PrepSmall {
  const int64_t* edge; long se, sc; const float4* attr; const int64_t* bbox; long P;
  int E, N, R, nprep, nseg;
  int* row_ptr; int* perm; int* src_csr; int* dst_csr; float4* attr_csr; int* seg_ptr; int* node_seg; int* status;
};

__device__ __forceinline__ void ps_emit(const PrepSmall& g, int e, int d, int pos) {
  int64_t s = g.edge[(long)e * g.se];
  const float4 av = g.attr[e];
  if (s < 0 || s >= g.N) {
    atomicOr(g.status, YOLAT_STATUS_EDGE_RANGE);
    s = s < 0 ? 0 : g.N - 1;
  }
  g.perm[pos] = e;
  g.src_csr[pos] = (int)s;
  g.dst_csr[pos] = d;
  g.attr_csr[pos] = av;
}

// MODE 0: below + row counters + list;  1: per-wave row counters (ordered);  2: ordered emit (cw = running counters)
template <int MODE>
__device__ __forceinline__ void ps_walk(const PrepSmall& g, int r0, int r1, int* rowcnt, int* cw, unsigned long long* mw,
                                        int* list_n, unsigned* list, int& below, int base, const int* rowstart) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const unsigned long long lt = (1ull << lane) - 1ull;
  const int per = ((g.E + PS_NW * 64 - 1) / (PS_NW * 64)) * 64;
  const int wb = w * per, we = yl_min(wb + per, g.E);
  for (int eb = wb; eb < we; eb += 64 * PS_PF) {
    int64_t dv[PS_PF];
#pragma unroll
    for (int j = 0; j < PS_PF; ++j) {
      const int e = yl_min(eb + 64 * j + lane, g.E - 1);
      dv[j] = g.edge[(long)e * g.se + g.sc];
    }
    if (MODE == 0) {
      unsigned long long om[PS_PF];
      int nown = 0;
#pragma unroll
      for (int j = 0; j < PS_PF; ++j) {
        const bool valid = eb + 64 * j + lane < we;
        const int d = dv[j] < 0 ? 0 : (dv[j] >= g.N ? g.N - 1 : (int)dv[j]);
        below += (valid && d < r0) ? 1 : 0;
        om[j] = __ballot(valid && d >= r0 && d < r1);
        nown += __popcll(om[j]);
      }
      if (nown == 0) continue;                                    // wave-uniform
      int s0 = 0;
      if (lane == 0) s0 = atomicAdd(list_n, nown);
      s0 = __builtin_amdgcn_readfirstlane(s0);
#pragma unroll
      for (int j = 0; j < PS_PF; ++j) {
        if ((om[j] >> lane) & 1ull) {
          const int dl = (dv[j] < 0 ? 0 : (dv[j] >= g.N ? g.N - 1 : (int)dv[j])) - r0, slot = s0 + __popcll(om[j] & lt);
          atomicAdd(&rowcnt[dl], 1);
          if (dv[j] < 0 || dv[j] >= g.N) atomicOr(g.status, YOLAT_STATUS_EDGE_RANGE);
          if (slot < PS_CAP) list[slot] = (unsigned)(eb + 64 * j + lane) | ((unsigned)dl << 20);
        }
        s0 += __popcll(om[j]);
      }
    } else {
#pragma unroll
      for (int j = 0; j < PS_PF; ++j) {
        const int e = eb + 64 * j + lane;
        const int d = dv[j] < 0 ? 0 : (dv[j] >= g.N ? g.N - 1 : (int)dv[j]);
        const bool own = e < we && d >= r0 && d < r1;
        if (__ballot(own) == 0ull) continue;                      // wave-uniform
        const int dl = own ? d - r0 : 0;
        const int c = cw[dl];
        if (own) atomicOr(&mw[dl], 1ull << lane);
        const unsigned long long m = mw[dl];                      // after every lane's OR (LDS serves a wave in order)
        if (own && (m & lt) == 0ull) {                            // the group's lowest lane
          cw[dl] = c + __popcll(m);
          mw[dl] = 0ull;
        }
        if (MODE == 2 && own) ps_emit(g, e, d, base + rowstart[dl] + c + __popcll(m & lt));
      }
    }
  }
}

__global__ void __launch_bounds__(PS_T) k_prep_small(PrepSmall g, NodeUv a, int with_node) {
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  if ((int)blockIdx.x >= g.nprep + g.nseg) {
    if (with_node) node3_smallk_body<PS_NW, PS_NODE_ITERS>(a, blockIdx.x - g.nprep - g.nseg);
    return;
  }
  if ((int)blockIdx.x >= g.nprep) {
    // proposal segments: thread t owns the boundary between rows t - 1 and t (as k_prep_count)
    const int t = (blockIdx.x - g.nprep) * PS_T + tid;
    if (t > g.N) return;
    long prev = (t == 0) ? -1 : g.bbox[t - 1];
    long cur = (t == g.N) ? g.P : g.bbox[t];
    if (t < g.N) {
      if (cur < 0 || cur >= g.P) { atomicOr(g.status, YOLAT_STATUS_SEG_RANGE); cur = cur < 0 ? 0 : g.P - 1; }
      g.node_seg[t] = (int)cur;
    }
    if (prev >= g.P) prev = g.P - 1;
    if (prev < -1) prev = -1;
    if (cur < prev) atomicOr(g.status, YOLAT_STATUS_SEG_UNSORTED);
    else for (long p = prev + 1; p <= cur; ++p) g.seg_ptr[p] = t;
    return;
  }
  // one pool, two layouts (the whole launch is sized by this role: below 80 KB two workgroups share a CU, so the
  // segment / node-side workgroups run beside the CSR ones instead of queueing behind them):
  //   list path     list[PS_CAP] | grp[PS_CAP]            (edge id | local row << 20)
  //   ordered path  msk[PS_NW][PS_RMAX] (64-bit) | cnt[PS_NW][PS_RMAX]
  __shared__ unsigned long long pool[PS_NW * PS_RMAX + PS_NW * PS_RMAX / 2];
  static_assert(2 * PS_CAP * sizeof(unsigned) <= sizeof(unsigned long long) * PS_NW * PS_RMAX, "list + grp fit the mask area");
  unsigned* list = reinterpret_cast<unsigned*>(pool);
  unsigned* grp = list + PS_CAP;
  unsigned long long (*msk)[PS_RMAX] = reinterpret_cast<unsigned long long (*)[PS_RMAX]>(pool);
  int (*cnt)[PS_RMAX] = reinterpret_cast<int (*)[PS_RMAX]>(pool + PS_NW * PS_RMAX);
  __shared__ int rowcnt[PS_RMAX], cur[PS_RMAX], rowstart[PS_RMAX];
  __shared__ int wbelow[PS_NW], wtot[PS_RMAX / 64], list_n;
  const int r0 = blockIdx.x * g.R, r1 = yl_min(r0 + g.R, g.N);
  const bool last = (int)blockIdx.x == g.nprep - 1;
  if (tid < PS_RMAX) { rowcnt[tid] = 0; cur[tid] = 0; }
  if (tid == 0) list_n = 0;
  __syncthreads();
  int below = 0;
  ps_walk<0>(g, r0, r1, rowcnt, nullptr, nullptr, &list_n, list, below, 0, nullptr);
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) below += __shfl_down(below, off);
  if (lane == 0) wbelow[w] = below;
  __syncthreads();
  // exclusive scan of the row counters
  const int rt = tid < PS_RMAX ? rowcnt[tid] : 0;
  int incl = rt;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const int nb = __shfl_up(incl, off);
    if (lane >= off) incl += nb;
  }
  if (tid < PS_RMAX && lane == 63) wtot[w] = incl;
  int base = 0;
#pragma unroll
  for (int v = 0; v < PS_NW; ++v) base += wbelow[v];
  const int n_list = list_n;
  __syncthreads();
  if (tid < PS_RMAX) {
    int off = 0;
#pragma unroll
    for (int v = 0; v < PS_RMAX / 64; ++v) off += (v < w) ? wtot[v] : 0;
    rowstart[tid] = off + incl - rt;
    if (r0 + tid < r1) g.row_ptr[r0 + tid] = base + off + incl - rt;
    if (last && tid == PS_RMAX - 1) g.row_ptr[g.N] = base + off + incl;
  }
  if (n_list <= PS_CAP) {
    // the gathers of the entries this thread will emit go out now (cold lines: the longest latency left), under the
    // grouping phases
    static_assert(PS_CAP == 4 * PS_T, "four list entries per thread, spelled out");
    __syncthreads();
    for (int i = tid; i < n_list; i += PS_T) {
      const unsigned en = list[i];
      const int dl = (int)(en >> 20);
      grp[rowstart[dl] + atomicAdd(&cur[dl], 1)] = en;
    }
    __syncthreads();
    // all gathers first (cold lines: the longest latency left; clamped, unconditional), then rank + store
    if (n_list > 0) {
      const int lastv = n_list - 1;
      auto emit_k = [&](int i, unsigned en, int64_t sv, float a0, float a1, float a2, float a3) {
        if (i >= n_list) return;
        const int dl = (int)(en >> 20), b = rowstart[dl], n = rowcnt[dl];
        int r = 0;
        for (int u = 0; u < n; ++u) r += (grp[b + u] < en) ? 1 : 0;   // same row: the comparison is on the edge id
        if (sv < 0 || sv >= g.N) { atomicOr(g.status, YOLAT_STATUS_EDGE_RANGE); sv = sv < 0 ? 0 : g.N - 1; }
        const int pos = base + b + r;
        g.perm[pos] = (int)(en & 0xFFFFFu);
        g.src_csr[pos] = (int)sv;
        g.dst_csr[pos] = r0 + dl;
        g.attr_csr[pos] = make_float4(a0, a1, a2, a3);
      };
#define PS_G(k)                                                                        \
      const unsigned en##k = list[yl_min(tid + (k) * PS_T, lastv)];                    \
      const int64_t sx##k = g.edge[(long)(en##k & 0xFFFFFu) * g.se];                   \
      const float4 ax##k = g.attr[en##k & 0xFFFFFu];
      PS_G(0) PS_G(1) PS_G(2) PS_G(3)
#undef PS_G
      emit_k(tid, en0, sx0, ax0.x, ax0.y, ax0.z, ax0.w);
      emit_k(tid + PS_T, en1, sx1, ax1.x, ax1.y, ax1.z, ax1.w);
      emit_k(tid + 2 * PS_T, en2, sx2, ax2.x, ax2.y, ax2.z, ax2.w);
      emit_k(tid + 3 * PS_T, en3, sx3, ax3.x, ax3.y, ax3.z, ax3.w);
    }
  } else {
    __syncthreads();
    for (int i = tid; i < PS_NW * PS_RMAX; i += PS_T) { (&cnt[0][0])[i] = 0; (&msk[0][0])[i] = 0ull; }
    __syncthreads();
    int unused = 0;
    ps_walk<1>(g, r0, r1, nullptr, cnt[w], msk[w], nullptr, nullptr, unused, 0, nullptr);
    __syncthreads();
    if (tid < PS_RMAX) {
      int run = 0;
#pragma unroll
      for (int v = 0; v < PS_NW; ++v) { const int t = cnt[v][tid]; cnt[v][tid] = run; run += t; }
    }
    __syncthreads();
    ps_walk<2>(g, r0, r1, nullptr, cnt[w], msk[w], nullptr, nullptr, unused, base, rowstart);
  }
}

// rows per workgroup of k_prep_small, or 0 when the four-launch form is the one to use.  Every workgroup reads all E
// edges: worth it while that is a few microseconds per workgroup and the whole grid is resident at once.
static int prep_small_rows(int64_t N, int64_t E, int other_blocks) {
  static const bool on = []() { const char* e = getenv("YOLAT_PREP_SMALL"); return !(e && e[0] == '0'); }();
  if (!on || E > 98304) return 0;
  const int wgs = 250 - other_blocks;            // one workgroup per CU (16 waves), the whole grid resident at once
  if (wgs < 16) return 0;
  int R = yl_cdiv(N, wgs);
  if (R < 16) R = 16;
  if (R > PS_RMAX) return 0;
  return R;
}

extern "C" size_t yolat_graph_work_elems(int64_t N, int64_t E) {
  return (size_t)(2 * (((N + 1) + 63) / 64 * 64) + 4 * E + (N + 1) / PREP_BLK + 16);
}

int yl_graph_prepare_impl(const int64_t* edge, int64_t stride_e, int64_t stride_c, const float* e_attr,
                              const int64_t* bbox_idx, int64_t E, int64_t N, int64_t P, int32_t* row_ptr,
                              int32_t* perm, int32_t* src_csr, int32_t* dst_csr, float* attr_csr, int32_t* seg_ptr,
                              int32_t* node_seg, int32_t* work, int32_t* status, const NodeUv* extra, bool primed,
                              yolat_stream_t stream) {
  if (N <= 0 || E < 0 || N >= (1LL << 31) - 8192 || E >= (1LL << 31) - 256) return YOLAT_E_INVALID;
  if (!row_ptr || !work || !status) return YOLAT_E_INVALID;
  if (E > 0 && (!edge || !e_attr || !perm || !src_csr || !dst_csr || !attr_csr)) return YOLAT_E_INVALID;
  if (bbox_idx && (P <= 0 || !seg_ptr || !node_seg)) return YOLAT_E_INVALID;
  if (E > 0 && !yl_aligned16(e_attr)) return YOLAT_E_UNSUPPORTED;
  hipStream_t st = (hipStream_t)stream;
  const int n1 = (int)N + 1;
  const int nblk = yl_cdiv(n1, PREP_BLK);
  const int n1p = (n1 + 63) / 64 * 64;   // memset a multiple of 256 B: one fill kernel, not two
  int* cnt = work;                 // [N+1] (padded)
  int* local = cnt + n1p;          // [N+1] (padded)
  int* btot = local + n1p;         // [nblk]
  int* src32 = btot + nblk + 1;
  int* dst32 = src32 + E;
  int* rank = dst32 + E;
  int* items = rank + E;
  // small graphs: everything in ONE launch (k_prep_small), no counters to zero
  const int node_blocks = extra ? yl_cdiv(extra->N, 8 * PS_NW * PS_NODE_ITERS) : 0;
  const int seg_blocks = bbox_idx ? yl_cdiv(N + 1, PS_T) : 0;
  const int Rs = (extra == nullptr || yl_node3_smallk_shape_ok(*extra)) ? prep_small_rows(N, E, node_blocks + seg_blocks) : 0;
  if (Rs > 0) {
    PrepSmall g;
    g.edge = edge; g.se = (long)stride_e; g.sc = (long)stride_c; g.attr = reinterpret_cast<const float4*>(e_attr);
    g.bbox = bbox_idx; g.P = (long)P; g.E = (int)E; g.N = (int)N; g.R = Rs; g.nprep = yl_cdiv(N, Rs);
    g.nseg = seg_blocks;
    g.row_ptr = row_ptr; g.perm = perm; g.src_csr = src_csr; g.dst_csr = dst_csr;
    g.attr_csr = reinterpret_cast<float4*>(attr_csr); g.seg_ptr = seg_ptr; g.node_seg = node_seg; g.status = status;
    NodeUv none{};
    hipLaunchKernelGGL(k_prep_small, dim3(g.nprep + g.nseg + node_blocks), dim3(PS_T), 0, st, g, extra ? *extra : none,
                       extra ? 1 : 0);
    YL_LAUNCH_CHECK();
    return 0;
  }
  // primed: the caller vouches that the counters are zero (left so by the previous call with this shape on this buffer)
  // (a kernel, not hipMemsetAsync: the eval plan may be captured into a hipGraph, and ROCm 7.2's memset graph node
  // faulted — "write access to a read-only page", gigabytes away from every buffer of the graph — after a few dozen
  // replays of a graph that was otherwise all kernel nodes; tools/exp/graphs_probe2.py)
  if (!primed) {
    hipLaunchKernelGGL(k_zero_i32, dim3(yl_cdiv(n1p, 1024)), dim3(256), 0, st, cnt, n1p);
    YL_LAUNCH_CHECK();
  }
  const long nthreads = (E > N + 1) ? E : N + 1;
  hipLaunchKernelGGL(k_prep_count_win, dim3(yl_cdiv(nthreads, PCW_E)), dim3(PCW_T), 0, st, edge, (long)stride_e,
                     (long)stride_c, (int)E, (int)N, src32, dst32, rank, cnt, bbox_idx, (long)P, seg_ptr,
                     node_seg, status);
  YL_LAUNCH_CHECK();
  hipLaunchKernelGGL(k_prep_scan, dim3(nblk), dim3(1024), 0, st, cnt, n1, local, btot);
  YL_LAUNCH_CHECK();
  if (E > 0) {
    hipLaunchKernelGGL(k_prep_fill, dim3(yl_cdiv(E, 256)), dim3(256), 0, st, dst32, rank, (int)E, local, btot,
                       items);
    YL_LAUNCH_CHECK();
  }
  const int rows_blocks = yl_cdiv(E > n1 ? E : n1, 256);
  if (extra != nullptr && yl_node3_smallk_ok(*extra)) {
    // large graph: the K = in_channels node side is an output stream of its own (common.hpp node3_smallk_body), not tiles
    // of this launch.  (Riding in the COUNTING launch instead, interleaved with its blocks, was measured: the stream slows
    // the returning atomics down — fp32 38.8 + 51 us as two launches, 109 us as one; bf16 34.6 + 43.5 vs 75.)
    hipLaunchKernelGGL(k_prep_rows, dim3(rows_blocks), dim3(256), 0, st, local, btot, (int)N, (int)E, items, src32, dst32,
                       reinterpret_cast<const float4*>(e_attr), row_ptr, perm, src_csr, dst_csr,
                       reinterpret_cast<float4*>(attr_csr));
    YL_LAUNCH_CHECK();
    return yl_node3_smallk(*extra, st);
  }
  if (extra != nullptr) {
    const unsigned total = (unsigned)rows_blocks + 4u * (unsigned)yl_cdiv(extra->N, 64);
    if (extra->Cin <= 16)
      hipLaunchKernelGGL(k_prep_rows_node3<16>, dim3(total), dim3(256), 0, st, local, btot, (int)N, (int)E, items, src32, dst32,
                         reinterpret_cast<const float4*>(e_attr), row_ptr, perm, src_csr, dst_csr,
                         reinterpret_cast<float4*>(attr_csr), rows_blocks, *extra);
    else
      hipLaunchKernelGGL(k_prep_rows_node3<32>, dim3(total), dim3(256), 0, st, local, btot, (int)N, (int)E, items, src32, dst32,
                         reinterpret_cast<const float4*>(e_attr), row_ptr, perm, src_csr, dst_csr,
                         reinterpret_cast<float4*>(attr_csr), rows_blocks, *extra);
  } else {
    hipLaunchKernelGGL(k_prep_rows, dim3(rows_blocks), dim3(256), 0, st, local, btot, (int)N, (int)E, items, src32, dst32,
                       reinterpret_cast<const float4*>(e_attr), row_ptr, perm, src_csr, dst_csr,
                       reinterpret_cast<float4*>(attr_csr));
  }
  YL_LAUNCH_CHECK();
  return 0;
}

extern "C" int yolat_graph_prepare(const int64_t* edge, int64_t stride_e, int64_t stride_c,
                                   const float* e_attr, const int64_t* bbox_idx, int64_t E, int64_t N,
                                   int64_t P, int32_t* row_ptr, int32_t* perm, int32_t* src_csr,
                                   int32_t* dst_csr, float* attr_csr, int32_t* seg_ptr, int32_t* node_seg,
                                   int32_t* work, int32_t* status, yolat_stream_t stream) {
  return yl_graph_prepare_impl(edge, stride_e, stride_c, e_attr, bbox_idx, E, N, P, row_ptr, perm, src_csr, dst_csr,
                            attr_csr, seg_ptr, node_seg, work, status, nullptr, false, stream);
}

// yolat_graph_prepare with the node side of the first conv layer (yolat_node_uv_eval on the raw node
// features, independent of the graph) riding in its last launch.
extern "C" int yolat_graph_prepare_node_uv(const int64_t* edge, int64_t stride_e, int64_t stride_c,
                                           const float* e_attr, const int64_t* bbox_idx, int64_t E, int64_t N,
                                           int64_t P, int32_t* row_ptr, int32_t* perm, int32_t* src_csr,
                                           int32_t* dst_csr, float* attr_csr, int32_t* seg_ptr, int32_t* node_seg,
                                           int32_t* work, int32_t* status, const float* x, int64_t ldx, int64_t Cin,
                                           const float* Wuv, const float* uv_bias, const float* Wr, const float* br,
                                           const float* Wn,
                                           const float* bn, const float* sn, const float* tn, int64_t C, float* UV,
                                           int64_t ld_uv, float* f_out, int64_t ld_fo, float* s_out, int64_t ld_so,
                                           yolat_stream_t stream) {
  NodeUv a;
  const int rc = yl_build_node_uv(&a, x, ldx, x, ldx, N, Cin, Wuv, uv_bias, Wr, br, Wn, bn, sn, tn, C, UV, ld_uv, f_out,
                                  ld_fo, s_out, ld_so);
  if (rc != 0) return rc;
  return yl_graph_prepare_impl(edge, stride_e, stride_c, e_attr, bbox_idx, E, N, P, row_ptr, perm, src_csr, dst_csr,
                            attr_csr, seg_ptr, node_seg, work, status, &a, false, stream);
}

__global__ void k_gather_rows(const float* src, long ld_src, const int* idx, long rows, int width,
                              float* dst, long ld_dst) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const long r = i / width;
  const int c = (int)(i % width);
  if (r >= rows) return;
  dst[r * ld_dst + c] = src[(long)idx[r] * ld_src + c];
}

extern "C" int yolat_gather_rows(const float* src, int64_t ld_src, const int32_t* idx,
                                 int64_t rows, int64_t width, float* dst, int64_t ld_dst,
                                 yolat_stream_t stream) {
  if (rows < 0 || width <= 0) return YOLAT_E_INVALID;
  if (rows == 0) return 0;
  if (!src || !idx || !dst) return YOLAT_E_INVALID;
  hipLaunchKernelGGL(k_gather_rows, dim3(yl_cdiv(rows * width, 256)), dim3(256), 0,
                     (hipStream_t)stream, src, (long)ld_src, idx, (long)rows, (int)width, dst,
                     (long)ld_dst);
  YL_LAUNCH_CHECK();
  return 0;
}
