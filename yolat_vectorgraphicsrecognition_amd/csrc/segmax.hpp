// segmax.hpp — run-length per-proposal max epilogue shared by the rows kernels (fusion_x6.hip: bf16x6-emulated fp32,
// fusion_h8.hip: bf16 storage).  Replaces torch_scatter.scatter(reduce='max') of architecture3cc_rpn_gp_iter2.py:122 on
// the fusion block's activations, which are never written to memory.
#pragma once
#include "common.hpp"

namespace {
// Run structure of a lane's 16 rows (C/D layout of a 32x32 MFMA tile) from their proposal ids — the same for every
// column tile: flush bit r = a run ends at row r (uflush: in some lane of the wave).  A run of row r+1 starts exactly
// where a run ended at row r, so the running maximum is reset inside the flush block (round 6: one v_max per element
// instead of a multiply by a keep flag + v_max3, and 16 registers fewer).  The proposal id of a flushed row is re-read
// from LDS (segs: the wave's 32 ids) — rare, and 16 registers cheaper than keeping the offsets.
struct FxRuns { unsigned flush_bits, uflush; };
__device__ __forceinline__ void fx_seg_runs(int segv, int lhi, FxRuns& sr) {
  int sgs[16];
#pragma unroll
  for (int r = 0; r < 16; ++r) sgs[r] = __shfl(segv, (r & 3) + 8 * (r >> 2) + 4 * lhi);
  unsigned fb = 0, uf = 0;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const bool fl = sgs[r] >= 0 && (r == 15 || sgs[r] != sgs[r + 1]);
    fb |= fl ? (1u << r) : 0u;
    uf |= (__builtin_amdgcn_ballot_w64(fl) != 0ull) ? (1u << r) : 0u;
  }
  sr.flush_bits = fb; sr.uflush = uf;
}
// values = relu(acc) (the shift is the accumulator's initial value, the scale is inside the weights).  Both column
// blocks in one pass: two independent running-max chains (the chain is latency bound) and one test per row.
__device__ __forceinline__ void fx_segmax2(const f32x16& acc0, const f32x16& acc1, float* pool, unsigned ldpool,
                                           const int* segs, int lhi, unsigned c0, bool ok0, bool ok1, const FxRuns& sr) {
  float cur0 = 0.f, cur1 = 0.f;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    cur0 = fmaxf(cur0, acc0[r]);                       // cur >= 0 always: the ReLU is the start value
    cur1 = fmaxf(cur1, acc1[r]);
    if ((sr.uflush >> r) & 1u) {
      if ((sr.flush_bits >> r) & 1u) {
        int* o = reinterpret_cast<int*>(pool) + ((unsigned)segs[(r & 3) + 8 * (r >> 2) + 4 * lhi] * ldpool + c0);
        if (ok0 && cur0 > 0.f) atomicMax(o, __float_as_int(cur0));
        if (ok1 && cur1 > 0.f) atomicMax(o + 32, __float_as_int(cur1));
        cur0 = 0.f; cur1 = 0.f;                          // the lane's next row starts a new run
      }
    }
  }
}

// ---- workgroup-level pre-reduction in LDS (round 3).  The run-length epilogue above issues one global atomicMax per
// (run, column, lane half): a 256-row tile that holds ~10 proposals sends ~26 per column (8 waves x 2 halves x 1.6 runs)
// — PMC at cfg 5: 163 MB written for a 65 MB pooled matrix, and the L2's atomic rate, not the matrix cores, set the
// kernel's time.  Here the runs of all 8 waves meet in an LDS table tab[FX_NP][64] (integer ds_max on the float bits,
// exact and order-independent like the global one) indexed by the proposal's position inside the tile; the table of a
// column tile is complete at the tile's closing barrier and is drained by all threads while the NEXT tile's MFMAs run
// (two tables, alternating: no extra barrier).  A proposal that lies wholly inside the tile's rows is written with a
// plain store — no other workgroup touches those elements — and only the (at most two) proposals shared with the
// neighbouring row tiles still use a global atomicMax.  Proposals beyond position FX_NP (tiles of very small proposals)
// keep the direct global atomics.
constexpr int FX_NP = 32;
struct FxTile {              // wave-uniform description of the workgroup's 256-row tile
  int seg_base;              // proposal id of the first row
  int np;                    // proposals reduced in LDS: min(#proposals in the tile, FX_NP)
  int first_shared, last_shared_pos;   // first proposal continues from the previous tile; position of a last proposal
                                       // that continues into the next tile (-1: none)
};
__device__ __forceinline__ FxTile fx_tile(const int* __restrict__ seg, int row_lo, int row_hi /* excl., <= N */, int N) {
  FxTile t;
  t.seg_base = seg[row_lo];
  const int seg_last = seg[row_hi - 1];
  const int npos = seg_last - t.seg_base + 1;
  t.np = npos < FX_NP ? npos : FX_NP;
  t.first_shared = (row_lo > 0 && seg[row_lo - 1] == t.seg_base) ? 1 : 0;
  t.last_shared_pos = (row_hi < N && seg[row_hi] == seg_last) ? npos - 1 : -1;
  return t;
}
__device__ __forceinline__ void fx_segmax2_lds(const f32x16& acc0, const f32x16& acc1, int* tab, float* pool, unsigned ldpool,
                                               const int* segs, int seg_base, int lhi, unsigned c0, unsigned cl, bool ok0,
                                               bool ok1, const FxRuns& sr) {
  float cur0 = 0.f, cur1 = 0.f;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    cur0 = fmaxf(cur0, acc0[r]);
    cur1 = fmaxf(cur1, acc1[r]);
    if ((sr.uflush >> r) & 1u) {
      if ((sr.flush_bits >> r) & 1u) {
        const int sg = segs[(r & 3) + 8 * (r >> 2) + 4 * lhi];
        const unsigned pos = (unsigned)(sg - seg_base);
        if (pos < (unsigned)FX_NP) {
          if (ok0 && cur0 > 0.f) atomicMax(tab + pos * 64 + cl, __float_as_int(cur0));
          if (ok1 && cur1 > 0.f) atomicMax(tab + pos * 64 + 32 + cl, __float_as_int(cur1));
        } else {
          int* o = reinterpret_cast<int*>(pool) + ((unsigned)sg * ldpool + c0);
          if (ok0 && cur0 > 0.f) atomicMax(o, __float_as_int(cur0));
          if (ok1 && cur1 > 0.f) atomicMax(o + 32, __float_as_int(cur1));
        }
        cur0 = 0.f; cur1 = 0.f;
      }
    }
  }
}
// drain the table of column tile `ct` (64 columns from 64 ct) into the pooled matrix and clear it
__device__ __forceinline__ void fx_tab_drain(int* tab, const FxTile& t, float* pool, unsigned ldpool, int ct, int F, int tid,
                                             int nthreads) {
  for (int e = tid; e < t.np * 64; e += nthreads) {
    const int v = tab[e];
    if (v != 0) {
      tab[e] = 0;
      const int pos = e >> 6, col = ct * 64 + (e & 63);
      if (col < F) {
        int* o = reinterpret_cast<int*>(pool) + ((unsigned)(t.seg_base + pos) * ldpool + (unsigned)col);
        if ((pos == 0 && t.first_shared) || pos == t.last_shared_pos) atomicMax(o, v);
        else *o = v;
      }
    }
  }
}
}  // namespace
