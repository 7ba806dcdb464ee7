// segmax.hpp — run-length per-proposal max epilogue shared by the rows kernels (fusion_x6.hip: bf16x6-emulated fp32,
// fusion_h8.hip: bf16 storage).  Replaces torch_scatter.scatter(reduce='max') of architecture3cc_rpn_gp_iter2.py:122 on
// the fusion block's activations, which are never written to memory.
#pragma once
#include "common.hpp"
#ifndef FX_EPI_STAMP
#define FX_EPI_STAMP(k) do { } while (0)      // debug builds of fusion_x6.hip define it (tools/exp/r06_fx_stamps.sh)
#endif

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
// The same with the 32-bit element offset of every row's proposal (id x stride) kept in 16 registers: the flush block of
// the direct-atomics epilogue then needs no LDS read (and no wait for it) — round 6, the fp32 rows kernel has the registers
// since its training epilogue shrank.
__device__ __forceinline__ void fx_seg_runs_off(int segv, int lhi, unsigned stride, FxRuns& sr, unsigned off[16]) {
  int sgs[16];
#pragma unroll
  for (int r = 0; r < 16; ++r) sgs[r] = __shfl(segv, (r & 3) + 8 * (r >> 2) + 4 * lhi);
  unsigned fb = 0, uf = 0;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    off[r] = (unsigned)(sgs[r] < 0 ? 0 : sgs[r]) * stride;
    const bool fl = sgs[r] >= 0 && (r == 15 || sgs[r] != sgs[r + 1]);
    fb |= fl ? (1u << r) : 0u;
    uf |= (__builtin_amdgcn_ballot_w64(fl) != 0ull) ? (1u << r) : 0u;
  }
  sr.flush_bits = fb; sr.uflush = uf;
}
// Direct-atomics epilogue on those offsets.  The running maximum is an INTEGER max on the float bits: cur >= 0 always (it
// starts at 0 = the ReLU), so a negative value (sign bit: a negative integer) never wins and non-negative floats order like
// their bits — one v_max_i32 per element where fmaxf costs three instructions (both operands are canonicalised first).
// A positive NaN wins (as it does in the integer atomicMax that follows, and in torch's max); fmaxf dropped it.
__device__ __forceinline__ void fx_segmax2_off(const f32x16& acc0, const f32x16& acc1, float* pool, const unsigned off[16],
                                               unsigned c0, bool ok0, bool ok1, const FxRuns& sr) {
  int cur0 = 0, cur1 = 0;
  FX_EPI_STAMP(0);
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    cur0 = yl_max(cur0, __float_as_int(acc0[r]));
    cur1 = yl_max(cur1, __float_as_int(acc1[r]));
    if ((sr.uflush >> r) & 1u) {
      if ((sr.flush_bits >> r) & 1u) {
        int* o = reinterpret_cast<int*>(pool) + (off[r] + c0);
        if (ok0 && cur0 > 0) atomicMax(o, cur0);
        if (ok1 && cur1 > 0) atomicMax(o + 32, cur1);
        cur0 = 0; cur1 = 0;                              // the lane's next row starts a new run
      }
    }
    if ((r & 3) == 3) FX_EPI_STAMP(1 + (r >> 2));
  }
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
// Round 6: the same with the table offset of every row's proposal (position x 64; >= FX_NP * 64: the proposal is
// beyond the table) in 16 registers and the running maximum as an integer max on the float bits (see fx_segmax2_off): per
// element one instruction instead of three, per flush no LDS read of the proposal id and no wait for it.  Stamps of the
// fp32 kernel (tools/exp/r06_fx_stamps.py) price this walk at ~1 us per column tile and wave — twice the bf16 kernel's
// 16 MFMAs per tile.
__device__ __forceinline__ void fx_seg_runs_tab(int segv, int lhi, int seg_base, FxRuns& sr, unsigned toff[16]) {
  int sgs[16];
#pragma unroll
  for (int r = 0; r < 16; ++r) sgs[r] = __shfl(segv, (r & 3) + 8 * (r >> 2) + 4 * lhi);
  unsigned fb = 0, uf = 0;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const unsigned pos = (unsigned)(sgs[r] - seg_base);
    toff[r] = (sgs[r] >= 0 && pos < (unsigned)FX_NP) ? pos * 64u : 0xFFFFFFFFu;
    const bool fl = sgs[r] >= 0 && (r == 15 || sgs[r] != sgs[r + 1]);
    fb |= fl ? (1u << r) : 0u;
    uf |= (__builtin_amdgcn_ballot_w64(fl) != 0ull) ? (1u << r) : 0u;
  }
  sr.flush_bits = fb; sr.uflush = uf;
}
__device__ __forceinline__ void fx_segmax2_lds_off(const f32x16& acc0, const f32x16& acc1, int* tab, float* pool,
                                                   unsigned ldpool, const int* segs, const unsigned toff[16], int lhi,
                                                   unsigned c0, unsigned cl, bool ok0, bool ok1, const FxRuns& sr) {
  int cur0 = 0, cur1 = 0;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    cur0 = yl_max(cur0, __float_as_int(acc0[r]));
    cur1 = yl_max(cur1, __float_as_int(acc1[r]));
    if ((sr.uflush >> r) & 1u) {
      if ((sr.flush_bits >> r) & 1u) {
        if (toff[r] != 0xFFFFFFFFu) {
          if (ok0 && cur0 > 0) atomicMax(tab + toff[r] + cl, cur0);
          if (ok1 && cur1 > 0) atomicMax(tab + toff[r] + 32 + cl, cur1);
        } else {
          int* o = reinterpret_cast<int*>(pool) + ((unsigned)segs[(r & 3) + 8 * (r >> 2) + 4 * lhi] * ldpool + c0);
          if (ok0 && cur0 > 0) atomicMax(o, cur0);
          if (ok1 && cur1 > 0) atomicMax(o + 32, cur1);
        }
        cur0 = 0; cur1 = 0;
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
