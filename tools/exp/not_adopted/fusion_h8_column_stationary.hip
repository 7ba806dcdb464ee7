// fusion_h8.hip — eval fusion block + per-proposal max (architecture3cc_rpn_gp_iter2.py:61-63,122) and
// fusion_block_super (arch:65-69) for the bf16-storage forward, as ONE launch of an A-in-registers rows kernel (round 3).
//
// What it replaces and why (profiles/r01_fwd_cfg5_bf16_kernel_stats_v1.txt, k_hfusion_rows<1,128>: 158 us for 52 GFLOP =
// 0.14 of the dense bf16 peak): that kernel keeps a 64-row tile of `feats` in LDS and re-streams every 64-column tile of
// the weights per 64 rows — 800 MB of L2 -> LDS traffic at N = 200 k for a 256 KB weight — and reads 16 KB of LDS
// fragments for every 8 MFMAs of a wave between two barriers.  Here (the structure of fusion_x6.hip's fp32 kernel, one
// bf16 part instead of three): a 512-thread workgroup owns 256 rows; every wave keeps ITS 32 rows x K of A as MFMA
// fragments in 32 registers for all the column tiles it walks; the weights — BatchNorm scale folded into their rows
// BEFORE the bf16 rounding, the shift is the accumulators' initial value — stream through a double-buffered LDS tile
// (64 columns x K), the next tile's 16-byte pieces in registers while the current tile's MFMAs and pooling epilogue run;
// one barrier per column tile; W is streamed once per 256 rows (200 MB).  The pooling epilogue is the run-length integer
// atomicMax of segmax.hpp (values >= 0: integer order == float order, exact, order-independent).
// The small problem (fusion_block_super on the P per-proposal means, fp32 rows converted while loading, plain ReLU
// store) rides in the same launch, its workgroups first.
#include "segmax.hpp"
#include <stdlib.h>

typedef unsigned short u16;
typedef unsigned h8_u32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 h8_bf16x8 __attribute__((ext_vector_type(8)));

#ifdef YOLAT_H8_STAMPS
// debug build only (tools/exp/r06_fx_stamps.sh <cfg> h8): wall-clock stamps (100 MHz) of thread 0 of every workgroup
__device__ long long h8_stamps_d[4096 * 32];
#define H8_STAMP(k) do { if (KD == 128 && threadIdx.x == 0 && blockIdx.x < 4096) h8_stamps_d[blockIdx.x * 32 + (k)] = wall_clock64(); } while (0)
#define H8_STAMP_META()                                                          \
  do {                                                                           \
    if (KD == 128 && threadIdx.x == 0 && blockIdx.x < 4096) {                    \
      h8_stamps_d[blockIdx.x * 32 + 21] = (long long)ngl;                        \
      h8_stamps_d[blockIdx.x * 32 + 22] = (long long)(small ? 1 : 0);            \
      h8_stamps_d[blockIdx.x * 32 + 23] = 0;                                     \
    }                                                                            \
  } while (0)
extern "C" int yolat_debug_h8_stamps(long long* out, int n) {
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(h8_stamps_d), sizeof(long long) * (size_t)n);
}
#else
#define H8_STAMP(k) do { } while (0)
#define H8_STAMP_META() do { } while (0)
#endif
struct H8Prob {
  const u16* Ah;        // bf16 rows (NULL -> Af)
  const float* Af;      // fp32 rows, converted while loading
  long lda; int N;
  const u16* W;         // [F, KD] bf16, BatchNorm scale folded into the rows
  const float* tfold;   // [F] shift (s*b + t)
  const int* seg;       // != NULL: pooling epilogue into out (pooled matrix); else out[row, col] = relu(.)
  float* out; long ldo;
  int F, tm, groups, ng;
};

template <int KD>
__global__ void __launch_bounds__(512, 2) k_hfusion_rows8(H8Prob p0, H8Prob p1) {
  constexpr int KS = KD / 16, RS = KD + 8, CPR = KD / 8;    // k steps, LDS row stride (bf16), 16-byte chunks per row
  constexpr int NW = 64 * CPR / 512;                        // 16-byte pieces of one W tile per thread
  static_assert((64 * CPR) % 512 == 0, "W tile / thread mismatch");
  __shared__ __attribute__((aligned(16))) u16 Ws[2][64 * RS];
  __shared__ int seg_s[256];
  __shared__ int tab_s[2][FX_NP * 64];                     // per-column-tile pooled maxima of the workgroup (segmax.hpp)
  const int tid = threadIdx.x;
  const int n1 = p1.tm * p1.groups, n1p = (n1 + 7) & ~7;
  const int id = blockIdx.x;
  int logical;
  if (id < n1p) {
    if (id >= n1) return;
    logical = id;
  } else {
    // (row tile, column group) pairs, column group fastest, dealt to the XCDs in contiguous ranges
    const int n0 = p0.tm * p0.groups, j0 = id - n1p;
    const int chunk = n0 >> 3, rem = n0 & 7;
    const int xcd = j0 & 7, slot = j0 >> 3;
    logical = xcd * chunk + (xcd < rem ? xcd : rem) + slot;
  }
  const bool small = id < n1p;
  const u16* const Ah = small ? p1.Ah : p0.Ah;
  const float* const Af = small ? p1.Af : p0.Af;
  const long lda = small ? p1.lda : p0.lda;
  const int N = small ? p1.N : p0.N, F = small ? p1.F : p0.F;
  const u16* const W = small ? p1.W : p0.W;
  const float* const tfold = small ? p1.tfold : p0.tfold;
  const int* const seg = small ? p1.seg : p0.seg;
  float* const out = small ? p1.out : p0.out;
  const long ldo = small ? p1.ldo : p0.ldo;
  const int groups = small ? p1.groups : p0.groups, ng = small ? p1.ng : p0.ng;
  const int rt = logical / groups, cg = logical % groups;
  const int tn = (F + 63) >> 6;
  const int ct0 = cg * ng;
  const int ngl = yl_min(ng, tn - ct0);
  if (ngl <= 0) return;

  const int wave = tid >> 6, lane = tid & 63, l31 = lane & 31, lhi = lane >> 5;
  const int row0 = rt * 256 + wave * 32;
  H8_STAMP(0);
  // ---- this wave's 32 rows of A as MFMA A fragments (lane = row, 8 consecutive k per lane half)
  h8_bf16x8 Afr[KS];
  {
    const long r = yl_min(row0 + l31, N - 1);
    if (Ah != nullptr) {
      const u16* ap = Ah + r * lda + 8 * lhi;
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) Afr[ks] = __builtin_bit_cast(h8_bf16x8, *reinterpret_cast<const h8_u32x4*>(ap + 16 * ks));
    } else {
      const float* ap = Af + r * lda + 8 * lhi;
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        const float4 a0 = *reinterpret_cast<const float4*>(ap + 16 * ks), a1 = *reinterpret_cast<const float4*>(ap + 16 * ks + 4);
        const h8_u32x4 v = {yl_pack_bf16(a0.x, a0.y), yl_pack_bf16(a0.z, a0.w), yl_pack_bf16(a1.x, a1.y),
                            yl_pack_bf16(a1.z, a1.w)};
        Afr[ks] = __builtin_bit_cast(h8_bf16x8, v);
      }
    }
  }
  H8_STAMP(1);
  const bool pooling = seg != nullptr;
  FxTile tile{};
  if (pooling) {
    const int row_lo = rt * 256, row_hi = yl_min(row_lo + 256, N);
    tile = fx_tile(seg, row_lo, row_hi, N);
    for (int e = tid; e < 2 * FX_NP * 64; e += 512) (&tab_s[0][0])[e] = 0;
  }
  FxRuns runs;
  {
    const int sv = (pooling && row0 + l31 < N) ? seg[row0 + l31] : -1;
    if (lhi == 0) seg_s[wave * 32 + l31] = sv;          // read back by the same wave only, after the barrier below
    fx_seg_runs(sv, lhi, runs);
  }
  const int* segs = seg_s + wave * 32;
  const unsigned wr0 = (unsigned)tid / CPR, wk = ((unsigned)tid % CPR) * 8;   // row (of the first piece), k offset
  auto load_w = [&](int ct, h8_u32x4* rw) {
#pragma unroll
    for (int t = 0; t < NW; ++t) {
      const unsigned r = wr0 + (unsigned)t * (512 / CPR);
      rw[t] = *reinterpret_cast<const h8_u32x4*>(W + (unsigned)yl_min(ct * 64 + (int)r, F - 1) * KD + wk);
    }
  };
  auto store_w = [&](int buf, const h8_u32x4* rw) {
#pragma unroll
    for (int t = 0; t < NW; ++t) {
      const unsigned r = wr0 + (unsigned)t * (512 / CPR);
      *reinterpret_cast<h8_u32x4*>(&Ws[buf][r * RS + wk]) = rw[t];
    }
  };
  h8_u32x4 rw[NW];
  load_w(ct0, rw);
  float t0 = tfold[yl_min(ct0 * 64 + l31, F - 1)], t1 = tfold[yl_min(ct0 * 64 + 32 + l31, F - 1)];
  H8_STAMP(2);
  store_w(0, rw);
  __syncthreads();
  H8_STAMP(3);
  for (int j = 0; j < ngl; ++j) {
    const int ct = ct0 + j, buf = j & 1;
    const int c0 = ct * 64 + l31, c1 = c0 + 32;
    f32x16 acc0, acc1;
#pragma unroll
    for (int r = 0; r < 16; ++r) { acc0[r] = t0; acc1[r] = t1; }
    if (j + 1 < ngl) {
      t0 = tfold[yl_min(c0 + 64, F - 1)];
      t1 = tfold[yl_min(c1 + 64, F - 1)];
      load_w(ct + 1, rw);                              // in flight while the MFMAs below run
    }
    // the previous column tile's pooled maxima (complete since its closing barrier) go out while this tile's MFMAs run
    if (pooling && j > 0) fx_tab_drain(tab_s[buf ^ 1], tile, out, (unsigned)ldo, ct - 1, F, tid, 512);
    if (j >= 4 && j < 8) H8_STAMP(24 + (j - 4));        // after the drain of tiles 4..7
    const u16* wb = &Ws[buf][l31 * RS + 8 * lhi];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      const h8_bf16x8 b0 = *reinterpret_cast<const h8_bf16x8*>(wb + 16 * ks);
      const h8_bf16x8 b1 = *reinterpret_cast<const h8_bf16x8*>(wb + 32 * RS + 16 * ks);
      acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Afr[ks], b0, acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Afr[ks], b1, acc1, 0, 0, 0);
    }
    // next W tile into the other buffer (its readers finished before the last barrier) BEFORE the epilogue: the wait
    // for its loads then never includes the epilogue's atomics / stores (vmcnt retires in order)
    if (j >= 4 && j < 8) H8_STAMP(4 + 4 * (j - 4));
    if (j + 1 < ngl) store_w(buf ^ 1, rw);
    __builtin_amdgcn_s_waitcnt(0x0F70);                // vmcnt(0)
    if (j >= 4 && j < 8) H8_STAMP(5 + 4 * (j - 4));
    if (pooling) {
      fx_segmax2_lds(acc0, acc1, tab_s[buf], out, (unsigned)ldo, segs, tile.seg_base, lhi, (unsigned)c0, (unsigned)l31, c0 < F,
                     c1 < F, runs);
    } else {
      unsigned rb = (unsigned)(row0 + 4 * lhi);
      asm volatile("" : "+v"(rb));
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const unsigned row = rb + (r & 3) + 8 * (r >> 2);
        if ((int)row < N) {
          float* o = out + (unsigned long)row * (unsigned long)ldo;
          if (c0 < F) o[c0] = fmaxf(acc0[r], 0.f);
          if (c1 < F) o[c1] = fmaxf(acc1[r], 0.f);
        }
      }
    }
    if (j >= 4 && j < 8) H8_STAMP(6 + 4 * (j - 4));
    __syncthreads();
    if (j >= 3 && j < 8) H8_STAMP(j == 3 ? 28 : 7 + 4 * (j - 4));
  }
  H8_STAMP(20);
  H8_STAMP_META();
  if (pooling) fx_tab_drain(tab_s[(ngl - 1) & 1], tile, out, (unsigned)ldo, ct0 + ngl - 1, F, tid, 512);
}

// ---------------------------------------------------------------------------------------------------------------------
// Round 6: the same launch "column-stationary".  Stamps of k_hfusion_rows8 (profiles/r06_h8_stamps.txt): a column tile
// costs a workgroup 2.2 us, 0.48 of them with MFMAs in flight; the rest are phases that each start with an LDS access — the
// table drain, the W tile's store, the walk's table atomics — queued behind the OTHER workgroup's 128 KB of B-fragment
// reads per tile, and a barrier per tile keeps the eight waves in lock step.  Here the roles of the operands are swapped:
//   * the workgroup's 256 rows of A are staged ONCE in LDS (16-byte pieces, whole rows per instruction; 68 KB), one barrier;
//   * a WAVE owns a 64-column tile: its 16 B fragments (64 registers) come straight from the L2-resident weights, and it
//     walks the tile's eight 32-row blocks — one A fragment read (1 KB) per TWO MFMAs instead of one per MFMA, no W tile in
//     LDS, no further barrier: the waves drift apart and one's walk runs under another's MFMAs;
//   * the wave sees ALL rows of its columns in order, so the running maximum carries across the row blocks: one flush per
//     proposal and lane half (global integer atomicMax on the float bits: exact, order-independent), no table, no drain.
// Column tiles ct0 + wave, ct0 + wave + 8, ...: used when a workgroup's column group holds a multiple of eight tiles.
template <int KD, int OCC>
__global__ void __launch_bounds__(512, OCC) k_hfusion_cols8(H8Prob p0, H8Prob p1) {
  constexpr int KS = KD / 16, RS = KD + 8, CPR = KD / 8;    // k steps, LDS row stride (bf16), 16-byte chunks per row
  __shared__ __attribute__((aligned(16))) u16 As[256 * RS];
  __shared__ int seg_s[256 + 32];                          // proposal id per row (-1: no row / no pooling); + a guard block
  const int tid = threadIdx.x;
  const int n1 = p1.tm * p1.groups, n1p = (n1 + 7) & ~7;
  const int id = blockIdx.x;
  int logical;
  if (id < n1p) {
    if (id >= n1) return;
    logical = id;
  } else {
    const int n0 = p0.tm * p0.groups, j0 = id - n1p;
    const int chunk = n0 >> 3, rem = n0 & 7;
    const int xcd = j0 & 7, slot = j0 >> 3;
    logical = xcd * chunk + (xcd < rem ? xcd : rem) + slot;
  }
  const bool small = id < n1p;
  const u16* const Ah = small ? p1.Ah : p0.Ah;
  const float* const Af = small ? p1.Af : p0.Af;
  const long lda = small ? p1.lda : p0.lda;
  const int N = small ? p1.N : p0.N, F = small ? p1.F : p0.F;
  const u16* const W = small ? p1.W : p0.W;
  const float* const tfold = small ? p1.tfold : p0.tfold;
  const int* const seg = small ? p1.seg : p0.seg;
  float* const out = small ? p1.out : p0.out;
  const long ldo = small ? p1.ldo : p0.ldo;
  const int groups = small ? p1.groups : p0.groups, ng = small ? p1.ng : p0.ng;
  const int rt = logical / groups, cg = logical % groups;
  const int tn = (F + 63) >> 6;
  const int ct0 = cg * ng;
  const int ngl = yl_min(ng, tn - ct0);
  if (ngl <= 0) return;
  const int wave = tid >> 6, lane = tid & 63, l31 = lane & 31, lhi = lane >> 5;
  const int row_lo = rt * 256;
  const bool pooling = seg != nullptr;

  // ---- the first column tile's B fragments and shifts go out before the A staging: one round trip under it
  h8_bf16x8 B0[KS], B1[KS];
  auto load_b = [&](int ct) {
    const u16* w0 = W + (long)yl_min(ct * 64 + l31, F - 1) * KD + 8 * lhi;
    const u16* w1 = W + (long)yl_min(ct * 64 + 32 + l31, F - 1) * KD + 8 * lhi;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      B0[ks] = __builtin_bit_cast(h8_bf16x8, *reinterpret_cast<const h8_u32x4*>(w0 + 16 * ks));
      B1[ks] = __builtin_bit_cast(h8_bf16x8, *reinterpret_cast<const h8_u32x4*>(w1 + 16 * ks));
    }
  };
  int ct = ct0 + wave;
  float t0 = 0.f, t1 = 0.f;
  if (wave < ngl) {
    load_b(ct);
    t0 = tfold[yl_min(ct * 64 + l31, F - 1)];
    t1 = tfold[yl_min(ct * 64 + 32 + l31, F - 1)];
  }
  // ---- A tile -> LDS: piece p = (row p / CPR, 16-byte chunk p % CPR); 256 * CPR / 512 pieces per thread
  {
    constexpr int NP = 256 * CPR / 512;
    if (Ah != nullptr) {
      h8_u32x4 v[NP];
#pragma unroll
      for (int i = 0; i < NP; ++i) {
        const int pc = tid + 512 * i, r = pc / CPR, c = pc % CPR;
        v[i] = *reinterpret_cast<const h8_u32x4*>(Ah + (long)yl_min(row_lo + r, N - 1) * lda + 8 * c);
      }
#pragma unroll
      for (int i = 0; i < NP; ++i) {
        const int pc = tid + 512 * i, r = pc / CPR, c = pc % CPR;
        *reinterpret_cast<h8_u32x4*>(&As[r * RS + 8 * c]) = v[i];
      }
    } else {
#pragma unroll
      for (int i = 0; i < NP; ++i) {
        const int pc = tid + 512 * i, r = pc / CPR, c = pc % CPR;
        const float* ap = Af + (long)yl_min(row_lo + r, N - 1) * lda + 8 * c;
        const float4 a0 = *reinterpret_cast<const float4*>(ap), a1 = *reinterpret_cast<const float4*>(ap + 4);
        const h8_u32x4 v = {yl_pack_bf16(a0.x, a0.y), yl_pack_bf16(a0.z, a0.w), yl_pack_bf16(a1.x, a1.y),
                            yl_pack_bf16(a1.z, a1.w)};
        *reinterpret_cast<h8_u32x4*>(&As[r * RS + 8 * c]) = v;
      }
    }
    if (tid < 256 + 32) seg_s[tid] = (pooling && tid < 256 && row_lo + tid < N) ? seg[row_lo + tid] : -1;
  }
  __syncthreads();
  if (wave >= ngl) return;

  // ---- run structure of the lane's rows, block by block, the run carried ACROSS the blocks: flush bit r of block rb = a
  // run ends at the lane's row r of that block (its next row — the lane's first row of block rb + 1 for r = 15 — belongs to
  // another proposal or does not exist); uflush = in some lane of the wave
  // (16 bits per block, blocks 0-3 / 4-7 in two 64-bit words: the block loop below is a real loop — unrolled, the compiler
  // keeps an exec mask per (block, row): 182 spilled scalar registers)
  unsigned long long fb_lo = 0ull, fb_hi = 0ull, uf_lo = 0ull, uf_hi = 0ull;
  if (pooling) {
#pragma unroll
    for (int rb = 0; rb < 8; ++rb) {
      unsigned f = 0u, u = 0u;
      int sg = seg_s[rb * 32 + 4 * lhi];
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int nxt = (r < 15) ? seg_s[rb * 32 + ((r + 1) & 3) + 8 * ((r + 1) >> 2) + 4 * lhi]
                                 : seg_s[(rb + 1) * 32 + 4 * lhi];          // rb = 7: the guard block (-1)
        const bool fl = sg >= 0 && sg != nxt;
        f |= fl ? (1u << r) : 0u;
        u |= (__builtin_amdgcn_ballot_w64(fl) != 0ull) ? (1u << r) : 0u;
        sg = nxt;
      }
      if (rb < 4) { fb_lo |= (unsigned long long)f << (16 * rb); uf_lo |= (unsigned long long)u << (16 * rb); }
      else { fb_hi |= (unsigned long long)f << (16 * (rb - 4)); uf_hi |= (unsigned long long)u << (16 * (rb - 4)); }
    }
  }
  for (; ct < ct0 + ngl; ct += 8) {
    const int c0 = ct * 64 + l31, c1 = c0 + 32;
    const bool ok0 = c0 < F, ok1 = c1 < F;
    int cur0 = 0, cur1 = 0;                                // integer max on the float bits: cur >= 0 (the ReLU), see segmax.hpp
#pragma unroll 1
    for (int rb = 0; rb < 8; ++rb) {
      const unsigned fbits = (unsigned)((rb < 4 ? fb_lo : fb_hi) >> (16 * (rb & 3))) & 0xFFFFu;
      const unsigned ubits = (unsigned)__builtin_amdgcn_readfirstlane((int)((unsigned)((rb < 4 ? uf_lo : uf_hi) >> (16 * (rb & 3))) & 0xFFFFu));
      f32x16 acc0, acc1;
#pragma unroll
      for (int r = 0; r < 16; ++r) { acc0[r] = t0; acc1[r] = t1; }
      const u16* ab = &As[(rb * 32 + l31) * RS + 8 * lhi];
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        const h8_bf16x8 a = *reinterpret_cast<const h8_bf16x8*>(ab + 16 * ks);
        acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, B0[ks], acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, B1[ks], acc1, 0, 0, 0);
      }
      if (pooling) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          cur0 = yl_max(cur0, __float_as_int(acc0[r]));
          cur1 = yl_max(cur1, __float_as_int(acc1[r]));
          if ((ubits >> r) & 1u) {
            if ((fbits >> r) & 1u) {
              const int sgid = seg_s[rb * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhi];
              int* o = reinterpret_cast<int*>(out) + ((unsigned)sgid * (unsigned)ldo + (unsigned)c0);
              if (ok0 && cur0 > 0) atomicMax(o, cur0);
              if (ok1 && cur1 > 0) atomicMax(o + 32, cur1);
              cur0 = 0; cur1 = 0;
            }
          }
        }
      } else {
        unsigned rbase = (unsigned)(row_lo + rb * 32 + 4 * lhi);
        asm volatile("" : "+v"(rbase));
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const unsigned row = rbase + (r & 3) + 8 * (r >> 2);
          if ((int)row < N) {
            float* o = out + (unsigned long)row * (unsigned long)ldo;
            if (ok0) o[c0] = fmaxf(acc0[r], 0.f);
            if (ok1) o[c1] = fmaxf(acc1[r], 0.f);
          }
        }
      }
    }
    if (ct + 8 < ct0 + ngl) {                                // the wave's next column tile
      load_b(ct + 8);
      t0 = tfold[yl_min(c0 + 512, F - 1)];
      t1 = tfold[yl_min(c1 + 512, F - 1)];
    }
  }
}

// A [N, lda] bf16 x Wf [F, D] (folded) with the per-proposal max into pool [*, ld_pool] (columns 0..F), and
// As [P, lda_s] fp32 x Wfs [F, D] (folded) -> relu -> sup_out [P, ld_sup].  D in {64, 128}; F % 64 == 0.
int yl_hfusion_rows8(const uint16_t* A, int64_t lda, int64_t N, int64_t D, const uint16_t* Wf, const float* tf,
                     const int32_t* seg, float* pool, int64_t ld_pool, int64_t F, const float* As, int64_t lda_s, int64_t P,
                     const uint16_t* Wfs, const float* tfs, float* sup_out, int64_t ld_sup, hipStream_t st) {
  if ((D != 64 && D != 128) || F % 64 != 0 || N <= 0 || P <= 0) return YOLAT_E_UNSUPPORTED;
  if (!yl_aligned16(A) || !yl_aligned16(As) || !yl_aligned16(Wf) || !yl_aligned16(Wfs) || lda % 8 != 0 || lda_s % 4 != 0)
    return YOLAT_E_UNSUPPORTED;
  const int tn = (int)(F / 64);
  auto split = [&](long rows, int& tm, int& groups, int& ng, bool is_small) {
    // column groups by the cost model of fusion_x6.hip: rounds x (1 prologue + tiles per workgroup), 2 workgroups per CU
    tm = yl_cdiv(rows, 256);
    long best = -1;
    groups = 1;
    for (int g = 1; g <= tn; g *= 2) {
      const long wgs = (long)tm * g, rounds = (wgs + 511) / 512;
      const long cost = rounds * (2 + yl_cdiv(tn, g));
      if (best < 0 || cost < best) { best = cost; groups = g; }
    }
    if (is_small) {
      // the P-row problem's workgroups start first: few of them (each walking several column tiles) so that they do not
      // hold the first round of CUs back from the N-row problem (measured at cfg 5: 512 one-tile workgroups 104 us,
      // 32 sixteen-tile workgroups 93 us for the launch)
      groups = 1;
      while ((long)tm * groups * 2 <= 128 && groups * 2 <= tn) groups *= 2;
    }
    ng = yl_cdiv(tn, groups);
    groups = yl_cdiv(tn, ng);
  };
  H8Prob p0{}, p1{};
  p0.Ah = A; p0.Af = nullptr; p0.lda = lda; p0.N = (int)N; p0.W = Wf; p0.tfold = tf; p0.seg = seg; p0.out = pool;
  p0.ldo = ld_pool; p0.F = (int)F;
  split(N, p0.tm, p0.groups, p0.ng, false);
  p1.Ah = nullptr; p1.Af = As; p1.lda = lda_s; p1.N = (int)P; p1.W = Wfs; p1.tfold = tfs; p1.seg = nullptr; p1.out = sup_out;
  p1.ldo = ld_sup; p1.F = (int)F;
  split(P, p1.tm, p1.groups, p1.ng, true);
  const long n1p = ((long)p1.tm * p1.groups + 7) & ~7L;
  const long total = n1p + (long)p0.tm * p0.groups;
  if (total >= (1LL << 31)) return YOLAT_E_UNSUPPORTED;
  // column-stationary form (round 6) when every workgroup of both problems gets a multiple of eight column tiles and the
  // pooled matrix can be addressed with 32-bit element offsets
  static const int cols_off = getenv("YOLAT_H8_ROWS") ? 1 : 0;          // A/B switch (tools/exp)
  const char* ab = getenv("YOLAT_H8_AB");
  const bool cols = (ab ? ab[0] == 'c' : !cols_off) && D == 128 && p0.ng % 8 == 0 && tn % p0.ng == 0 &&
                    (long long)P * ld_pool < (1LL << 32);
  if (cols) {
    // the small problem in workgroups of eight column tiles
    p1.ng = 8; p1.groups = yl_cdiv(tn, 8);
    const long n1c = ((long)p1.tm * p1.groups + 7) & ~7L;
    const long totc = n1c + (long)p0.tm * p0.groups;
    if (totc >= (1LL << 31)) return YOLAT_E_UNSUPPORTED;
    if (ab && ab[1] == '4') hipLaunchKernelGGL((k_hfusion_cols8<128, 4>), dim3((unsigned)totc), dim3(512), 0, st, p0, p1);
    else hipLaunchKernelGGL((k_hfusion_cols8<128, 2>), dim3((unsigned)totc), dim3(512), 0, st, p0, p1);
  } else if (D == 128) hipLaunchKernelGGL(k_hfusion_rows8<128>, dim3((unsigned)total), dim3(512), 0, st, p0, p1);
  else hipLaunchKernelGGL(k_hfusion_rows8<64>, dim3((unsigned)total), dim3(512), 0, st, p0, p1);
  YL_LAUNCH_CHECK();
  return 0;
}
