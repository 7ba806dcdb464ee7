// common.hpp — shared device helpers for libyolat_hip.so (gfx950 / CDNA4 only).
//
// Hardware model used throughout (MI355X): 64-lane wavefronts, 256-thread workgroups = 4 waves =
// one wave per SIMD, fp32-input MFMA v_mfma_f32_32x32x2_f32 (exact fp32 fma chain, 64 cycles per
// instruction per SIMD), LDS tiles padded by one dword so that the per-lane column reads of the
// MFMA A/B fragments (ds_read_b32, 2 x 32-lane groups) are bank-conflict free.
//
// Loader rule (learned from the ISA): every global load in a staging routine must be UNCONDITIONAL.
// A per-lane "load or zero" branch makes hipcc wrap each load in its own exec-mask region followed
// by s_waitcnt vmcnt(0), i.e. one dependent L2 round trip per element.  So out-of-range rows /
// columns are handled by clamping the address (the loaded value is then discarded by a select, or
// lands in output rows/columns that the epilogue masks) and the aligned interior case is a
// compile-time FAST path with plain 16-byte loads.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/yolat_hip.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

#define YL_LAUNCH_CHECK()                          \
  do {                                             \
    hipError_t e__ = hipGetLastError();            \
    if (e__ != hipSuccess) return (int)e__;        \
  } while (0)

static inline int yl_cdiv(long a, long b) { return (int)((a + b - 1) / b); }
static inline bool yl_aligned16(const void* p) { return (((uintptr_t)p) & 15) == 0; }
// YOLAT_STRICT_FP32=1: every GEMM of the fp32 mode runs on the fp32-input MFMA kernels (v_mfma_f32_32x32x2_f32) instead
// of the bf16x6 emulation (x6.hpp) — the one switch for strict-parity runs (IEEE Inf / NaN propagation, fp32 MFMA
// summation order); the Python side (plan.py, ops.py) reads the same variable.
static inline bool yl_strict_fp32() {
  static int v = -1;
  if (v < 0) { const char* e = getenv("YOLAT_STRICT_FP32"); v = (e && e[0] == '1') ? 1 : 0; }
  return v != 0;
}

// A launch that is the fall-back of another one (conv_local.hip's one-launch conv stack): it runs only when the word at p
// holds val — the epoch the forward raised the flag with — and is a dead launch otherwise.  p == nullptr: always runs.
struct YlGate { const int* p; int val; };
__device__ __forceinline__ bool yl_gate_dead(const YlGate& g) { return g.p != nullptr && *g.p != g.val; }

__device__ __forceinline__ int yl_min(int a, int b) { return a < b ? a : b; }
__device__ __forceinline__ int yl_max(int a, int b) { return a > b ? a : b; }
// a*b rounded on its own: the empty asm keeps the compiler from contracting it with a following add into an
// fma (-ffp-contract=fast does that even through __fmul_rn/__fadd_rn), so two kernels that must agree bit
// for bit can pin the same two roundings
__device__ __forceinline__ float yl_mul_rn(float a, float b) {
  float p = a * b;
  asm volatile("" : "+v"(p));
  return p;
}

// two floats -> packed bfloat16 pair (a in the low half), round-to-nearest-even: v_cvt_pk_bf16_f32
typedef __bf16 yl_bf16x2 __attribute__((ext_vector_type(2)));
typedef float yl_f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned yl_pack_bf16(float a, float b) {
  yl_f32x2 v; v.x = a; v.y = b;
  yl_bf16x2 h = __builtin_convertvector(v, yl_bf16x2);
  return *reinterpret_cast<unsigned*>(&h);
}
__device__ __forceinline__ float yl_bf16_lo(unsigned u) { return __uint_as_float(u << 16); }
__device__ __forceinline__ float yl_bf16_hi(unsigned u) { return __uint_as_float(u & 0xFFFF0000u); }

// ------------------------------------------------------------------------------------------------
// Operand loaders.  load4<FAST>(r, k, v): 4 consecutive k-elements of logical row r (k % 4 == 0).
//   FAST  : caller guarantees `vec` (16-byte loads legal) and k + 3 < cols.
//   !FAST : any alignment / K tail; columns >= cols read as exactly 0.
// Rows >= rows return the (finite or not) contents of the last valid row: every consumer masks
// them (NT GEMM: epilogue row/column masks; TN GEMM: explicit select).
// ------------------------------------------------------------------------------------------------

// Row-major dense operand; PRO adds the per-column affine + ReLU prologue (BatchNorm1d+ReLU of the
// producer applied on the fly):  v = max(v*scale[k] + shift[k], floor),  floor = 0 or -inf.
template <bool PRO>
struct DenseOpT {
  const float* p;
  long ld;
  int rows, cols;
  const float* scale;
  const float* shift;
  float floor;
  int vec;  // ld % 4 == 0, base (and scale/shift) 16-byte aligned

  template <bool FAST>
  __device__ __forceinline__ void load4(int r, int k, float v[4]) const {
    const float* q = p + (long)yl_min(r, rows - 1) * ld;
    if (FAST) {
      const float4 t = *reinterpret_cast<const float4*>(q + k);
      v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
      if (PRO) {
        const float4 s = *reinterpret_cast<const float4*>(scale + k);
        const float4 h = *reinterpret_cast<const float4*>(shift + k);
        v[0] = fmaxf(fmaf(v[0], s.x, h.x), floor);
        v[1] = fmaxf(fmaf(v[1], s.y, h.y), floor);
        v[2] = fmaxf(fmaf(v[2], s.z, h.z), floor);
        v[3] = fmaxf(fmaf(v[3], s.w, h.w), floor);
      }
    } else {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int kc = yl_min(k + j, cols - 1);
        float t = q[kc];
        if (PRO) t = fmaxf(fmaf(t, scale[kc], shift[kc]), floor);
        v[j] = (k + j < cols) ? t : 0.f;
      }
    }
  }
};
typedef DenseOpT<false> DenseOp;
typedef DenseOpT<true> DenseProOp;

// bfloat16-STORED dense operand (training with bf16 storage of the [E,*] tensors, train_bf16.hip): the same loader
// contract, elements converted to fp32 while loading (bf16 -> fp32 is a 16-bit shift: exact), prologue in fp32.
typedef unsigned short yl_bf16_t;
__device__ __forceinline__ float4 yl_ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ float4 yl_ld4(const yl_bf16_t* p) {
  const uint2 u = *reinterpret_cast<const uint2*>(p);
  return make_float4(yl_bf16_lo(u.x), yl_bf16_hi(u.x), yl_bf16_lo(u.y), yl_bf16_hi(u.y));
}
__device__ __forceinline__ void yl_st4(float* p, const float4& v) { *reinterpret_cast<float4*>(p) = v; }
__device__ __forceinline__ void yl_st4(yl_bf16_t* p, const float4& v) {       // round-to-nearest-even
  uint2 u;
  u.x = yl_pack_bf16(v.x, v.y); u.y = yl_pack_bf16(v.z, v.w);
  *reinterpret_cast<uint2*>(p) = u;
}
__device__ __forceinline__ float yl_ld1(const float* p) { return *p; }
__device__ __forceinline__ float yl_ld1(const yl_bf16_t* p) { return __uint_as_float(((unsigned)*p) << 16); }

template <bool PRO>
struct HalfOpT {
  const yl_bf16_t* p;
  long ld;
  int rows, cols;
  const float* scale;
  const float* shift;
  float floor;
  int vec;  // ld % 4 == 0, base 8-byte aligned (and scale/shift 16-byte aligned)

  template <bool FAST>
  __device__ __forceinline__ void load4(int r, int k, float v[4]) const {
    const yl_bf16_t* q = p + (long)yl_min(r, rows - 1) * ld;
    if (FAST) {
      const float4 t = yl_ld4(q + k);
      v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
      if (PRO) {
        const float4 s = *reinterpret_cast<const float4*>(scale + k);
        const float4 h = *reinterpret_cast<const float4*>(shift + k);
        v[0] = fmaxf(fmaf(v[0], s.x, h.x), floor);
        v[1] = fmaxf(fmaf(v[1], s.y, h.y), floor);
        v[2] = fmaxf(fmaf(v[2], s.z, h.z), floor);
        v[3] = fmaxf(fmaf(v[3], s.w, h.w), floor);
      }
    } else {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int kc = yl_min(k + j, cols - 1);
        float t = yl_ld1(q + kc);
        if (PRO) t = fmaxf(fmaf(t, scale[kc], shift[kc]), floor);
        v[j] = (k + j < cols) ? t : 0.f;
      }
    }
  }
};
typedef HalfOpT<false> HalfOp;
typedef HalfOpT<true> HalfProOp;

// Dense operand used transposed: logical element (r, k) = p[k*ld + r]  (r fast in memory).
struct TransOp {
  const float* p;
  long ld;
  int rows, cols;
  int vec;  // 1: the element path is already coalesced along r; FAST only skips the K-tail select
  template <bool FAST>
  __device__ __forceinline__ void load4(int r, int k, float v[4]) const {
    const int rc = yl_min(r, rows - 1);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int kc = yl_min(k + j, cols - 1);
      const float t = p[(long)kc * ld + rc];
      v[j] = (FAST || k + j < cols) ? t : 0.f;
    }
  }
};

// Edge feature rows of AttrRelativeEdgeConvGlobalPool2.message (torch_vertex.py:331), gathered on
// the fly:  row q (CSR slot) = [ x[dst_q] | x[src_q] - x[dst_q] | attr_q ],  K = 2*Cin + 4.
// Branch-free: two unconditional loads (pa, pb) and a select per segment.
struct EdgeOp {
  const float* x;
  long ldx;
  int Cin;
  const int* src;
  const int* dst;
  const float* attr;  // [E,4] contiguous
  int rows;           // E
  int cols;           // 2*Cin + 4
  int vec;            // Cin % 4 == 0, ldx % 4 == 0, x and attr 16-byte aligned

  template <bool FAST>
  __device__ __forceinline__ void load4(int q, int k, float v[4]) const {
    const int qc = yl_min(q, rows - 1);
    const long s = src[qc], d = dst[qc];
    if (FAST) {
      // a float4 never straddles a segment because Cin % 4 == 0
      const bool seg0 = k < Cin, seg1 = !seg0 && (k < 2 * Cin);
      const int kx = seg0 ? k : (seg1 ? k - Cin : 0);
      const float* pa = seg0 ? x + d * ldx + kx : (seg1 ? x + s * ldx + kx : attr + (long)qc * 4);
      const float* pb = x + d * ldx + kx;
      const float4 a = *reinterpret_cast<const float4*>(pa);
      const float4 b = *reinterpret_cast<const float4*>(pb);
      v[0] = seg1 ? a.x - b.x : a.x;
      v[1] = seg1 ? a.y - b.y : a.y;
      v[2] = seg1 ? a.z - b.z : a.z;
      v[3] = seg1 ? a.w - b.w : a.w;
    } else {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int kk = k + j;
        const bool seg0 = kk < Cin, seg1 = !seg0 && (kk < 2 * Cin);
        const bool seg2 = !seg0 && !seg1 && (kk < cols);
        const int kx = seg0 ? kk : (seg1 ? kk - Cin : 0);
        const int ka = seg2 ? kk - 2 * Cin : 0;
        const float xd = x[d * ldx + kx];
        const float xs = x[s * ldx + kx];
        const float at = attr[(long)qc * 4 + ka];
        v[j] = seg0 ? xd : (seg1 ? xs - xd : (seg2 ? at : 0.f));
      }
    }
  }
};

// Combined weight of the edge-MLP input gradient (yolat_edge_lin1_bwd_x): logical (n, k) with
// n in [0, 2*Cin) the input column and k in [0, C) the hidden channel:
//   n <  Cin : W1[k][n] - W1[k][Cin + n]     (gradient reaching x[dst])
//   n >= Cin : W1[k][n]                      (gradient reaching x[src])
struct EdgeWcOp {
  const float* W1;
  long ldw;
  int Cin, C;
  int vec;
  template <bool FAST>
  __device__ __forceinline__ void load4(int n, int k, float v[4]) const {
    const int nc = yl_min(n, 2 * Cin - 1);
    const bool lo = nc < Cin;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int kc = yl_min(k + j, C - 1);
      const float* row = W1 + (long)kc * ldw;
      const float a = row[nc];
      const float b = row[lo ? Cin + nc : nc];
      const float t = lo ? a - b : a;
      v[j] = (FAST || k + j < C) ? t : 0.f;
    }
  }
};

// XCD-aware workgroup -> tile map.  The dispatcher places linear workgroup id b on XCD b % 8 (observed,
// MI355X_MICROARCH.md "Workgroup dispatch"); each XCD has a private 4 MiB L2.  With the plain (x, y) map
// the gridDim.y workgroups that share one A row-tile land on 8 different XCDs, and every one of them
// pulls the tile over the fabric (PMC: 79 MB fetched for the 5.6 MB fusion GEMM at cfg 2).  Here XCD x
// owns a contiguous range of row-major (row-tile, col-tile) pairs, column tile fastest, so the sharers of
// an A tile run back-to-back on one XCD and hit its L2.  A speed choice only: any placement is correct.
__device__ __forceinline__ int l31_of_lane() { return (int)(threadIdx.x & 31); }

__device__ __forceinline__ void yl_xcd_tile(int& rt, int& ct) {
  const int tm = gridDim.x, tn = gridDim.y;
  const int total = tm * tn, id = blockIdx.x + tm * blockIdx.y;
  const int chunk = total >> 3, rem = total & 7;
  const int xcd = id & 7, slot = id >> 3;
  const int logical = xcd * chunk + (xcd < rem ? xcd : rem) + slot;
  rt = logical / tn;
  ct = logical - rt * tn;
}

// ------------------------------------------------------------------------------------------------
// Epilogue of the NT GEMM
// ------------------------------------------------------------------------------------------------
struct Epilogue {
  const float* bias;   // nullable [N]
  const float* scale;  // nullable [N]
  const float* shift;
  int relu;
  float* Y;
  long ldy;
  int accumulate;
  float* stats;  // nullable: float2 [ceil(M/32)][N]  (sum, M2) of (acc + bias) per 32-row group
  // fused per-proposal max pooling (eval): when seg != nullptr the tile is NOT stored; instead
  // pool[seg[row]][col] = max(pool, value) with integer atomicMax on the float bits (values are
  // post-ReLU, i.e. >= 0, so int order == float order and max is exact and order-independent).
  const int* seg;  // nullable [M]
  float* pool;
  long ldpool;
  // training-mode fused pooling (fusion_train.hip): when key64 != nullptr the tile is not stored; for every
  // (segment, column) the row with the largest s*z (s = sign of `scale`, z = acc + bias) is recorded as
  //   key = orderable(s*z) << 32 | ~row     via 64-bit atomicMax  (ties -> lowest row)
  unsigned long long* key64 = nullptr;
  // fused CSR mean aggregation (eval node-side kernel): when agg != nullptr the value stored for row n is
  //   epi(acc)[n] + mean_{q in [agg_ptr[n], agg_ptr[n+1])} agg[q, col]      (ascending q, like k_csr_mean_fwd)
  const float* agg = nullptr;
  long ldagg = 0;
  const int* agg_ptr = nullptr;
  int agg_rows = 0;     // number of rows of agg (E), for address clamping
  // bf16 storage mode (bf16_eval.hip): when Yh != nullptr the tile is stored as bfloat16 (round-to-nearest-even)
  // at Yh[row*ldy + col] instead of fp32 at Y; ldy even, Yh 4-byte aligned.  Not combined with accumulate / agg.
  unsigned short* Yh = nullptr;
};

// Epilogue of one 32x32 MFMA sub-tile held by one wave (C/D layout: col = lane&31,
// row = (reg&3) + 8*(reg>>2) + 4*(lane>>5)).  Everything is wave-local: the training-mode BatchNorm
// partial statistics of the 32-row group (sum, M2 around the group mean) need one cross-half
// shuffle and no LDS, and are written once per (32-row group, column) => deterministic.
// Per-column epilogue constants and the tile's segment ids, loaded at kernel START so their latency hides
// under the K loop (a dependent load after the last MFMA costs ~1-2k cycles of a 4k-cycle K=128 tile).
struct EpiPre { float bias, sc, sh; int segv; };
__device__ __forceinline__ EpiPre epi_prefetch(const Epilogue& ep, int row_base, int col, int M, int N) {
  EpiPre p;
  const int cc = col < N ? col : N - 1;
  p.bias = ep.bias != nullptr ? ep.bias[cc] : 0.f;
  p.sc = ep.scale != nullptr ? ep.scale[cc] : 1.f;
  p.sh = ep.scale != nullptr ? ep.shift[cc] : 0.f;
  const int my_row = row_base + l31_of_lane();
  p.segv = (ep.seg != nullptr && my_row < M) ? ep.seg[my_row] : -1;
  return p;
}

// Fused per-proposal max pooling of one 32x32 sub-tile (bias already added to acc): rows of a proposal are
// consecutive -> run-length max over this lane's 16 rows, one integer atomicMax per run (values are post-ReLU,
// i.e. >= 0, so int order == float order; exact and order-independent).  The 32 segment ids of the tile come from
// ONE coalesced load (pre.segv: lane l31 <- row row_base+l31, -1 beyond M) distributed by cross-lane reads; 16
// dependent per-row loads cost more than the tile's MFMAs.
// sgs[r] = segment id of this lane's r-th row (-1 beyond M), from the tile's coalesced id vector
__device__ __forceinline__ void yl_tile_segs(int segv, int lhi, int sgs[16]) {
#pragma unroll
  for (int r = 0; r < 16; ++r) sgs[r] = __shfl(segv, (r & 3) + 8 * (r >> 2) + 4 * lhi);   // all lanes active here
}
__device__ __forceinline__ void wave_epilogue_segmax(const f32x16& acc, int col, const Epilogue& ep, int N, float sc,
                                                     float sh, const int sgs[16]) {
  const bool col_ok = col < N;
  int cur_seg = -1;
  float cur = 0.f;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int sg = sgs[r];
    const float v = fmaxf(fmaf(acc[r], sc, sh), 0.f);
    if (sg != cur_seg) {
      if (cur_seg >= 0 && col_ok && cur > 0.f)
        atomicMax(reinterpret_cast<int*>(ep.pool + (long)cur_seg * ep.ldpool + col), __float_as_int(cur));
      cur_seg = sg;
      cur = v;
    } else {
      cur = fmaxf(cur, v);
    }
  }
  if (cur_seg >= 0 && col_ok && cur > 0.f)
    atomicMax(reinterpret_cast<int*>(ep.pool + (long)cur_seg * ep.ldpool + col), __float_as_int(cur));
}

// stage (optional): this wave's private [32][YL_STAGE_LD] fp32 region of LDS (free once the K loop's last barrier has
// passed).  The plain stores of a full 32 x 32 sub-tile then go row-major through it: every lane writes 16 bytes, one
// instruction covers 8 rows x 128 bytes (fp32) — instead of 4 bytes per lane and 2 rows x 128 bytes per instruction in
// the accumulator layout, 4x the store instructions for the same bytes.
constexpr int YL_STAGE_LD = 36;
__device__ __forceinline__ bool yl_stage_ok(const Epilogue& ep, int row_base, int col_base, int M, int N) {
  if (row_base + 32 > M || col_base + 32 > N) return false;
  if (ep.Yh != nullptr) return ep.ldy % 8 == 0 && (((uintptr_t)ep.Yh) & 15) == 0;
  return ep.ldy % 4 == 0 && (((uintptr_t)ep.Y) & 15) == 0;
}
__device__ __forceinline__ void wave_epilogue(f32x16 acc, int row_base, int col, int lhi,
                                              const Epilogue& ep, int M, int N, const EpiPre& pre,
                                              float* stage = nullptr) {
  const bool col_ok = col < N;
  const int cc = col_ok ? col : N - 1;
  if (ep.bias != nullptr) {
    const float bv = pre.bias;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] += bv;
  }
  if (ep.stats != nullptr) {
    int cnt = M - row_base;
    cnt = cnt > 32 ? 32 : cnt;
    float s = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = row_base + (r & 3) + 8 * (r >> 2) + 4 * lhi;
      s += (row < M) ? acc[r] : 0.f;
    }
    s += __shfl_xor(s, 32);
    const float mu = cnt > 0 ? s / (float)cnt : 0.f;
    float m2 = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = row_base + (r & 3) + 8 * (r >> 2) + 4 * lhi;
      const float d = acc[r] - mu;
      m2 += (row < M) ? d * d : 0.f;
    }
    m2 += __shfl_xor(m2, 32);
    if (lhi == 0 && col_ok && cnt > 0) {
      float2* dst = reinterpret_cast<float2*>(ep.stats) + (long)(row_base >> 5) * N + col;
      *dst = make_float2(s, m2);
    }
  }
  const float sc = pre.sc, sh = pre.sh;
  const float floor = ep.relu ? 0.f : -INFINITY;
  if (ep.key64 != nullptr) {
    const int segv = pre.segv;
    int sgs[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) sgs[r] = __shfl(segv, (r & 3) + 8 * (r >> 2) + 4 * lhi);
    const bool neg = sc < 0.f;
    int cur_seg = -1;
    unsigned long long cur = 0ull;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = row_base + (r & 3) + 8 * (r >> 2) + 4 * lhi;
      const int sg = sgs[r];
      const float z = neg ? -acc[r] : acc[r];
      unsigned int u = __float_as_uint(z);
      u = (u & 0x80000000u) ? ~u : (u | 0x80000000u);            // order-preserving float -> uint
      const unsigned long long key = ((unsigned long long)u << 32) | (unsigned long long)(0xFFFFFFFFu - (unsigned)row);
      if (sg != cur_seg) {
        if (cur_seg >= 0 && col_ok) atomicMax(ep.key64 + (long)cur_seg * ep.ldpool + col, cur);
        cur_seg = sg;
        cur = key;
      } else {
        cur = key > cur ? key : cur;
      }
    }
    if (cur_seg >= 0 && col_ok) atomicMax(ep.key64 + (long)cur_seg * ep.ldpool + col, cur);
    return;
  }
  if (ep.seg != nullptr) {
    int sgs[16];
    yl_tile_segs(pre.segv, lhi, sgs);
    wave_epilogue_segmax(acc, col, ep, N, pre.sc, pre.sh, sgs);
    return;
  }
  if (ep.Yh != nullptr && stage != nullptr && yl_stage_ok(ep, row_base, col - (int)(threadIdx.x & 31), M, N)) {
    const int l31s = threadIdx.x & 31, lane = threadIdx.x & 63, col_base = col - l31s;
#pragma unroll
    for (int r = 0; r < 16; ++r)
      stage[((r & 3) + 8 * (r >> 2) + 4 * lhi) * YL_STAGE_LD + l31s] = fmaxf(fmaf(acc[r], sc, sh), floor);
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const int chunk = lane + 64 * t, R = chunk >> 2, c = (chunk & 3) * 8;
      const float4 x0 = *reinterpret_cast<const float4*>(stage + R * YL_STAGE_LD + c);
      const float4 x1 = *reinterpret_cast<const float4*>(stage + R * YL_STAGE_LD + c + 4);
      uint4 o;
      o.x = yl_pack_bf16(x0.x, x0.y); o.y = yl_pack_bf16(x0.z, x0.w);
      o.z = yl_pack_bf16(x1.x, x1.y); o.w = yl_pack_bf16(x1.z, x1.w);
      *reinterpret_cast<uint4*>(ep.Yh + (long)(row_base + R) * ep.ldy + col_base + c) = o;
    }
    return;
  }
  if (ep.Yh != nullptr) {
    // lanes l and l^1 hold neighbouring columns: each pair exchanges one value per two rows so that every lane
    // stores one packed 4-byte (col, col+1) pair for 8 of its 16 rows (even lanes: even r, odd lanes: odd r)
    const bool odd = (threadIdx.x & 1) != 0;
    const bool pair_ok = (col | 1) < N;          // both columns of the pair inside N (col < N is implied)
#pragma unroll
    for (int r = 0; r < 16; r += 2) {
      const float v0 = fmaxf(fmaf(acc[r], sc, sh), floor), v1 = fmaxf(fmaf(acc[r + 1], sc, sh), floor);
      const float got = __shfl_xor(odd ? v0 : v1, 1);
      const int row = row_base + ((r + (odd ? 1 : 0)) & 3) + 8 * (r >> 2) + 4 * lhi;
      const float lo = odd ? got : v0, hi = odd ? v1 : got;
      unsigned short* dst = ep.Yh + (long)row * ep.ldy + (col & ~1);
      if (row < M) {
        if (pair_ok) *reinterpret_cast<unsigned*>(dst) = yl_pack_bf16(lo, hi);
        else if (!odd && col_ok) *dst = (unsigned short)(yl_pack_bf16(lo, 0.f) & 0xFFFFu);
      }
    }
    // when N is odd the last column's even-lane owner also has to store the rows its odd partner would have
    if (!pair_ok && !odd && col_ok) {
#pragma unroll
      for (int r = 1; r < 16; r += 2) {
        const int row = row_base + (r & 3) + 8 * (r >> 2) + 4 * lhi;
        const float v = fmaxf(fmaf(acc[r], sc, sh), floor);
        if (row < M) ep.Yh[(long)row * ep.ldy + col] = (unsigned short)(yl_pack_bf16(v, 0.f) & 0xFFFFu);
      }
    }
    return;
  }
  float old[16];
  if (ep.agg != nullptr) {
    // CSR range boundaries of the tile's 32 rows: two coalesced loads + cross-lane reads
    const int n_l = yl_min(row_base + l31_of_lane(), M);
    const int rp_lo = ep.agg_ptr[n_l], rp_hi = ep.agg_ptr[yl_min(n_l + 1, M)];
    int q0s[16], q1s[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int off = (r & 3) + 8 * (r >> 2) + 4 * lhi;
      q0s[r] = __shfl(rp_lo, off);
      q1s[r] = __shfl(rp_hi, off);
    }
    const float* hp = ep.agg + cc;
    const int qmax = ep.agg_rows - 1;
#pragma unroll
    for (int r0 = 0; r0 < 16; r0 += 4) {      // 4 rows x 4 edges = 16 independent (clamped) loads in flight
      float v[4][4];
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int e = 0; e < 4; ++e)
          v[j][e] = hp[(long)yl_min(q0s[r0 + j] + e, qmax) * ep.ldagg];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int q0 = q0s[r0 + j], q1 = q1s[r0 + j], deg = q1 - q0;
        float sacc = 0.f;
#pragma unroll
        for (int e = 0; e < 4; ++e) sacc += (e < deg) ? v[j][e] : 0.f;
        for (int q = q0 + 4; q < q1; ++q) sacc += hp[(long)q * ep.ldagg];
        old[r0 + j] = yl_mul_rn(sacc, 1.f / (float)(deg > 1 ? deg : 1));
      }
    }
  } else if (ep.accumulate) {   // all 16 reads issued back to back (clamped rows), one wait
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = yl_min(row_base + (r & 3) + 8 * (r >> 2) + 4 * lhi, M - 1);
      old[r] = ep.Y[(long)row * ep.ldy + cc];
    }
  } else {
#pragma unroll
    for (int r = 0; r < 16; ++r) old[r] = 0.f;
  }
  if (stage != nullptr && yl_stage_ok(ep, row_base, col - (int)(threadIdx.x & 31), M, N)) {
    const int l31s = threadIdx.x & 31, lane = threadIdx.x & 63, col_base = col - l31s;
#pragma unroll
    for (int r = 0; r < 16; ++r)
      stage[((r & 3) + 8 * (r >> 2) + 4 * lhi) * YL_STAGE_LD + l31s] = fmaxf(fmaf(acc[r], sc, sh), floor) + old[r];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const int chunk = lane + 64 * t, R = chunk >> 3, c = (chunk & 7) * 4;
      *reinterpret_cast<float4*>(ep.Y + (long)(row_base + R) * ep.ldy + col_base + c) =
          *reinterpret_cast<const float4*>(stage + R * YL_STAGE_LD + c);
    }
    return;
  }
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int row = row_base + (r & 3) + 8 * (r >> 2) + 4 * lhi;
    const float v = fmaxf(fmaf(acc[r], sc, sh), floor) + old[r];
    if (col_ok && row < M) ep.Y[(long)row * ep.ldy + col] = v;
  }
}

// ------------------------------------------------------------------------------------------------
// NT GEMM:  Y[M,N] = epi( A[M,K] . B[N,K]^T )   — 256 threads, 2x2 waves, wave tile (BM/2)x(BN/2)
// built from 32x32x2 fp32 MFMAs.  LDS tiles As[BM][BK+1], Bs[BN][BK+1] (the +1 dword pad makes the
// per-lane column reads of the fragments conflict-free).  The global loads of the next K-tile are
// issued into registers before the MFMAs of the current one (software pipeline), so HBM/L2 latency
// hides under the 64-cycle fp32 MFMAs.
//   MFMA operand layout (cdna_hip_programming.md §3): A: lane l holds A[i=l&31][k=l>>5];
//   B: lane l holds B[k=l>>5][j=l&31].
// B_NFAST: the B loader is contiguous along n (transposed operands) -> map consecutive threads to
// consecutive n when staging.
// ------------------------------------------------------------------------------------------------
template <int BM, int BN, int BK, class AL, class BL, bool B_NFAST>
__device__ __forceinline__ void gemm_nt_tile(const AL& A, const BL& B, const Epilogue& ep, int M, int N, int K,
                                             int rt_, int ct_) {
  constexpr int WM = BM / 2, WN = BN / 2, TM = WM / 32, TN = WN / 32;
  constexpr int LD = BK + 1;
  constexpr int KQ = BK / 4;
  constexpr int NA = (BM * KQ) / 256, NB = (BN * KQ) / 256;   // float4 loads per thread per tile
  static_assert((BM * KQ) % 256 == 0 && (BN * KQ) % 256 == 0, "tile/thread mismatch");
  // (at least the 4 x [32][YL_STAGE_LD] floats of the epilogue's store staging, which re-uses the operand tiles)
  constexpr int SMEM = (BM + BN) * LD > 4 * 32 * YL_STAGE_LD ? (BM + BN) * LD : 4 * 32 * YL_STAGE_LD;
  __shared__ __attribute__((aligned(16))) float smem[SMEM];
  float* As = smem;
  float* Bs = smem + BM * LD;

  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int wm = wave >> 1, wn = wave & 1;
  const int l31 = lane & 31, lhi = lane >> 5;
  const int row0 = rt_ * BM, col0 = ct_ * BN;
  const bool fastA = A.vec != 0, fastB = B.vec != 0;

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  float ra[NA][4], rb[NB][4];
  auto fetch = [&](int k0) {
    const bool full = (k0 + BK <= K);           // wave-uniform
    if (full && fastA) {
#pragma unroll
      for (int t = 0; t < NA; ++t) {
        const int i = tid + t * 256;
        A.template load4<true>(row0 + i / KQ, k0 + 4 * (i % KQ), ra[t]);
      }
    } else {
#pragma unroll
      for (int t = 0; t < NA; ++t) {
        const int i = tid + t * 256;
        A.template load4<false>(row0 + i / KQ, k0 + 4 * (i % KQ), ra[t]);
      }
    }
    if (full && fastB) {
#pragma unroll
      for (int t = 0; t < NB; ++t) {
        const int i = tid + t * 256;
        const int n = B_NFAST ? (i % BN) : (i / KQ);
        const int kq = B_NFAST ? (i / BN) : (i % KQ);
        B.template load4<true>(col0 + n, k0 + 4 * kq, rb[t]);
      }
    } else {
#pragma unroll
      for (int t = 0; t < NB; ++t) {
        const int i = tid + t * 256;
        const int n = B_NFAST ? (i % BN) : (i / KQ);
        const int kq = B_NFAST ? (i / BN) : (i % KQ);
        B.template load4<false>(col0 + n, k0 + 4 * kq, rb[t]);
      }
    }
  };
  auto stage = [&]() {
#pragma unroll
    for (int t = 0; t < NA; ++t) {
      const int i = tid + t * 256;
      float* d = As + (i / KQ) * LD + 4 * (i % KQ);
      d[0] = ra[t][0]; d[1] = ra[t][1]; d[2] = ra[t][2]; d[3] = ra[t][3];
    }
#pragma unroll
    for (int t = 0; t < NB; ++t) {
      const int i = tid + t * 256;
      const int n = B_NFAST ? (i % BN) : (i / KQ);
      const int kq = B_NFAST ? (i / BN) : (i % KQ);
      float* d = Bs + n * LD + 4 * kq;
      d[0] = rb[t][0]; d[1] = rb[t][1]; d[2] = rb[t][2]; d[3] = rb[t][3];
    }
  };

  EpiPre pre[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
      pre[i][j] = epi_prefetch(ep, row0 + wm * WM + i * 32, col0 + wn * WN + j * 32 + l31, M, N);
  fetch(0);
  for (int k0 = 0; k0 < K; k0 += BK) {
    stage();
    __syncthreads();
    if (k0 + BK < K) fetch(k0 + BK);      // in flight while the MFMAs below run
#pragma unroll
    for (int kk = 0; kk < BK; kk += 2) {
      float a[TM], b[TN];
#pragma unroll
      for (int i = 0; i < TM; ++i) a[i] = As[(wm * WM + i * 32 + l31) * LD + kk + lhi];
#pragma unroll
      for (int j = 0; j < TN; ++j) b[j] = Bs[(wn * WN + j * 32 + l31) * LD + kk + lhi];
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[j], acc[i][j], 0, 0, 0);
    }
    __syncthreads();
  }

  // spelled out with compile-time indices: the compiler refuses to unroll a loop around the (large) inlined
  // epilogue for TM*TN = 4, and a rolled loop indexes acc[][] dynamically = accumulators in scratch memory
  static_assert(TM <= 2 && TN <= 2, "epilogue expansion covers up to 2x2 sub-tiles");
#define YL_EPI(i, j)                                                                                              \
  if constexpr ((i) < TM && (j) < TN)                                                                             \
    wave_epilogue(acc[i][j], row0 + wm * WM + (i) * 32, col0 + wn * WN + (j) * 32 + l31, lhi, ep, M, N, pre[i][j], \
                  smem + wave * (32 * YL_STAGE_LD));
  YL_EPI(0, 0) YL_EPI(0, 1) YL_EPI(1, 0) YL_EPI(1, 1)
#undef YL_EPI
}

template <int BM, int BN, int BK, class AL, class BL, bool B_NFAST>
__global__ void __launch_bounds__(256) k_gemm_nt(AL A, BL B, Epilogue ep, int M, int N, int K) {
  int rt_, ct_;
  yl_xcd_tile(rt_, ct_);
  gemm_nt_tile<BM, BN, BK, AL, BL, B_NFAST>(A, B, ep, M, N, K, rt_, ct_);
}

// Two independent GEMMs with the same M, N <= BN and K in ONE launch (blockIdx.y picks the problem): the
// root Linear (+ fused CSR mean) and the node-branch Linear+BN+ReLU of a conv layer are 157 workgroups each
// at cfg 2 — separately they are two launch/drain latencies on a quarter-filled GPU.
template <int BM, int BN, int BK, class AL, class BL>
__global__ void __launch_bounds__(256) k_gemm_nt_pair(AL A0, BL B0, Epilogue e0, AL A1, BL B1, Epilogue e1,
                                                      int M, int N, int K) {
  if (blockIdx.y == 0) gemm_nt_tile<BM, BN, BK, AL, BL, false>(A0, B0, e0, M, N, K, blockIdx.x, 0);
  else gemm_nt_tile<BM, BN, BK, AL, BL, false>(A1, B1, e1, M, N, K, blockIdx.x, 0);
}

// Two independent 64x64x32 GEMMs in one flattened 1-D launch: the first (padded) tm1*tn1 workgroups tile
// problem 1 row-major, the rest tile problem 0 with the XCD-aware map over its tm0 x tn0 tiles.  Used for the fusion block over the N
// nodes (+ segment-max epilogue) together with fusion_block_super over the P proposals: the small GEMM's
// 112 workgroups ride along with the big one's 2512 instead of being a launch of their own.
static __global__ void __launch_bounds__(256) k_gemm_nt_two(DenseOp A0, DenseOp B0, Epilogue e0, int M0, int N0, int K0, int tm0,
                                                     int tn0, DenseOp A1, DenseOp B1, Epilogue e1, int M1, int N1,
                                                     int K1, int tm1, int tn1) {
  // the small problem's workgroups come FIRST (padded to a multiple of 8 so that the big problem's ids keep
  // their id % 8 = XCD alignment): they start with the first wave of workgroups instead of forming a tail
  const int n1 = tm1 * tn1, n1p = (n1 + 7) & ~7;
  const int id = blockIdx.x;
  if (id < n1p) {
    if (id < n1) gemm_nt_tile<64, 64, 32, DenseOp, DenseOp, false>(A1, B1, e1, M1, N1, K1, id / tn1, id % tn1);
    return;
  }
  const int n0 = tm0 * tn0, j = id - n1p;
  const int chunk = n0 >> 3, rem = n0 & 7;
  const int xcd = j & 7, slot = j >> 3;
  const int logical = xcd * chunk + (xcd < rem ? xcd : rem) + slot;
  gemm_nt_tile<64, 64, 32, DenseOp, DenseOp, false>(A0, B0, e0, M0, N0, K0, logical / tn0, logical % tn0);
}

// Operands / epilogues of the node side of a factorised conv layer (built by yl_build_node_uv, dense.hip)
struct NodeUv {
  DenseOp af, wuv, wr, as, wn;
  Epilogue euv, er, en;
  int N, C, Cin;
};
int yl_build_node_uv(NodeUv* a, const float* f_in, int64_t ld_f, const float* s_in, int64_t ld_s, int64_t N,
                     int64_t Cin, const float* Wuv, const float* uv_bias, const float* Wr, const float* br,
                     const float* Wn, const float* bn, const float* sn, const float* tn, int64_t C, float* UV,
                     int64_t ld_uv, float* f_out, int64_t ld_fo, float* s_out, int64_t ld_so);
// ------------------------------------------------------------------------------------------------
// Node side of the FIRST conv layer of a large graph as an output stream (launcher: dense.hip yl_node3_smallk).  K = in_channels = 5 raw Bezier features -> 256 outputs per node
// (UV [N, 128], root [N, 64], node branch [N, 64]; architecture3cc_rpn_gp_iter2.py:110-113 with torch_vertex.py:319-337
// factorised): with K <= 8 there is nothing for the matrix cores to do, the launch is a 100 - 200 MB output stream.  One
// wave owns 8 consecutive rows per step, lane = four consecutive output columns with their weights in registers (4 x K
// fma per row), 16-byte (fp32) / 8-byte (bf16) stores that cover whole cache lines.  Same epilogue arithmetic as wave_epilogue (bias, scale / shift, ReLU, bf16 = nearest even);
// the sum over k runs in ascending order.  Measured at N = 200 k (tools/exp/node3_bench.py): 39 us for the 205 MB of
// fp32 outputs = 5.3 TB/s, against 6.8 - 7.8 TB/s of a plain fill on this part (tools/exp/stream_bw.py) and 57 - 65 us
// for the 64x64 MFMA tiles of k_gemm_nt_node3 (one load -> LDS -> MFMA -> store chain per 16 KB of output; 80 - 90 us
// together with the CSR emission they were co-scheduled with).  With the stores switched off the kernel runs 22 us
// (its ~44 vector-ALU operations per row), with the FMAs switched off 45: the store stream is the bound.
// ------------------------------------------------------------------------------------------------
constexpr int N3_KMAX = 8, N3_ROWS = 32, N3_ITERS = 4;        // 4 waves x 8 consecutive rows per step, steps per workgroup
// WAVES waves per workgroup (8 consecutive rows per wave and step), ITERS steps: the 256-thread stream kernel is <4, 4>,
// the node-side workgroups of the one-launch graph preparation (graph.hip k_prep_small, 1024 threads) are <16, 1>
template <int WAVES = 4, int ITERS = N3_ITERS>
__device__ __forceinline__ void node3_smallk_body(const NodeUv& a, int vb) {
  constexpr int ROWS = 8 * WAVES;
  // ALL inputs of the workgroup's 128 rows are fetched up front into LDS (lane (row u = lane / 8, k = lane % 8) of the
  // row's wave; read back as two uniform-address 16-byte reads per row), so that the row loop holds no load: on this
  // ISA loads and stores share one in-order counter, and a load issued behind the previous step's stores made every
  // step wait for those stores' acknowledgements (51 -> 44 us).
  __shared__ __attribute__((aligned(16))) float xs[ITERS][WAVES][2][64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int C = a.C, Cin = a.Cin, N = a.N;
  const int r0 = vb * (ROWS * ITERS), r1 = yl_min(r0 + ROWS * ITERS, N);
  const int c4 = 4 * lane;
  const int prob = c4 < 2 * C ? 0 : (c4 < 3 * C ? 1 : 2);
  const int col = prob == 0 ? c4 : (prob == 1 ? c4 - 2 * C : c4 - 3 * C);
  const int relu = prob == 0 ? a.euv.relu : (prob == 1 ? a.er.relu : a.en.relu);
  float* Y = prob == 0 ? a.euv.Y : (prob == 1 ? a.er.Y : a.en.Y);
  unsigned short* Yh = prob == 0 ? a.euv.Yh : (prob == 1 ? a.er.Yh : a.en.Yh);
  const long ldy = prob == 0 ? a.euv.ldy : (prob == 1 ? a.er.ldy : a.en.ldy);
  const int lu = lane >> 3, lk = lane & 7;
  float fa[ITERS], sa[ITERS];
#pragma unroll
  for (int it = 0; it < ITERS; ++it) {
    // (clamped addresses + select, here and for the weights below: a load under a condition becomes a branch with
    // its own wait, and the kernel's start was a chain of ~30 such round trips)
    const long row = yl_min(r0 + ROWS * it + 8 * wave + lu, N - 1);
    const int kc = yl_min(lk, Cin - 1);
    const float f = a.af.p[row * a.af.ld + kc], g = a.as.p[row * a.as.ld + kc];
    fa[it] = lk < Cin ? f : 0.f;
    sa[it] = lk < Cin ? g : 0.f;
  }
  // the 256 weight rows (5 KB) and per-column constants: one row / one column per thread, through LDS.  (Every lane
  // fetching its own 4 x K weights straight from memory touched 64 different cache lines per load instruction, in
  // every one of 1563 workgroups: 44 -> 39 us.)
  __shared__ __attribute__((aligned(16))) float wl[256][N3_KMAX];
  __shared__ float cl[3][256];
  if (threadIdx.x < 256) {
    const int t = threadIdx.x;                              // virtual column t of [UV | root | node branch]
    const int tp = t < 2 * C ? 0 : (t < 3 * C ? 1 : 2);
    const int tc = tp == 0 ? t : (tp == 1 ? t - 2 * C : t - 3 * C);
    const float* __restrict__ twp = tp == 0 ? a.wuv.p : (tp == 1 ? a.wr.p : a.wn.p);
    const long tld = tp == 0 ? a.wuv.ld : (tp == 1 ? a.wr.ld : a.wn.ld);
    const float* tb = tp == 0 ? a.euv.bias : (tp == 1 ? a.er.bias : a.en.bias);
    const float* ts = tp == 0 ? a.euv.scale : (tp == 1 ? a.er.scale : a.en.scale);
    const float* th = tp == 0 ? a.euv.shift : (tp == 1 ? a.er.shift : a.en.shift);
    float wr_[N3_KMAX];
#pragma unroll
    for (int k = 0; k < N3_KMAX; ++k) wr_[k] = twp[(long)tc * tld + yl_min(k, Cin - 1)];
    const float vb_ = (tb ? tb : twp)[tb ? tc : 0], vs_ = (ts ? ts : twp)[ts ? tc : 0], vh_ = (ts ? th : twp)[ts ? tc : 0];
#pragma unroll
    for (int k = 0; k < N3_KMAX; ++k) wl[t][k] = k < Cin ? wr_[k] : 0.f;
    cl[0][t] = tb ? vb_ : 0.f;
    cl[1][t] = ts ? vs_ : 1.f;
    cl[2][t] = ts ? vh_ : 0.f;
  }
#pragma unroll
  for (int it = 0; it < ITERS; ++it) { xs[it][wave][0][lane] = fa[it]; xs[it][wave][1][lane] = sa[it]; }
  __syncthreads();
  float w[4][N3_KMAX], b[4], sc[4], sh[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const float4 w0 = *reinterpret_cast<const float4*>(&wl[c4 + j][0]), w1 = *reinterpret_cast<const float4*>(&wl[c4 + j][4]);
    w[j][0] = w0.x; w[j][1] = w0.y; w[j][2] = w0.z; w[j][3] = w0.w;
    w[j][4] = w1.x; w[j][5] = w1.y; w[j][6] = w1.z; w[j][7] = w1.w;
    b[j] = cl[0][c4 + j]; sc[j] = cl[1][c4 + j]; sh[j] = cl[2][c4 + j];
  }
  const float floor = relu ? 0.f : -INFINITY;
#pragma unroll 1
  for (int it = 0; it < ITERS; ++it) {
    const float* mine = xs[it][wave][prob == 2 ? 1 : 0];
    const int rb = r0 + ROWS * it + 8 * wave;
    if (rb >= r1) break;
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const float4 x0 = *reinterpret_cast<const float4*>(mine + 8 * u);
      const float4 x1 = *reinterpret_cast<const float4*>(mine + 8 * u + 4);
      const float x[8] = {x0.x, x0.y, x0.z, x0.w, x1.x, x1.y, x1.z, x1.w};
      float v[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        float acc = 0.f;
#pragma unroll
        for (int k = 0; k < N3_KMAX; ++k) acc = fmaf(x[k], w[j][k], acc);
        v[j] = fmaxf(fmaf(acc + b[j], sc[j], sh[j]), floor);
      }
      const int row = rb + u;
      if (row < r1) {
        if (Yh != nullptr) {
          uint2 o;
          o.x = yl_pack_bf16(v[0], v[1]); o.y = yl_pack_bf16(v[2], v[3]);
          *reinterpret_cast<uint2*>(Yh + (long)row * ldy + col) = o;
        } else {
          *reinterpret_cast<float4*>(Y + (long)row * ldy + col) = make_float4(v[0], v[1], v[2], v[3]);
        }
      }
    }
  }
}

// the node side of the first conv layer (K = in_channels <= 8) of a large graph as an output stream (dense.hip)
bool yl_node3_smallk_ok(const NodeUv& a);
bool yl_node3_smallk_shape_ok(const NodeUv& a);
int yl_node3_smallk(const NodeUv& a, hipStream_t st, YlGate gate = YlGate{nullptr, 0});
// training-mode fusion GEMM with the key64 pooling epilogue on the bf16x6 rows kernel (fusion_x6.hip)
int yl_fusion_rows_x6_key64(const float* A, long lda, long N, long K, const float* W, const float* bias, long F,
                            const float* sgn, const int* node_seg, unsigned long long* keys, uint16_t* wsplit,
                            yolat_stream_t stream);
// ------------------------------------------------------------------------------------------------
// Pooling prologue as a RIDER of other launches (small graphs, where a launch of its own is ~5 us of latency):
// the parts of k_pool_prepare (segment.hip) done by `blocks` extra workgroups appended to a kernel whose own
// workgroups do not depend on them:
//   YL_POOL_ZERO  Z[p, 0:F] = 0                      } ride in the LAST conv layer's edge kernel (need the node branch
//   YL_POOL_MEAN  Z[p, 2F+D:2F+2D] = mean(fsup rows)  } of that layer, ready one launch earlier; read by the fusion launch)
//   YL_POOL_MAX   Z[p, F:F+D] = max(feats rows)        ride in the fusion launch itself (reads feats, like its GEMM)
// Same arithmetic, in the same (row) order, as k_pool_prepare.
// ------------------------------------------------------------------------------------------------
#define YL_POOL_ZERO 1
#define YL_POOL_MAX 2
#define YL_POOL_MEAN 4
struct PoolRider {
  const float* feats; const float* fsup; long ld; int D, F, P; const int* seg_ptr; float* Z; long ldz;
  int parts, blocks;
};
// virtual block vb of nvb, thread tid of nthreads (grid-stride over the items of the selected parts)
__device__ __forceinline__ void yl_pool_rider(const PoolRider& r, int vb, int nvb, int tid, int nthreads) {
  const int wz = (r.parts & YL_POOL_ZERO) ? r.F : 0, wx = (r.parts & YL_POOL_MAX) ? r.D : 0,
            wm = (r.parts & YL_POOL_MEAN) ? r.D : 0;
  const int W = wz + wx + wm;
  const long total = (long)r.P * W;
  for (long i = (long)vb * nthreads + tid; i < total; i += (long)nvb * nthreads) {
    const int p = (int)(i / W), c = (int)(i - (long)p * W);
    float* z = r.Z + (long)p * r.ldz;
    if (c < wz) { z[c] = 0.f; continue; }
    const bool is_max = c < wz + wx;
    const int k = is_max ? c - wz : c - wz - wx;
    const float* src = (is_max ? r.feats : r.fsup) + k;
    const int r0 = r.seg_ptr[p], r1 = r.seg_ptr[p + 1];
    float best = 0.f, sm = 0.f;
    bool any = false;
    int q = r0;
    for (; q + 8 <= r1; q += 8) {
      float v[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = src[(long)(q + j) * r.ld];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        if (!any || v[j] > best) { best = v[j]; any = true; }
        sm += v[j];
      }
    }
    for (; q < r1; ++q) {
      const float v = src[(long)q * r.ld];
      if (!any || v > best) { best = v; any = true; }
      sm += v;
    }
    if (is_max) z[r.F + k] = best;
    else {
      const int cnt = r1 - r0;
      z[2 * r.F + r.D + k] = sm / (float)(cnt > 1 ? cnt : 1);
    }
  }
}
// The node side of the NEXT conv layer computed inside the edge launch (small graphs, node tiles of <= 16 nodes), so
// that a conv layer is ONE launch instead of two (k_gemm_nt_node3 was ~10 us of latency at N = 10 k):
//  * UV' / root': a tile owns its destination nodes, so once their rows of f_out are final it multiplies them by the next
//    layer's stacked [Wuv' ; Wr'] (192 x 64) — 12 column tiles x 16 v_mfma_f32_16x16x4_f32, B fragments prefetched from the
//    packed weight — and writes the next UV rows and the next root term;
//      Wp   Wp[(ct * 16 + ks) * 64 + l] = W'[ct * 16 + (l & 15)][4 * ks + (l >> 4)]   (coalesced B fragments)
//      bias [192] = [uvb' ; br'];  UV [N, 2C] of the next layer (NOT the buffer this launch gathers from), root = next f_out
//  * the node branch s' = relu(bn(s . Wn'^T)) of the next layer does not depend on this layer's messages at all: s_tiles
//    extra workgroups run it as 64 x 64 x 64 tiles on the edge kernel's own LDS tiles and MFMA loop.
struct EdgeNext {
  const float* Wp; const float* bias; float* UV; long ld_uv; float* root; long ld_root;
  const float* s_in; long ld_si; const float* Wn; const float* bn; const float* sn; const float* tn;
  float* s_out; long ld_so; int s_tiles;
};
// 0: the automatic choice is not the node-tile kernel; 1 / 4: node tiles with <= 16 / <= 64 nodes (edge.hip)
int yl_edge_tile_groups(int64_t N, int64_t E);
// yolat_edge_uv_mlp2_mean_eval_variant with an optional rider / next-layer node side (edge.hip): *rode = 1 when the
// launched kernel carried the rider, *did_next = 1 when it computed `next`
int yl_edge_uv_mlp2_mean_eval_impl(const float* UV, int64_t ld_uv, const int32_t* src_csr, const int32_t* dst_csr,
                                   const float* attr_csr, const int32_t* row_ptr, int64_t N, int64_t E, const float* Wc4,
                                   const float* b1, const float* s1, const float* t1, const float* W2, const float* b2,
                                   const float* s2, const float* t2, int64_t C, float* f_out, int64_t ld_fo, int variant,
                                   const PoolRider* rider, int* rode, const EdgeNext* next, int* did_next,
                                   yolat_stream_t stream);
// yolat_fusion_pair_eval_x6 with an optional rider (fusion_x6.hip)
int yl_fusion_pair_eval_x6_impl(const float* A, int64_t lda, int64_t N, int64_t D, const uint16_t* Wh, const uint16_t* Wm,
                                const uint16_t* Wl, const float* tfold, int64_t F, const int32_t* node_seg, float* pool,
                                int64_t ldpool, const float* S, int64_t lds, int64_t P, const uint16_t* Wsh,
                                const uint16_t* Wsm, const uint16_t* Wsl, const float* tsfold, float* Ys, int64_t ldys,
                                const PoolRider* rider, yolat_stream_t stream);
// k_pool_prepare restricted to `parts` (segment.hip)
int yl_pool_prepare_parts(const float* feats, const float* fsup, int64_t ld, int64_t D, int64_t F, const int32_t* seg_ptr,
                          int64_t P, float* Z, int64_t ldz, int parts, yolat_stream_t stream);
// stage profiler hooks (forward_eval.hip): HIP-event pair around one stage of a whole-forward entry point
bool yl_profile_on();
void yl_stage_begin(const char* name, double flops, double bytes, yolat_stream_t stream);
void yl_stage_end(yolat_stream_t stream);

// inputs of the one-launch conv stack (conv_local.hip): a prepared destination-sorted graph (row_ptr / src / dst / attr in
// CSR order), or — edge != nullptr — the raw COO list (attr in COO order) with per-proposal edge ranges eptr
struct YlLocalIn {
  const int32_t *row_ptr, *src, *dst; const float* attr; const int32_t* seg_ptr;
  const int64_t* edge; int64_t se, sc; const int32_t* eptr;
  int32_t* status;         // non-null: the caller vouched for the batch, a violation is reported as YOLAT_STATUS_NOT_LOCAL
};
// seg_ptr / node_seg / eptr of a COO batch grouped by proposal + YOLAT_LOC_* violations (conv_local.hip)
int yl_local_prep(const int64_t* edge, int64_t se, int64_t sc, const int64_t* bbox_idx, int64_t N, int64_t E, int64_t P,
                  int32_t* seg_ptr, int32_t* node_seg, int32_t* eptr, int32_t* status, int32_t* info, bool vouched,
                  hipStream_t st);

// yolat_graph_prepare with an optional co-scheduled node-side GEMM set (graph.hip)
int yl_graph_prepare_impl(const int64_t* edge, int64_t stride_e, int64_t stride_c, const float* e_attr,
                          const int64_t* bbox_idx, int64_t E, int64_t N, int64_t P, int32_t* row_ptr, int32_t* perm,
                          int32_t* src_csr, int32_t* dst_csr, float* attr_csr, int32_t* seg_ptr, int32_t* node_seg,
                          int32_t* work, int32_t* status, const NodeUv* extra, bool primed, yolat_stream_t stream);
// tile y of row tile x: y = 0,1 -> UV halves, 2 -> root Linear, 3 -> node-branch Linear
template <int BK>
__device__ __forceinline__ void node_uv_tile(const NodeUv& a, int x, int y) {
  if (y < 2) gemm_nt_tile<64, 64, BK, DenseOp, DenseOp, false>(a.af, a.wuv, a.euv, a.N, 2 * a.C, a.Cin, x, y);
  else if (y == 2) gemm_nt_tile<64, 64, BK, DenseOp, DenseOp, false>(a.af, a.wr, a.er, a.N, a.C, a.Cin, x, 0);
  else gemm_nt_tile<64, 64, BK, DenseOp, DenseOp, false>(a.as, a.wn, a.en, a.N, a.C, a.Cin, x, 0);
}

// Node side of a factorised conv layer in one launch: blockIdx.y = 0,1 -> the two 64-column halves of
// UV = f_in.Wuv^T (N = 128 outputs), 2 -> root Linear, 3 -> node-branch Linear+BN+ReLU.
template <int BK>
__global__ void __launch_bounds__(256) k_gemm_nt_node3(NodeUv a) {
  node_uv_tile<BK>(a, blockIdx.x, blockIdx.y);
}

// ------------------------------------------------------------------------------------------------
// Skinny NT GEMM with split-K inside the workgroup, for few rows x long K (the per-proposal
// classifier layers: P x 2304 -> 512 ...).  One workgroup = one 32x32 output tile; its 4 waves each
// own a quarter of every 128-deep K chunk, so a P=400 problem becomes ~200 workgroups instead of
// 56.  The four partial accumulators are summed through LDS in a fixed order by wave 0, which then
// runs the same epilogue (bias / BatchNorm statistics / scale-shift-ReLU / store).
// ------------------------------------------------------------------------------------------------
template <class AL, class BL, bool B_NFAST>
__global__ void __launch_bounds__(256) k_gemm_nt_sk(AL A, BL B, Epilogue ep, int M, int N, int K) {
  constexpr int BT = 32, BK = 128, LD = BK + 1;
  __shared__ float smem[2 * BT * LD];
  float* As = smem;
  float* Bs = smem + BT * LD;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int l31 = lane & 31, lhi = lane >> 5;
  int rt_, ct_;
  yl_xcd_tile(rt_, ct_);
  const int row0 = rt_ * BT, col0 = ct_ * BT;
  const bool fastA = A.vec != 0, fastB = B.vec != 0;
  // Staging map: float4 slot i (0..1023) -> (row, kq).  32 consecutive lanes cover a 4-row x 8-kq patch:
  // each row gets a full 128-B line from global, and with LD = 129 the ds_write_b32 banks
  // (row + 4*kq + j) mod 32 of a 32-lane group are all distinct.  (The naive row = i/32, kq = i%32 map
  // is a 4-way bank conflict on every write and made this kernel LDS-write-bound.)
  auto map_r = [](int i) { return 4 * ((i >> 5) & 7) + ((i & 31) >> 3); };
  auto map_q = [](int i) { return 8 * (i >> 8) + (i & 7); };
  const EpiPre pre = epi_prefetch(ep, row0, col0 + l31, M, N);

  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;

  // Two register sets = two K-chunks in flight: with ~1 workgroup per CU (P is a few hundred rows)
  // there is no other wave to hide the L2/MALL latency of the 4.7 MB classifier weight behind.
  float ra0[4][4], rb0[4][4], ra1[4][4], rb1[4][4];
  auto fetch = [&](int k0, float (&ra)[4][4], float (&rb)[4][4]) {
    const bool full = (k0 + BK <= K);
    if (full && fastA) {
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const int i = tid + t * 256;
        A.template load4<true>(row0 + map_r(i), k0 + 4 * map_q(i), ra[t]);
      }
    } else {
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const int i = tid + t * 256;
        A.template load4<false>(row0 + map_r(i), k0 + 4 * map_q(i), ra[t]);
      }
    }
    if (full && fastB) {
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const int i = tid + t * 256;
        const int n = B_NFAST ? (i % BT) : map_r(i);
        const int kq = B_NFAST ? (i / BT) : map_q(i);
        B.template load4<true>(col0 + n, k0 + 4 * kq, rb[t]);
      }
    } else {
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const int i = tid + t * 256;
        const int n = B_NFAST ? (i % BT) : map_r(i);
        const int kq = B_NFAST ? (i / BT) : map_q(i);
        B.template load4<false>(col0 + n, k0 + 4 * kq, rb[t]);
      }
    }
  };
  auto stage = [&](float (&ra)[4][4], float (&rb)[4][4]) {
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const int i = tid + t * 256;
      float* d = As + map_r(i) * LD + 4 * map_q(i);
      d[0] = ra[t][0]; d[1] = ra[t][1]; d[2] = ra[t][2]; d[3] = ra[t][3];
      const int n = B_NFAST ? (i % BT) : map_r(i);
      const int kq = B_NFAST ? (i / BT) : map_q(i);
      float* e = Bs + n * LD + 4 * kq;
      e[0] = rb[t][0]; e[1] = rb[t][1]; e[2] = rb[t][2]; e[3] = rb[t][3];
    }
  };
  auto compute = [&]() {
    const int kb = wave * 32;
#pragma unroll
    for (int kk = 0; kk < 32; kk += 2) {
      const float a = As[l31 * LD + kb + kk + lhi];
      const float b = Bs[l31 * LD + kb + kk + lhi];
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
    }
  };

  fetch(0, ra0, rb0);
  if (BK < K) fetch(BK, ra1, rb1);
  for (int k0 = 0; k0 < K; k0 += 2 * BK) {
    stage(ra0, rb0);
    __syncthreads();
    if (k0 + 2 * BK < K) fetch(k0 + 2 * BK, ra0, rb0);
    compute();
    __syncthreads();
    if (k0 + BK >= K) break;
    stage(ra1, rb1);
    __syncthreads();
    if (k0 + 3 * BK < K) fetch(k0 + 3 * BK, ra1, rb1);
    compute();
    __syncthreads();
  }
  // fixed-order reduction of the 4 K-partials: waves 1..3 park theirs in LDS, wave 0 adds 1,2,3
  float* red = smem;   // [3][16][64]
  if (wave > 0) {
#pragma unroll
    for (int r = 0; r < 16; ++r) red[((wave - 1) * 16 + r) * 64 + lane] = acc[r];
  }
  __syncthreads();
  if (wave == 0) {
#pragma unroll
    for (int w = 0; w < 3; ++w)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[r] += red[(w * 16 + r) * 64 + lane];
    wave_epilogue(acc, row0, col0 + l31, lhi, ep, M, N, pre);
  }
}

// ------------------------------------------------------------------------------------------------
// EXPERIMENT (round 5): k_gemm_nt_sk with its operand tiles brought in by LDS-DMA
// (global_load_lds_dwordx4: global -> LDS without passing through registers) instead of global -> VGPR -> ds_write.
// The DMA writes a wave's 64 x 16 bytes to CONSECUTIVE LDS addresses, so the tile rows cannot be padded; the bank
// conflicts of the fragment reads are avoided by swizzling the 16-byte granules through the per-lane SOURCE address:
//   LDS[row r][granule g'] = X[r][k0 + 4 (g' ^ (r & 15))]          (32 granules of 4 floats per 128-deep row)
// and the MFMA operands are read with ds_read_b128 (16 lanes = 16 rows at one logical granule hit 16 different bank
// groups) instead of 4 x ds_read_b32 on a 129-float padded row.  Two LDS stages; a stage's 8 DMA pieces per wave are
// waited for with a counted vmcnt, the barrier is the raw one (no vmcnt drain), so the next stage's pieces stay in flight
// under the MFMAs.  Same products in the same order as k_gemm_nt_sk (wave w owns k = 32 w .. 32 w + 31 of every chunk, in
// steps of two): bit-identical results.  Plain row-major operands only (16-byte aligned rows, K % 128 == 0).
// Selected by yolat_debug_gemm_sk_dma (tools/exp/lds_dma_bench.py: 0 = VGPR-staged kernel, 4 / 8 = waves of this one).
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ void yl_glds16(const void* gsrc, unsigned lds_dst) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep)
               : "v"(gsrc), "s"(lds_dst)
               : "memory");
}
template <int NW>      // waves per workgroup: each owns 128 / NW of every chunk's k (NW = 8: two waves per SIMD, one's MFMAs under the other's reads)
__global__ void __launch_bounds__(64 * NW) k_gemm_nt_sk_dma(const float* __restrict__ A, long lda, int rowsA,
                                                           const float* __restrict__ B, long ldb, int rowsB, Epilogue ep,
                                                           int M, int N, int K, long long* stamps) {
  constexpr int BT = 32, BK = 128, NP = 32 / NW, NG = 32 / NW;      // DMA pieces / 16-byte granules per wave and chunk
  __shared__ __attribute__((aligned(16))) float smem[2][2][BT * BK];        // [stage][A | B][row][128]: 64 KB
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, lhi = lane >> 5;
  int rt_, ct_;
  yl_xcd_tile(rt_, ct_);
  const int row0 = rt_ * BT, col0 = ct_ * BT;
  const EpiPre pre = epi_prefetch(ep, row0, col0 + l31, M, N);
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  const unsigned lds0 = (unsigned)(size_t)&smem[0][0][0];
  // this wave's pieces of a stage: piece i = NP wave + j -> operand i >> 4, row pair i & 15; lane -> (row, granule)
  const float* src[NP];
#pragma unroll
  for (int j = 0; j < NP; ++j) {
    const int i = NP * wave + j, op = i >> 4, pair = i & 15;
    const int r = 2 * pair + (lane >> 5), g = (lane & 31) ^ (r & 15);
    src[j] = op == 0 ? A + (long)yl_min(row0 + r, rowsA - 1) * lda + 4 * g : B + (long)yl_min(col0 + r, rowsB - 1) * ldb + 4 * g;
  }
  auto stage = [&](int s, int k0) {
#pragma unroll
    for (int j = 0; j < NP; ++j) {
      const int i = NP * wave + j, op = i >> 4, pair = i & 15;
      yl_glds16(src[j] + k0, lds0 + (unsigned)(((s * 2 + op) * BT * BK + pair * 256) * 4));
    }
  };
  auto compute = [&](int s) {
    const float* As = &smem[s][0][0];
    const float* Bs = &smem[s][1][0];
#pragma unroll
    for (int gi = 0; gi < NG; ++gi) {
      const int g = NG * wave + gi, pos = g ^ (l31 & 15);
      const float4 a4 = *reinterpret_cast<const float4*>(As + l31 * BK + 4 * pos);
      const float4 b4 = *reinterpret_cast<const float4*>(Bs + l31 * BK + 4 * pos);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(lhi ? a4.y : a4.x, lhi ? b4.y : b4.x, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(lhi ? a4.w : a4.z, lhi ? b4.w : b4.z, acc, 0, 0, 0);
    }
  };
  const int nch = K / BK;
  long long t0 = 0, t_wait = 0, t_comp = 0;
  const bool st_on = stamps != nullptr && tid == 0;
  stage(0, 0);
  for (int kc = 0; kc < nch; ++kc) {
    long long ta = 0, tb = 0, tc = 0;
    if (kc + 1 < nch) {
      stage((kc + 1) & 1, (kc + 1) * BK);
      if (st_on) ta = clock64();
      // this chunk's pieces have landed; the next chunk's stay in flight
      if (NP == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    } else {
      if (st_on) ta = clock64();
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");     // every wave's pieces of the chunk are in LDS
    if (st_on) tb = clock64();
    compute(kc & 1);
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");     // the stage may be overwritten (two iterations on)
    if (st_on) { tc = clock64(); t_wait += tb - ta; t_comp += tc - tb; if (kc == 0) t0 = ta; }
  }
  if (st_on) {
    const long o = 3 * (blockIdx.x + (long)gridDim.x * blockIdx.y);
    stamps[o] = t_wait; stamps[o + 1] = t_comp; stamps[o + 2] = clock64() - t0;
  }
  // fixed-order reduction of the NW K-partials: waves 1.. park theirs in LDS, wave 0 adds them in wave order
  float* red = &smem[0][0][0];   // [NW - 1][16][64]
  if (wave > 0) {
#pragma unroll
    for (int r = 0; r < 16; ++r) red[((wave - 1) * 16 + r) * 64 + lane] = acc[r];
  }
  __syncthreads();
  if (wave == 0) {
#pragma unroll
    for (int w = 0; w < NW - 1; ++w)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[r] += red[(w * 16 + r) * 64 + lane];
    wave_epilogue(acc, row0, col0 + l31, lhi, ep, M, N, pre);
  }
}
int yl_gemm_sk_dma_on();
long long* yl_gemm_sk_dma_stamps();

// ------------------------------------------------------------------------------------------------
// TN GEMM (weight gradients):  P[s][n][k] = sum_{r in split s} Y[r][n] * A[r][k]
// Output tile 64(n) x 64(k); 64 rows per LDS stage (32 MFMAs per wave between barriers; 32-row stages left the
// N-row weight gradients latency bound: 97 us for [64,64] over 200 k rows); grid = (n tiles, k tiles, splits).
// ------------------------------------------------------------------------------------------------
template <class YL, class AL>
__global__ void __launch_bounds__(256) k_gemm_tn(YL Yop, AL Aop, float* partial, float* dbpart,
                                                  int M, int Nout, int K, int rows_per_split) {
  constexpr int BR = 64, BT = 64, NT = BR * 16 / 256;   // rows per LDS stage, tile edge, float4 slots per thread
  __shared__ __attribute__((aligned(16))) float Ys[BR][BT];
  __shared__ __attribute__((aligned(16))) float As[BR][BT];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int wm = wave >> 1, wn = wave & 1;
  const int l31 = lane & 31, lhi = lane >> 5;
  const int n0 = blockIdx.x * BT, k0 = blockIdx.y * BT, s = blockIdx.z;
  const int r_begin = s * rows_per_split;
  int r_end = r_begin + rows_per_split;
  if (r_end > M) r_end = M;
  const bool fastY = (Yop.vec != 0) && (n0 + BT <= Nout);
  const bool fastA = (Aop.vec != 0) && (k0 + BT <= K);

  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  float dbacc = 0.f;
  const bool do_db = (dbpart != nullptr) && (blockIdx.y == 0);

  float ry[NT][4], rx[NT][4];
  auto fetch = [&](int r0) {
    if (fastY) {
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        const int i = tid + t * 256;
        Yop.template load4<true>(r0 + i / 16, n0 + 4 * (i % 16), ry[t]);
      }
    } else {
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        const int i = tid + t * 256;
        Yop.template load4<false>(r0 + i / 16, n0 + 4 * (i % 16), ry[t]);
      }
    }
    if (fastA) {
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        const int i = tid + t * 256;
        Aop.template load4<true>(r0 + i / 16, k0 + 4 * (i % 16), rx[t]);
      }
    } else {
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        const int i = tid + t * 256;
        Aop.template load4<false>(r0 + i / 16, k0 + 4 * (i % 16), rx[t]);
      }
    }
  };

  if (r_begin < r_end) fetch(r_begin);
  for (int r0 = r_begin; r0 < r_end; r0 += BR) {
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      const int i = tid + t * 256;
      const int r = i / 16, q = i % 16;
      const bool ok = (r0 + r) < r_end;          // rows are the reduction dimension: mask them
      *reinterpret_cast<float4*>(&Ys[r][4 * q]) =
          make_float4(ok ? ry[t][0] : 0.f, ok ? ry[t][1] : 0.f, ok ? ry[t][2] : 0.f, ok ? ry[t][3] : 0.f);
      *reinterpret_cast<float4*>(&As[r][4 * q]) =
          make_float4(ok ? rx[t][0] : 0.f, ok ? rx[t][1] : 0.f, ok ? rx[t][2] : 0.f, ok ? rx[t][3] : 0.f);
    }
    __syncthreads();
    if (r0 + BR < r_end) fetch(r0 + BR);
#pragma unroll
    for (int rr = 0; rr < BR; rr += 2) {
      const float a = Ys[rr + lhi][wm * 32 + l31];
      const float b = As[rr + lhi][wn * 32 + l31];
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
    }
    if (do_db && tid < BT) {
#pragma unroll
      for (int rr = 0; rr < BR; ++rr) dbacc += Ys[rr][tid];
    }
    __syncthreads();
  }
  float* P = partial + (long)s * Nout * K;
  const int kc = k0 + wn * 32 + l31;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int n = n0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhi;
    if (n < Nout && kc < K) P[(long)n * K + kc] = acc[r];
  }
  if (do_db && tid < BT && n0 + tid < Nout) dbpart[(long)s * Nout + n0 + tid] = dbacc;
}

// dst[i] (+)= sum_s partial[s][i]   (fixed order: 8 contiguous split groups summed in order, then the 8
// group sums in order).  32 consecutive elements x 8 split groups per workgroup; 8 independent loads in
// flight per thread — a one-thread-per-element loop over S=512 splits is a chain of 512 dependent
// L2 round trips (130 us for a 64x64 weight).
__device__ __forceinline__ void reduce_splits_body(long blk, const float* partial, long elems, int S, float* dst,
                                                   long ld_dst, int cols, int accumulate) {
  __shared__ float gs[8][33];
  const int e = threadIdx.x & 31, g = threadIdx.x >> 5;
  const long i = blk * 32 + e;
  const int per = (S + 7) / 8;
  const int t0 = g * per, t1 = (t0 + per < S) ? t0 + per : S;
  float s = 0.f;
  if (i < elems) {
    const float* p = partial + i;
    int t = t0;
    for (; t + 8 <= t1; t += 8) {
      float v[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = p[(long)(t + j) * elems];
#pragma unroll
      for (int j = 0; j < 8; ++j) s += v[j];
    }
    for (; t < t1; ++t) s += p[(long)t * elems];
  }
  gs[g][e] = s;
  __syncthreads();
  if (g == 0 && i < elems) {
    float tot = gs[0][e];
#pragma unroll
    for (int k = 1; k < 8; ++k) tot += gs[k][e];
    float* d = dst + (i / cols) * ld_dst + (i % cols);
    if (accumulate) tot += *d;
    *d = tot;
  }
}
static __global__ void __launch_bounds__(256) k_reduce_splits(const float* partial, long elems, int S,
                                                              float* dst, long ld_dst, int cols,
                                                              int accumulate) {
  reduce_splits_body(blockIdx.x, partial, elems, S, dst, ld_dst, cols, accumulate);
}
// two reductions with the same split count in one launch (a weight gradient and its bias gradient: the second one is a
// handful of workgroups that used to be a ~5 us launch of their own)
static __global__ void __launch_bounds__(256) k_reduce_splits2(const float* pa, long ea, float* da, long lda, int ca,
                                                               const float* pb, long eb, float* db, long ldb, int cb,
                                                               int S, int accumulate, int blocks_a) {
  if ((int)blockIdx.x < blocks_a) reduce_splits_body(blockIdx.x, pa, ea, S, da, lda, ca, accumulate);
  else reduce_splits_body(blockIdx.x - blocks_a, pb, eb, S, db, ldb, cb, accumulate);
}
// dW (+)= sum of `partial` [S][Nout*K];  db (+)= sum of `dbpart` [S][Nout] when db != NULL
static inline void yl_reduce_dw_db(hipStream_t st, const float* partial, long elems, int S, float* dW, long lddw, int K,
                                   const float* dbpart, float* db, long Nout, int accumulate) {
  const int ba = yl_cdiv(elems, 32);
  if (db != nullptr) {
    hipLaunchKernelGGL(k_reduce_splits2, dim3(ba + yl_cdiv(Nout, 32)), dim3(256), 0, st, partial, elems, dW, lddw, K,
                       dbpart, Nout, db, Nout, (int)Nout, S, accumulate, ba);
  } else {
    hipLaunchKernelGGL(k_reduce_splits, dim3(ba), dim3(256), 0, st, partial, elems, S, dW, lddw, K, accumulate);
  }
}

struct TnPlan { int S; int rows_per_split; };
static inline TnPlan yl_tn_plan(long M, long Nout, long K) {
  const long tiles = (long)yl_cdiv(Nout, 64) * yl_cdiv(K, 64);
  long S = 1024 / tiles;
  if (S < 1) S = 1;
  if (S > 512) S = 512;
  long rps = (M + S - 1) / S;
  rps = ((rps + 31) / 32) * 32;
  if (rps < 64) rps = 64;
  S = (M + rps - 1) / rps;
  if (S < 1) S = 1;
  TnPlan p; p.S = (int)S; p.rows_per_split = (int)rps;
  return p;
}

// ---- host-side operand constructors --------------------------------------------------------------
static inline DenseOp yl_dense(const float* p, long ld, long rows, long cols) {
  DenseOp d;
  d.p = p; d.ld = ld; d.rows = (int)rows; d.cols = (int)cols;
  d.scale = nullptr; d.shift = nullptr; d.floor = -INFINITY;
  d.vec = (ld % 4 == 0) && yl_aligned16(p);
  return d;
}
static inline DenseProOp yl_dense_pro(const float* p, long ld, long rows, long cols,
                                      const float* scale, const float* shift, int relu) {
  DenseProOp d;
  d.p = p; d.ld = ld; d.rows = (int)rows; d.cols = (int)cols;
  d.scale = scale; d.shift = shift; d.floor = relu ? 0.f : -INFINITY;
  d.vec = (ld % 4 == 0) && yl_aligned16(p) && yl_aligned16(scale) && yl_aligned16(shift);
  return d;
}
static inline HalfOp yl_half(const yl_bf16_t* p, long ld, long rows, long cols) {
  HalfOp d;
  d.p = p; d.ld = ld; d.rows = (int)rows; d.cols = (int)cols;
  d.scale = nullptr; d.shift = nullptr; d.floor = -INFINITY;
  d.vec = (ld % 4 == 0) && ((((uintptr_t)p) & 7) == 0);
  return d;
}
static inline HalfProOp yl_half_pro(const yl_bf16_t* p, long ld, long rows, long cols, const float* scale,
                                    const float* shift, int relu) {
  HalfProOp d;
  d.p = p; d.ld = ld; d.rows = (int)rows; d.cols = (int)cols;
  d.scale = scale; d.shift = shift; d.floor = relu ? 0.f : -INFINITY;
  d.vec = (ld % 4 == 0) && ((((uintptr_t)p) & 7) == 0) && yl_aligned16(scale) && yl_aligned16(shift);
  return d;
}
static inline EdgeOp yl_edge(const float* x, long ldx, long Cin, const int* src, const int* dst,
                             const float* attr, long E) {
  EdgeOp a;
  a.x = x; a.ldx = ldx; a.Cin = (int)Cin; a.src = src; a.dst = dst; a.attr = attr;
  a.rows = (int)E; a.cols = (int)(2 * Cin + 4);
  a.vec = (Cin % 4 == 0) && (ldx % 4 == 0) && yl_aligned16(x) && yl_aligned16(attr);
  return a;
}
