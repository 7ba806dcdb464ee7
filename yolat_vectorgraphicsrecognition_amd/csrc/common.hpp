// common.hpp — shared device helpers for libyolat_hip.so (gfx950 / CDNA4 only).
//
// Hardware model used throughout (MI355X): 64-lane wavefronts, 256-thread workgroups = 4 waves =
// one wave per SIMD, fp32-input MFMA v_mfma_f32_32x32x2_f32 (exact fp32 fma chain, 64 cycles per
// instruction per SIMD), LDS tiles padded by one dword so that the per-lane column reads of the
// MFMA A/B fragments (ds_read_b32, 2 x 32-lane groups) are bank-conflict free.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/yolat_hip.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));

#define YL_LAUNCH_CHECK()                          \
  do {                                             \
    hipError_t e__ = hipGetLastError();            \
    if (e__ != hipSuccess) return (int)e__;        \
  } while (0)

static inline int yl_cdiv(long a, long b) { return (int)((a + b - 1) / b); }
static inline bool yl_aligned16(const void* p) { return (((uintptr_t)p) & 15) == 0; }

// ------------------------------------------------------------------------------------------------
// Operand loaders.  load4(r, k, v) returns 4 consecutive k-elements of logical row r, zero beyond
// the matrix.  k is always a multiple of 4.
// ------------------------------------------------------------------------------------------------

// Row-major dense operand with optional per-column affine + ReLU prologue (BatchNorm1d+ReLU of the
// producer applied on the fly).
struct DenseOp {
  const float* p;
  long ld;
  int rows, cols;
  const float* scale;  // nullable
  const float* shift;
  int relu;
  int vec;  // 16-byte loads legal (ld % 4 == 0 and base aligned)

  __device__ __forceinline__ void load4(int r, int k, float v[4]) const {
    v[0] = v[1] = v[2] = v[3] = 0.f;
    if (r >= rows || k >= cols) return;
    const float* q = p + (long)r * ld + k;
    if (vec && k + 3 < cols) {
      const float4 t = *reinterpret_cast<const float4*>(q);
      v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
    } else {
#pragma unroll
      for (int j = 0; j < 4; ++j)
        if (k + j < cols) v[j] = q[j];
    }
    if (scale != nullptr) {
#pragma unroll
      for (int j = 0; j < 4; ++j)
        if (k + j < cols) v[j] = fmaf(v[j], scale[k + j], shift[k + j]);
    }
    if (relu) {
#pragma unroll
      for (int j = 0; j < 4; ++j) v[j] = fmaxf(v[j], 0.f);
    }
  }
};

// Dense operand used transposed: logical element (r, k) = p[k*ld + r].
struct TransOp {
  const float* p;
  long ld;
  int rows, cols;  // logical rows (fast in memory), logical cols
  __device__ __forceinline__ void load4(int r, int k, float v[4]) const {
#pragma unroll
    for (int j = 0; j < 4; ++j)
      v[j] = (r < rows && k + j < cols) ? p[(long)(k + j) * ld + r] : 0.f;
  }
};

// Edge feature rows of AttrRelativeEdgeConvGlobalPool2.message (torch_vertex.py:331), gathered on
// the fly:  row q (CSR slot) = [ x[dst_q] | x[src_q] - x[dst_q] | attr_q ],  K = 2*Cin + 4.
struct EdgeOp {
  const float* x;
  long ldx;
  int Cin;
  const int* src;
  const int* dst;
  const float* attr;  // [E,4] contiguous
  int E;
  int vec;  // Cin % 4 == 0, ldx % 4 == 0, x 16-byte aligned

  __device__ __forceinline__ float elem(int s, int d, int q, int k) const {
    if (k < Cin) return x[(long)d * ldx + k];
    if (k < 2 * Cin) return x[(long)s * ldx + (k - Cin)] - x[(long)d * ldx + (k - Cin)];
    if (k < 2 * Cin + 4) return attr[(long)q * 4 + (k - 2 * Cin)];
    return 0.f;
  }
  __device__ __forceinline__ void load4(int q, int k, float v[4]) const {
    v[0] = v[1] = v[2] = v[3] = 0.f;
    if (q >= E || k >= 2 * Cin + 4) return;
    const int s = src[q], d = dst[q];
    if (vec) {
      if (k < Cin) {
        const float4 t = *reinterpret_cast<const float4*>(x + (long)d * ldx + k);
        v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
      } else if (k < 2 * Cin) {
        const float4 a = *reinterpret_cast<const float4*>(x + (long)s * ldx + (k - Cin));
        const float4 b = *reinterpret_cast<const float4*>(x + (long)d * ldx + (k - Cin));
        v[0] = a.x - b.x; v[1] = a.y - b.y; v[2] = a.z - b.z; v[3] = a.w - b.w;
      } else {
        const float4 t = *reinterpret_cast<const float4*>(attr + (long)q * 4);
        v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
      }
    } else {
#pragma unroll
      for (int j = 0; j < 4; ++j) v[j] = elem(s, d, q, k + j);
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
  __device__ __forceinline__ void load4(int n, int k, float v[4]) const {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      float t = 0.f;
      if (n < 2 * Cin && k + j < C) {
        const float* row = W1 + (long)(k + j) * ldw;
        t = (n < Cin) ? row[n] - row[Cin + n] : row[n];
      }
      v[j] = t;
    }
  }
};

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
  float* stats;  // nullable: float2 [ceil(M/64)][N]  (sum, M2) of (acc + bias)
};

// ------------------------------------------------------------------------------------------------
// NT GEMM:  Y[M,N] = epi( A[M,K] . B[N,K]^T )   — 256 threads, 2x2 waves, wave tile (BM/2)x(BN/2)
// built from 32x32x2 fp32 MFMAs.  LDS tiles As[BM][BK+1], Bs[BN][BK+1].
//   MFMA operand layout (cdna_hip_programming.md §3): A: lane l holds A[i=l&31][k=l>>5];
//   B: lane l holds B[k=l>>5][j=l&31]; C/D: col = l&31, row = (reg&3) + 8*(reg>>2) + 4*(l>>5).
// B_NFAST: the B loader is contiguous along n (transposed operands) -> map consecutive threads to
// consecutive n when staging.
// ------------------------------------------------------------------------------------------------
template <int BM, int BN, int BK, class AL, class BL, bool B_NFAST>
__global__ void __launch_bounds__(256) k_gemm_nt(AL A, BL B, Epilogue ep, int M, int N, int K) {
  constexpr int WM = BM / 2, WN = BN / 2, TM = WM / 32, TN = WN / 32;
  constexpr int LD = BK + 1;
  constexpr int KQ = BK / 4;
  constexpr int STAT_FLOATS = (BM / 32) * BN + (BM / 64) * BN;
  constexpr int TILE_FLOATS = (BM + BN) * LD;
  __shared__ float smem[TILE_FLOATS > STAT_FLOATS ? TILE_FLOATS : STAT_FLOATS];
  float* As = smem;
  float* Bs = smem + BM * LD;

  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int wm = wave >> 1, wn = wave & 1;
  const int l31 = lane & 31, lhi = lane >> 5;
  const int row0 = blockIdx.x * BM, col0 = blockIdx.y * BN;

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  for (int k0 = 0; k0 < K; k0 += BK) {
    // ---- stage A tile
    for (int i = tid; i < BM * KQ; i += 256) {
      const int r = i / KQ, kq = i % KQ;
      float v[4];
      A.load4(row0 + r, k0 + 4 * kq, v);
      float* d = As + r * LD + 4 * kq;
      d[0] = v[0]; d[1] = v[1]; d[2] = v[2]; d[3] = v[3];
    }
    // ---- stage B tile
    for (int i = tid; i < BN * KQ; i += 256) {
      int n, kq;
      if (B_NFAST) { n = i % BN; kq = i / BN; } else { n = i / KQ; kq = i % KQ; }
      float v[4];
      B.load4(col0 + n, k0 + 4 * kq, v);
      float* d = Bs + n * LD + 4 * kq;
      d[0] = v[0]; d[1] = v[1]; d[2] = v[2]; d[3] = v[3];
    }
    __syncthreads();
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

  // ---- bias
  if (ep.bias != nullptr) {
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int c = col0 + wn * WN + j * 32 + l31;
      const float bv = (c < N) ? ep.bias[c] : 0.f;
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] += bv;
    }
  }

  // ---- training-mode BatchNorm partial statistics over 64-row groups (deterministic)
  if (ep.stats != nullptr) {
    float* red = smem;                      // [BM/32][BN]
    float* meanS = smem + (BM / 32) * BN;   // [BM/64][BN]
    // pass 1: sums
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        float s = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = row0 + wm * WM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhi;
          s += (row < M) ? acc[i][j][r] : 0.f;
        }
        s += __shfl_xor(s, 32);
        if (lhi == 0) red[(wm * TM + i) * BN + wn * WN + j * 32 + l31] = s;
      }
    __syncthreads();
    for (int i = tid; i < (BM / 64) * BN; i += 256) {
      const int g = i / BN, c = i % BN;
      int cnt = M - (row0 + 64 * g);
      cnt = cnt < 0 ? 0 : (cnt > 64 ? 64 : cnt);
      const float s = red[(2 * g) * BN + c] + red[(2 * g + 1) * BN + c];
      meanS[i] = cnt > 0 ? s / (float)cnt : 0.f;
    }
    __syncthreads();
    // pass 2: M2 around the 64-row group mean
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        const int g = (wm * WM + i * 32) / 64;
        const float mu = meanS[g * BN + wn * WN + j * 32 + l31];
        float s = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = row0 + wm * WM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhi;
          const float d = acc[i][j][r] - mu;
          s += (row < M) ? d * d : 0.f;
        }
        s += __shfl_xor(s, 32);
        if (lhi == 0) red[(wm * TM + i) * BN + wn * WN + j * 32 + l31] = s;
      }
    __syncthreads();
    // red now holds M2 per 32-row group.  One writer per (64-row group, column): the wave owning
    // the even 32-row group; the group sum is mean*cnt.
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        const int g32 = wm * TM + i;
        const int cl = wn * WN + j * 32 + l31;
        const int c = col0 + cl;
        const int rb = (row0 / 64) + (g32 >> 1);
        if ((g32 & 1) == 0 && lhi == 0 && c < N && (long)rb * 64 < M) {
          int cnt = M - rb * 64;
          cnt = cnt > 64 ? 64 : cnt;
          const float m2 = red[g32 * BN + cl] + red[(g32 + 1) * BN + cl];
          const float sum = meanS[(g32 >> 1) * BN + cl] * (float)cnt;
          float2* dst = reinterpret_cast<float2*>(ep.stats) + (long)rb * N + c;
          *dst = make_float2(sum, m2);
        }
      }
  }

  // ---- scale/shift/relu + store
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int c = col0 + wn * WN + j * 32 + l31;
    if (c >= N) continue;
    const float sc = ep.scale != nullptr ? ep.scale[c] : 1.f;
    const float sh = ep.scale != nullptr ? ep.shift[c] : 0.f;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = row0 + wm * WM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhi;
        if (row < M) {
          float v = acc[i][j][r];
          if (ep.scale != nullptr) v = fmaf(v, sc, sh);
          if (ep.relu) v = fmaxf(v, 0.f);
          float* y = ep.Y + (long)row * ep.ldy + c;
          if (ep.accumulate) v += *y;
          *y = v;
        }
      }
  }
}

// ------------------------------------------------------------------------------------------------
// TN GEMM (weight gradients):  P[s][n][k] = sum_{r in split s} Y[r][n] * A[r][k]
// Output tile 64(n) x 64(k); 32 rows per LDS stage; grid = (n tiles, k tiles, splits).
// ------------------------------------------------------------------------------------------------
template <class YL, class AL>
__global__ void __launch_bounds__(256) k_gemm_tn(YL Yop, AL Aop, float* partial, float* dbpart,
                                                  int M, int Nout, int K, int rows_per_split) {
  constexpr int BR = 32, BT = 64;
  __shared__ __attribute__((aligned(16))) float Ys[BR][BT];
  __shared__ __attribute__((aligned(16))) float As[BR][BT];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int wm = wave >> 1, wn = wave & 1;
  const int l31 = lane & 31, lhi = lane >> 5;
  const int n0 = blockIdx.x * BT, k0 = blockIdx.y * BT, s = blockIdx.z;
  const int r_begin = s * rows_per_split;
  int r_end = r_begin + rows_per_split;
  if (r_end > M) r_end = M;

  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  float dbacc = 0.f;
  const bool do_db = (dbpart != nullptr) && (blockIdx.y == 0);

  for (int r0 = r_begin; r0 < r_end; r0 += BR) {
    for (int i = tid; i < BR * (BT / 4); i += 256) {
      const int r = i / (BT / 4), q = i % (BT / 4);
      float v[4];
      const int rr = r0 + r;
      if (rr < r_end) Yop.load4(rr, n0 + 4 * q, v); else v[0] = v[1] = v[2] = v[3] = 0.f;
      *reinterpret_cast<float4*>(&Ys[r][4 * q]) = make_float4(v[0], v[1], v[2], v[3]);
      if (rr < r_end) Aop.load4(rr, k0 + 4 * q, v); else v[0] = v[1] = v[2] = v[3] = 0.f;
      *reinterpret_cast<float4*>(&As[r][4 * q]) = make_float4(v[0], v[1], v[2], v[3]);
    }
    __syncthreads();
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

// dst[i] (+)= sum_s partial[s][i]   (fixed order)
static __global__ void k_reduce_splits(const float* partial, long elems, int S, float* dst, long ld_dst,
                                int cols, int accumulate) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= elems) return;
  float s = 0.f;
  for (int t = 0; t < S; ++t) s += partial[(long)t * elems + i];
  float* d = dst + (i / cols) * ld_dst + (i % cols);
  if (accumulate) s += *d;
  *d = s;
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
