// conv_fused.hip — one AttrRelativeEdgeConvGlobalPool2 layer in eval mode as ONE persistent kernel
// (gcn_lib/sparse/torch_vertex.py:319-337 with BatchNorm folded):
//
//   out[i]   = mean_{e:(j->i)} relu(bn2(W2 . relu(bn1(W1 . [x_i, x_j - x_i, a_e] + b1)) + b2)) + Wr . x_i + br
//   s_out[i] = relu(bn_n(Wn . xn_i + bn))
//
// Structure (MI355X: 256 CUs, 160 KB LDS/CU, fp32 MFMA 32x32x2, one wave per SIMD = up to 512 VGPRs):
//   * grid = min(#node tiles, 256) persistent workgroups, one per CU (LDS footprint ~158 KB); each
//     loads the layer's five weight matrices into LDS ONCE (16-byte loads, all in flight together) and
//     then walks node tiles (32 destination nodes) with stride gridDim.x;
//   * a node tile owns the contiguous CSR edge range [row_ptr[n0], row_ptr[n0+32]); edges are processed
//     in chunks of 64.  The gather of chunk c+1 (x[dst], x[src], attr: 16 float4 per thread) is issued
//     into registers BEFORE the MFMAs of chunk c and only written to LDS after them, so its L2/HBM
//     latency hides under ~7k cycles of matrix work;
//   * per chunk: GEMM1 (K = 2Cin+4) -> BN/ReLU -> LDS, GEMM2 (K = 64) -> BN/ReLU -> LDS, then the mean
//     aggregation as a third MFMA: agg[32 nodes x 64] += Sel[32 x 64 edges] . M[64 edges x 64], where
//     Sel[n][e] = 1/deg(n) if edge e points to node n else 0.  The k order of the MFMA chain is the edge
//     order, so the sum runs in the same order as torch_scatter's CPU scatter_add — no atomics;
//   * the root term accumulates into the same accumulator (acc += X . Wr^T), the node branch is one more
//     32x64 tile.
// No [E,*] intermediate ever goes to HBM: algorithmic traffic = the gathers + one [N,64] write per output.
#include "common.hpp"

struct ConvArgs {
  const float* x; long ldx;       // [N,Cin] layer input (gathered and root term)
  const float* xn; long ldxn;     // [N,Cin] node-branch input
  int Cin, N, E;
  const int* row_ptr; const int* src; const int* dst; const float* attr;
  const float *W1, *b1, *s1, *t1, *W2, *b2, *s2, *t2, *Wr, *br, *Wn, *bn, *sn, *tn;
  float* f_out; long ldf;
  float* s_out; long lds;
  int ntiles;
  int wvec;                       // weights 16-byte aligned (host check)
  EdgeOp eop;                     // gather descriptor (built on the host)
  int dbg;                        // experiment switches (tools/exp/conv_bench.hip); 0 in production
};

int g_conv_dbg = 0;
#define CF_TN 32   // destination nodes per tile
#define CF_TE 64   // edges per chunk
#define CF_C 64    // output channels

static inline size_t conv_fused_lds_floats(int Cin) {
  const int K1P = (2 * Cin + 4 + 1) & ~1, LD1 = K1P + 1, CinP = (Cin + 1) & ~1, LDC = CinP + 1;
  const int LDF = LD1 > 65 ? LD1 : 65;   // the feature tile is re-used as the [64][65] message tile
  return (size_t)CF_C * LD1 + CF_C * 65 + 2 * CF_C * LDC + CF_TE * LDF + CF_TE * 65 + 2 * CF_TN * LDC +
         CF_TN * 65 + 128;
}

// 16-byte batched copy of a [rows][K] row-major matrix (K % 4 == 0, 16-B aligned) into LDS [rows][ld]
template <int NB>
__device__ __forceinline__ void stage_matrix_vec(const float* __restrict__ W, int rows, int K, float* dst,
                                                 int ld, int tid) {
  const int KQ = K >> 2, total = rows * KQ;
  for (int base = tid; base < total; base += 256 * NB) {
    float4 v[NB];
#pragma unroll
    for (int j = 0; j < NB; ++j) {
      const int i = yl_min(base + j * 256, total - 1);
      v[j] = *reinterpret_cast<const float4*>(W + (long)(i / KQ) * K + 4 * (i % KQ));
    }
#pragma unroll
    for (int j = 0; j < NB; ++j) {
      const int i = base + j * 256;
      if (i < total) {
        float* d = dst + (i / KQ) * ld + 4 * (i % KQ);
        d[0] = v[j].x; d[1] = v[j].y; d[2] = v[j].z; d[3] = v[j].w;
      }
    }
  }
}

__device__ __forceinline__ void stage_matrix_any(const float* __restrict__ W, int rows, int K, int KP,
                                                 float* dst, int ld, int tid) {
  const int total = rows * KP;
  for (int base = tid; base < total; base += 256 * 8) {
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int i = yl_min(base + j * 256, total - 1);
      v[j] = W[(long)(i / KP) * K + yl_min(i % KP, K - 1)];
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int i = base + j * 256;
      if (i < total) dst[(i / KP) * ld + (i % KP)] = ((i % KP) < K) ? v[j] : 0.f;
    }
  }
}

template <bool VEC64>
__global__ void __launch_bounds__(256) k_conv_fused_eval(ConvArgs a) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const int Cin = a.Cin, K1 = 2 * Cin + 4, K1P = (K1 + 1) & ~1, LD1 = K1P + 1;
  const int CinP = (Cin + 1) & ~1, LDC = CinP + 1;
  float* W1s = sm;                          // [64][LD1]
  float* W2s = W1s + CF_C * LD1;            // [64][65]
  float* Wrs = W2s + CF_C * 65;             // [64][LDC]
  float* Wns = Wrs + CF_C * LDC;            // [64][LDC]
  float* Fs = Wns + CF_C * LDC;             // [64][LD1]   edge features of the chunk; later M tile [64][65]
  float* H1s = Fs + CF_TE * (LD1 > 65 ? LD1 : 65);   // [64][65]   (re-used as the combine buffer of a tile)
  float* Xs = H1s + CF_TE * 65;             // [32][LDC]
  float* XNs = Xs + CF_TN * LDC;            // [32][LDC]
  float* Sel = XNs + CF_TN * LDC;           // [32][65]    aggregation operator of the chunk
  int* rps = reinterpret_cast<int*>(Sel + CF_TN * 65);   // [33] row pointers of the tile
  float* invd = reinterpret_cast<float*>(rps + 40);      // [32] 1/max(deg,1)

  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int wm = wave >> 1, wn = wave & 1, l31 = lane & 31, lhi = lane >> 5;
  const EdgeOp& eop = a.eop;
  constexpr bool vec64 = VEC64;   // Cin == 64 with 16-byte aligned rows; else the generic path (Cin <= 16)

  // ---- weights -> LDS, once per workgroup
  if (!(a.dbg & 2)) {
    if (vec64 && a.wvec) {
      stage_matrix_vec<9>(a.W1, CF_C, K1, W1s, LD1, tid);
      stage_matrix_vec<4>(a.W2, CF_C, CF_C, W2s, 65, tid);
      stage_matrix_vec<4>(a.Wr, CF_C, CF_C, Wrs, LDC, tid);
      stage_matrix_vec<4>(a.Wn, CF_C, CF_C, Wns, LDC, tid);
    } else {
      stage_matrix_any(a.W1, CF_C, K1, K1P, W1s, LD1, tid);
      stage_matrix_any(a.W2, CF_C, CF_C, CF_C, W2s, 65, tid);
      stage_matrix_any(a.Wr, CF_C, Cin, CinP, Wrs, LDC, tid);
      stage_matrix_any(a.Wn, CF_C, Cin, CinP, Wns, LDC, tid);
    }
  }
  // per-lane epilogue constants (the lane's output column is fixed by the MFMA C/D layout)
  const int ccol = wn * 32 + l31;
  const float b1c = a.b1[ccol], s1c = a.s1[ccol], t1c = a.t1[ccol];
  const float b2c = a.b2[ccol], s2c = a.s2[ccol], t2c = a.t2[ccol];
  const float brc = a.br[ccol];
  const float bnc = a.bn[ccol], snc = a.sn[ccol], tnc = a.tn[ccol];

  // ---- gather of one chunk into registers (issue) and registers -> LDS (commit)
  constexpr int NG = VEC64 ? 9 : 3;         // vec64: 8 patch float4 + 1 attr float4; generic: <= 3 slots
  float gv[NG][4];
  const int KQg = (K1P + 3) / 4;            // generic path: float4 slots per row (host: 64*KQg <= 3*256)
  auto gather_issue = [&](int qc) {
    if constexpr (VEC64) {
#pragma unroll
      for (int t = 0; t < 8; ++t) {
        const int i = tid + t * 256;
        const int r = 4 * ((i >> 5) & 15) + ((i & 31) >> 3), kq = 8 * (i >> 9) + (i & 7);
        eop.load4<true>(qc + r, 4 * kq, gv[t]);
      }
      eop.load4<true>(qc + (tid & 63), 128, gv[8]);
    } else {
#pragma unroll
      for (int t = 0; t < 3; ++t) {
        if (t * 256 < CF_TE * KQg) {            // uniform: only as many slot rounds as the row width needs
          const int i = yl_min(tid + t * 256, CF_TE * KQg - 1);
          eop.load4<false>(qc + i / KQg, 4 * (i % KQg), gv[t]);
        }
      }
    }
  };
  auto gather_commit = [&]() {
    if constexpr (VEC64) {
      // 32 consecutive lanes cover a 4-row x 8-float4 patch: full 128-B lines from global, and the
      // ds_write_b32 banks (5*row + 4*kq + j) mod 32 of a 32-lane group are all distinct (LD1 = 133)
#pragma unroll
      for (int t = 0; t < 8; ++t) {
        const int i = tid + t * 256;
        const int r = 4 * ((i >> 5) & 15) + ((i & 31) >> 3), kq = 8 * (i >> 9) + (i & 7);
        float* d = Fs + r * LD1 + 4 * kq;
        d[0] = gv[t][0]; d[1] = gv[t][1]; d[2] = gv[t][2]; d[3] = gv[t][3];
      }
      if (tid < CF_TE) {
        float* d = Fs + tid * LD1 + 128;
        d[0] = gv[8][0]; d[1] = gv[8][1]; d[2] = gv[8][2]; d[3] = gv[8][3];
      }
    } else {
#pragma unroll
      for (int t = 0; t < 3; ++t) {
        const int i = tid + t * 256;
        if (i < CF_TE * KQg) {
          float* d = Fs + (i / KQg) * LD1 + 4 * (i % KQg);
#pragma unroll
          for (int j = 0; j < 4; ++j)
            if (4 * (i % KQg) + j < K1P) d[j] = gv[t][j];
        }
      }
    }
  };

  int tile = blockIdx.x;
  int q0 = 0, q1 = 0;
  int pf_q = -1;                                         // chunk start whose gather sits in gv (-1: none)
  if (tile < a.ntiles) {
    q0 = a.row_ptr[tile * CF_TN];
    q1 = a.row_ptr[yl_min(tile * CF_TN + CF_TN, a.N)];
    if (q0 < q1 && !(a.dbg & 1)) { gather_issue(q0); pf_q = q0; }   // first chunk of the first tile
  }
  __syncthreads();                                       // weights visible

  for (; tile < a.ntiles; tile += gridDim.x) {
    const int n0 = tile * CF_TN;
    const int tnext = tile + gridDim.x;
    int q0n = 0, q1n = 0;                                // edge range of this workgroup's next tile
    if (tnext < a.ntiles) {
      q0n = a.row_ptr[tnext * CF_TN];
      q1n = a.row_ptr[yl_min(tnext * CF_TN + CF_TN, a.N)];
    }
    if (tid <= CF_TN) rps[tid] = a.row_ptr[yl_min(n0 + tid, a.N)];
    if (tid < CF_TN) {
      const int nn = yl_min(n0 + tid, a.N - 1);
      const int dg = a.row_ptr[nn + 1] - a.row_ptr[nn];
      invd[tid] = 1.f / (float)(dg > 1 ? dg : 1);
    }
    // node inputs of the tile (root term uses x, node branch uses xn), zero-padded columns; batched
    {
      const int total = CF_TN * CinP;
      for (int base = tid; base < total; base += 256 * 4) {
        float vx[4], vn[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int i = yl_min(base + j * 256, total - 1);
          const int r = i / CinP, k = i % CinP;
          const int n = yl_min(n0 + r, a.N - 1), kc = yl_min(k, Cin - 1);
          vx[j] = a.x[(long)n * a.ldx + kc];
          vn[j] = a.xn[(long)n * a.ldxn + kc];
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int i = base + j * 256;
          if (i < total) {
            const int r = i / CinP, k = i % CinP;
            Xs[r * LDC + k] = (k < Cin) ? vx[j] : 0.f;
            XNs[r * LDC + k] = (k < Cin) ? vn[j] : 0.f;
          }
        }
      }
    }
    // tile accumulator: rows = the 32 nodes, cols = this wave's 32 channels (wn); the two k-halves
    // (wm = 0: edges 0..31 of a chunk + root term, wm = 1: edges 32..63) are combined at the end
    f32x16 tacc;
#pragma unroll
    for (int r = 0; r < 16; ++r) tacc[r] = 0.f;
    __syncthreads();                                     // rps / invd / Xs published

    const int qend = (a.dbg & 1) ? q0 : q1;
    for (int qc = q0; qc < qend; qc += CF_TE) {
      if (pf_q != qc) gather_issue(qc);                  // not prefetched (previous tile had no edges)
      gather_commit();                                   // registers (issued one chunk ago) -> Fs
      // aggregation operator of the chunk: Sel[n][e] = 1/deg(n) if edge qc+e belongs to node n0+n
      {
        const int e = tid & 63, g8 = tid >> 6;
        const int q = qc + e;
        const int dl = (q < q1) ? (eop.dst[yl_min(q, eop.rows - 1)] - n0) : -1;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const int n = g8 * 8 + j;
          Sel[n * 65 + e] = (dl == n) ? invd[n] : 0.f;
        }
      }
      __syncthreads();
      // prefetch the next chunk (same tile, or the first chunk of this workgroup's next tile)
      {
        const int qn = qc + CF_TE;
        if (qn < qend) { gather_issue(qn); pf_q = qn; }
        else if (q0n < q1n && !(a.dbg & 1)) { gather_issue(q0n); pf_q = q0n; }
        else pf_q = -1;
      }
      if (a.dbg & 8) { __syncthreads(); continue; }
      // ---- GEMM1: [64 edges x K1] . W1^T -> 64x64 (2x2 waves, one 32x32 tile each)
      f32x16 acc;
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[r] = 0.f;
      {
        const float* ap = Fs + (wm * 32 + l31) * LD1 + lhi;
        const float* bp = W1s + (wn * 32 + l31) * LD1 + lhi;
        int kk = (a.dbg & 16) ? K1P : 0;
        for (; kk + 8 <= K1P; kk += 8) {      // 4 k-steps: all 8 LDS reads in flight before the MFMAs
          const float a0 = ap[kk], a1 = ap[kk + 2], a2 = ap[kk + 4], a3 = ap[kk + 6];
          const float c0 = bp[kk], c1 = bp[kk + 2], c2 = bp[kk + 4], c3 = bp[kk + 6];
          acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, c0, acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, c1, acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a2, c2, acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a3, c3, acc, 0, 0, 0);
        }
        for (; kk < K1P; kk += 2)
          acc = __builtin_amdgcn_mfma_f32_32x32x2f32(ap[kk], bp[kk], acc, 0, 0, 0);
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhi;
        H1s[row * 65 + ccol] = fmaxf(fmaf(acc[r] + b1c, s1c, t1c), 0.f);
      }
      __syncthreads();          // H1s complete; every wave is done reading Fs
      // ---- GEMM2: [64 x 64] . W2^T
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[r] = 0.f;
      {
        const float* ap = H1s + (wm * 32 + l31) * 65 + lhi;
        const float* bp = W2s + (wn * 32 + l31) * 65 + lhi;
#pragma unroll 4
        for (int kk = (a.dbg & 32) ? CF_C : 0; kk < CF_C; kk += 8) {
          const float a0 = ap[kk], a1 = ap[kk + 2], a2 = ap[kk + 4], a3 = ap[kk + 6];
          const float c0 = bp[kk], c1 = bp[kk + 2], c2 = bp[kk + 4], c3 = bp[kk + 6];
          acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, c0, acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, c1, acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a2, c2, acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a3, c3, acc, 0, 0, 0);
        }
      }
      float* Ms = Fs;           // [64][65] messages of the chunk
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhi;
        Ms[row * 65 + ccol] = fmaxf(fmaf(acc[r] + b2c, s2c, t2c), 0.f);
      }
      __syncthreads();
      // ---- mean aggregation as an MFMA: tacc[32 nodes][32 ch] += Sel[32][k-half] . Ms[k-half][32 ch]
      {
        const float* ap = Sel + l31 * 65 + wm * 32 + lhi;               // A[i = node][k = edge]
        const float* bp = Ms + (wm * 32 + lhi) * 65 + wn * 32 + l31;    // B[k = edge][j = channel]
#pragma unroll 4
        for (int kk = (a.dbg & 64) ? 32 : 0; kk < 32; kk += 8) {
          const float a0 = ap[kk], a1 = ap[kk + 2], a2 = ap[kk + 4], a3 = ap[kk + 6];
          const float c0 = bp[kk * 65], c1 = bp[(kk + 2) * 65], c2 = bp[(kk + 4) * 65], c3 = bp[(kk + 6) * 65];
          tacc = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, c0, tacc, 0, 0, 0);
          tacc = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, c1, tacc, 0, 0, 0);
          tacc = __builtin_amdgcn_mfma_f32_32x32x2f32(a2, c2, tacc, 0, 0, 0);
          tacc = __builtin_amdgcn_mfma_f32_32x32x2f32(a3, c3, tacc, 0, 0, 0);
        }
      }
      __syncthreads();          // Ms / Fs / Sel free for the next chunk
    }

    // ---- root term (waves wm = 0 accumulate X . Wr^T into tacc) and node branch (waves wm = 1)
    f32x16 nacc;
#pragma unroll
    for (int r = 0; r < 16; ++r) nacc[r] = 0.f;
    if (!(a.dbg & 4)) {
      const float* ap = (wm == 0 ? Xs : XNs) + l31 * LDC + lhi;
      const float* bp = (wm == 0 ? Wrs : Wns) + (wn * 32 + l31) * LDC + lhi;
      if (wm == 0) {
        for (int kk = 0; kk < CinP; kk += 2)
          tacc = __builtin_amdgcn_mfma_f32_32x32x2f32(ap[kk], bp[kk], tacc, 0, 0, 0);
      } else {
        for (int kk = 0; kk < CinP; kk += 2)
          nacc = __builtin_amdgcn_mfma_f32_32x32x2f32(ap[kk], bp[kk], nacc, 0, 0, 0);
      }
    }
    // combine the two k-halves of the aggregation: wm = 1 parks its partial in LDS, wm = 0 adds + stores
    float* Cmb = H1s;           // [32][65]
    if (wm == 1) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = (r & 3) + 8 * (r >> 2) + 4 * lhi;
        Cmb[row * 65 + ccol] = tacc[r];
        if (n0 + row < a.N)
          a.s_out[(long)(n0 + row) * a.lds + ccol] = fmaxf(fmaf(nacc[r] + bnc, snc, tnc), 0.f);
      }
    }
    __syncthreads();
    if (wm == 0) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = (r & 3) + 8 * (r >> 2) + 4 * lhi;
        if (n0 + row < a.N)
          a.f_out[(long)(n0 + row) * a.ldf + ccol] = tacc[r] + Cmb[row * 65 + ccol] + brc;
      }
    }
    __syncthreads();            // rps / Xs / Cmb reused by the next tile
    q0 = q0n; q1 = q1n;
  }
}

extern "C" int yolat_conv_eval_fused(const float* x, int64_t ldx, const float* xn, int64_t ldxn, int64_t N,
                                     int64_t Cin, const int32_t* row_ptr, const int32_t* src_csr,
                                     const int32_t* dst_csr, const float* attr_csr, int64_t E,
                                     const yolat_conv_eval* w, int64_t C, float* f_out, int64_t ldf,
                                     float* s_out, int64_t lds, yolat_stream_t stream) {
  if (!x || !xn || !row_ptr || !w || !f_out || !s_out || N <= 0 || E < 0 || Cin <= 0) return YOLAT_E_INVALID;
  if (C != CF_C || !(Cin == 64 || Cin <= 8)) return YOLAT_E_UNSUPPORTED;   // generic path: 64*ceil((2Cin+4)/4) <= 768 slots
  if (E > 0 && (!src_csr || !dst_csr || !attr_csr)) return YOLAT_E_INVALID;
  if (ldx < Cin || ldxn < Cin || ldf < C || lds < C || N >= (1LL << 31) - 64) return YOLAT_E_INVALID;
  ConvArgs a;
  a.x = x; a.ldx = ldx; a.xn = xn; a.ldxn = ldxn; a.Cin = (int)Cin; a.N = (int)N; a.E = (int)E;
  a.row_ptr = row_ptr; a.src = src_csr; a.dst = dst_csr; a.attr = attr_csr;
  a.W1 = w->W1; a.b1 = w->b1; a.s1 = w->s1; a.t1 = w->t1; a.W2 = w->W2; a.b2 = w->b2; a.s2 = w->s2; a.t2 = w->t2;
  a.Wr = w->Wr; a.br = w->br; a.Wn = w->Wn; a.bn = w->bn; a.sn = w->sn; a.tn = w->tn;
  a.f_out = f_out; a.ldf = ldf; a.s_out = s_out; a.lds = lds;
  a.ntiles = yl_cdiv(N, CF_TN);
  a.eop = yl_edge(x, ldx, Cin, src_csr, dst_csr, attr_csr, E > 0 ? E : 1);
  if (Cin == 64 && !a.eop.vec) return YOLAT_E_UNSUPPORTED;     // needs 16-byte aligned rows
  a.wvec = yl_aligned16(w->W1) && yl_aligned16(w->W2) && yl_aligned16(w->Wr) && yl_aligned16(w->Wn);
  a.dbg = g_conv_dbg;
  const size_t lds_bytes = conv_fused_lds_floats((int)Cin) * sizeof(float);
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k_conv_fused_eval<true>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e == hipSuccess)
      e = hipFuncSetAttribute(reinterpret_cast<const void*>(k_conv_fused_eval<false>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) return (int)e;
    attr_set = true;
  }
  int grid = a.ntiles < 256 ? a.ntiles : 256;
  if (Cin == 64 && a.eop.vec)
    hipLaunchKernelGGL(k_conv_fused_eval<true>, dim3(grid), dim3(256), lds_bytes, (hipStream_t)stream, a);
  else
    hipLaunchKernelGGL(k_conv_fused_eval<false>, dim3(grid), dim3(256), lds_bytes, (hipStream_t)stream, a);
  YL_LAUNCH_CHECK();
  return 0;
}
