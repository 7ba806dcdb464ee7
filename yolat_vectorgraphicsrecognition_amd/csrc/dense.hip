// dense.hip — nn.Linear / BatchNorm1d / ReLU of gcn_lib/sparse/torch_nn.py:50-71 on gfx950.
// Entry points: yolat_linear_fwd, yolat_linear_fwd_wt, yolat_linear_bwd_w, yolat_bn_finalize,
// yolat_bn_eval_coeffs, yolat_scale_shift_relu, yolat_bn_relu_bwd.
#include "common.hpp"
#include <type_traits>

// switch of the LDS-DMA skinny GEMM (tools/exp/lds_dma_bench.py, tests; not part of the C ABI header)
static int g_sk_dma = -1;      // -1: automatic (8-wave LDS-DMA kernel for K >= 1024), 0: never, 4 / 8: always (where it applies)
static long long* g_sk_dma_stamps = nullptr;
int yl_gemm_sk_dma_on() { return g_sk_dma; }
long long* yl_gemm_sk_dma_stamps() { return g_sk_dma_stamps; }
extern "C" void yolat_debug_gemm_sk_dma(int on, long long* stamps) { g_sk_dma = on; g_sk_dma_stamps = stamps; }

// ------------------------------------------------------------------------------------------------
// GEMM dispatch
// ------------------------------------------------------------------------------------------------
template <class AL, class BL, bool NFAST>
static int launch_gemm_nt(const AL& A, const BL& B, const Epilogue& ep, long M, long N, long K,
                          hipStream_t st) {
  if (M <= 0 || N <= 0) return 0;
  const long tiles64 = (long)yl_cdiv(M, 64) * yl_cdiv(N, 64);
  const long tiles128 = (long)yl_cdiv(M, 128) * yl_cdiv(N, 128);
  // measured on MI355X (tools/exp/gemm_bench.hip): 4096^3 136.7 TF/s with 128x128x32; for the
  // K=128 fusion GEMM 64x64x32 wins below ~2k 128-tiles (tail quantisation over 256 CUs).
  if (K >= 256 && tiles64 < 160) {
    // few rows, long K: split K across the 4 waves of a 32x32-tile workgroup
    dim3 grid(yl_cdiv(M, 32), yl_cdiv(N, 32));
    bool dma = false;
    if constexpr (std::is_same<AL, DenseOp>::value && std::is_same<BL, DenseOp>::value && !NFAST) {
      // the LDS-DMA experiment (common.hpp k_gemm_nt_sk_dma): plain operands with 16-byte rows, whole 128-deep chunks
      // measured at the cfg-2 classifier (P = 400, tools/exp/lds_dma_bench.py): 2304 -> 512: 20.7 us VGPR-staged, 20.1 us
      // LDS-DMA with 4 waves, 18.6 us with 8; 512 -> 256 (4 chunks): 8.9 / 9.2 / 9.3 us — the DMA pipeline needs a long K
      const int mode = yl_gemm_sk_dma_on() < 0 ? (K >= 1024 ? 8 : 0) : yl_gemm_sk_dma_on();
      if (mode && A.vec && B.vec && K % 128 == 0 && A.cols >= K && B.cols >= K) {
        dma = true;
        if (mode == 8)
          hipLaunchKernelGGL(k_gemm_nt_sk_dma<8>, grid, dim3(512), 0, st, A.p, A.ld, A.rows, B.p, B.ld, B.rows, ep, (int)M,
                             (int)N, (int)K, yl_gemm_sk_dma_stamps());
        else
          hipLaunchKernelGGL(k_gemm_nt_sk_dma<4>, grid, dim3(256), 0, st, A.p, A.ld, A.rows, B.p, B.ld, B.rows, ep, (int)M,
                             (int)N, (int)K, yl_gemm_sk_dma_stamps());
      }
    }
    if (!dma)
    hipLaunchKernelGGL((k_gemm_nt_sk<AL, BL, NFAST>), grid, dim3(256), 0, st, A, B, ep, (int)M,
                       (int)N, (int)K);
  } else if (N <= 64 || tiles128 < 2048) {
    dim3 grid(yl_cdiv(M, 64), yl_cdiv(N, 64));
    if (K <= 16)
      hipLaunchKernelGGL((k_gemm_nt<64, 64, 16, AL, BL, NFAST>), grid, dim3(256), 0, st, A, B, ep,
                         (int)M, (int)N, (int)K);
    else
      hipLaunchKernelGGL((k_gemm_nt<64, 64, 32, AL, BL, NFAST>), grid, dim3(256), 0, st, A, B, ep,
                         (int)M, (int)N, (int)K);
  } else {
    dim3 grid(yl_cdiv(M, 128), yl_cdiv(N, 128));
    hipLaunchKernelGGL((k_gemm_nt<128, 128, 32, AL, BL, NFAST>), grid, dim3(256), 0, st, A, B, ep,
                       (int)M, (int)N, (int)K);
  }
  YL_LAUNCH_CHECK();
  return 0;
}

// ------------------------------------------------------------------------------------------------
// Many-row 64 -> 64 Linear as a row STREAM (round 4): the second edge Linear of a training conv layer (torch_vertex.py:331
// nn.3 on the [E, 64] hidden activation, BatchNorm-1 + ReLU applied while loading, BatchNorm-2 statistics in the epilogue).
// On the generic 64 x 64 tiles (k_gemm_nt) it was one workgroup per tile: 18 750 workgroups at E = 1.2 M that each fetch
// the 16 KB weight from L2, stage, run 32 MFMAs per wave and drain — 178 us for 614 MB (0.43 of HBM, 0.35 of the fp32
// MFMA rate).  Here the weight sits in LDS for the whole persistent workgroup, the next tile's rows are in flight under
// the current tile's MFMAs, and the finished tile leaves through LDS with 16-byte stores: 131 us = 4.7 TB/s, the rate of a
// plain device copy on this part (profiles/r03_stream_bw.txt: 4.85 TB/s read + write; two tiles of rows in flight
// instead of one: 142 us).  Same products in the same
// order as the tile kernel (k ascending in steps of two) and the same statistics arithmetic (wave_epilogue): bit-identical
// outputs and statistics.
// ------------------------------------------------------------------------------------------------
// KH: K = 64 KH (the rows' K halves pass through one 64 x 64 LDS tile one after the other); WT: the weight is given
// transposed ([K, 64] row-major: yolat_linear_fwd_wt, the input-gradient Linears of the training backward); ACC: Y += .
// (the generic epilogue's order: (acc + bias) + old).  The a_scale prologue and the statistics are for KH == 1 only.
template <int KH, bool WT, bool ACC>
__global__ void __launch_bounds__(256) k_lin64_stream(const float* __restrict__ A, long lda, int M,
                                                      const float* __restrict__ a_scale, const float* __restrict__ a_shift,
                                                      float a_floor, const float* __restrict__ W, long ldw,
                                                      const float* __restrict__ bias, float* Y, long ldy,
                                                      float2* __restrict__ stats, int tiles_per_wg) {
  constexpr int LD = 65, LDW = 64 * KH + 1, LDO = 68;
  __shared__ float As[64 * LD], Ws[64 * LDW];
  __shared__ __attribute__((aligned(16))) float Os[64 * LDO];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, l31 = lane & 31, lhi = lane >> 5;
  const int wm = wave >> 1, wn = wave & 1;
  const int q = tid & 15, rb = tid >> 4;                 // staging role: columns 4q.., rows rb + 16 t
  const int ntiles = (M + 63) >> 6;
  const int t0 = blockIdx.x * tiles_per_wg, t1 = yl_min(ntiles, t0 + tiles_per_wg);
  // Ws[n][k] = the weight of output column n
#pragma unroll
  for (int h = 0; h < KH; ++h)
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const int r = rb + 16 * t;
      if (WT) {                                          // row r + 64 h of Wt = input channel k, columns n = 4q..
        const float4 w = *reinterpret_cast<const float4*>(W + (long)(r + 64 * h) * ldw + 4 * q);
        Ws[(4 * q + 0) * LDW + 64 * h + r] = w.x; Ws[(4 * q + 1) * LDW + 64 * h + r] = w.y;
        Ws[(4 * q + 2) * LDW + 64 * h + r] = w.z; Ws[(4 * q + 3) * LDW + 64 * h + r] = w.w;
      } else {
        const float4 w = *reinterpret_cast<const float4*>(W + (long)r * ldw + 64 * h + 4 * q);
        float* d = Ws + r * LDW + 64 * h + 4 * q;
        d[0] = w.x; d[1] = w.y; d[2] = w.z; d[3] = w.w;
      }
    }
  float4 as = make_float4(1.f, 1.f, 1.f, 1.f), ah = make_float4(0.f, 0.f, 0.f, 0.f);
  if (a_scale) { as = *reinterpret_cast<const float4*>(a_scale + 4 * q); ah = *reinterpret_cast<const float4*>(a_shift + 4 * q); }
  const float bv = bias ? bias[wn * 32 + l31] : 0.f;
  float4 ra[KH][4], ro[4];
  auto fetch = [&](int tile, int h) {
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const long e = yl_min(tile * 64 + rb + 16 * t, M - 1);
      ra[h][t] = *reinterpret_cast<const float4*>(A + e * lda + 64 * h + 4 * q);
    }
  };
  if (t0 < t1) {
#pragma unroll
    for (int h = 0; h < KH; ++h) fetch(t0, h);
  }
  for (int tile = t0; tile < t1; ++tile) {
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
    for (int h = 0; h < KH; ++h) {
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        float* a = As + (rb + 16 * t) * LD + 4 * q;
        if (a_scale) {
          a[0] = fmaxf(fmaf(ra[h][t].x, as.x, ah.x), a_floor); a[1] = fmaxf(fmaf(ra[h][t].y, as.y, ah.y), a_floor);
          a[2] = fmaxf(fmaf(ra[h][t].z, as.z, ah.z), a_floor); a[3] = fmaxf(fmaf(ra[h][t].w, as.w, ah.w), a_floor);
        } else {
          a[0] = ra[h][t].x; a[1] = ra[h][t].y; a[2] = ra[h][t].z; a[3] = ra[h][t].w;
        }
      }
      __syncthreads();                                   // As complete (and the previous tile's Os reads are done)
      if (tile + 1 < t1) fetch(tile + 1, h);             // in flight under the MFMAs
      if (ACC && h == 0) {
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          const long e = yl_min(tile * 64 + rb + 16 * t, M - 1);
          ro[t] = *reinterpret_cast<const float4*>(Y + e * ldy + 4 * q);
        }
      }
#pragma unroll 8
      for (int k = 0; k < 64; k += 2) {
        const float av = As[(wm * 32 + l31) * LD + k + lhi];
        const float wv = Ws[(wn * 32 + l31) * LDW + 64 * h + k + lhi];
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av, wv, acc, 0, 0, 0);
      }
      if (h + 1 < KH) __syncthreads();                   // every read of this half's As is done
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] += bv;
    const int row_base = tile * 64 + wm * 32;
    if (stats != nullptr) {                              // wave_epilogue's arithmetic (common.hpp), per 32-row group
      int cnt = M - row_base;
      cnt = cnt > 32 ? 32 : cnt;
      float sm = 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = row_base + (r & 3) + 8 * (r >> 2) + 4 * lhi;
        sm += (row < M) ? acc[r] : 0.f;
      }
      sm += __shfl_xor(sm, 32);
      const float mu = cnt > 0 ? sm / (float)cnt : 0.f;
      float m2 = 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = row_base + (r & 3) + 8 * (r >> 2) + 4 * lhi;
        const float d = acc[r] - mu;
        m2 += (row < M) ? d * d : 0.f;
      }
      m2 += __shfl_xor(m2, 32);
      if (lhi == 0 && cnt > 0) stats[(long)(row_base >> 5) * 64 + wn * 32 + l31] = make_float2(sm, m2);
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) Os[(wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhi) * LDO + wn * 32 + l31] = acc[r];
    __syncthreads();                                     // Os complete; every read of As is done
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const int r = rb + 16 * t;
      const long row = (long)tile * 64 + r;
      float4 o = *reinterpret_cast<const float4*>(Os + r * LDO + 4 * q);
      if (ACC) { o.x += ro[t].x; o.y += ro[t].y; o.z += ro[t].z; o.w += ro[t].w; }
      if (row < M) *reinterpret_cast<float4*>(Y + row * ldy + 4 * q) = o;
    }
  }
}

// launcher shared by yolat_linear_fwd (W [64, K]) and yolat_linear_fwd_wt (Wt [K, 64])
template <int KH, bool WT, bool ACC>
static int yl_launch_lin64_stream(const float* A, long lda, long M, const float* a_scale, const float* a_shift, int a_relu,
                                  const float* W, long ldw, const float* bias, float* Y, long ldy, float* stats,
                                  hipStream_t st) {
  const int ntiles = (int)yl_cdiv(M, 64);
  // two workgroups per CU: three fit (51 KB of LDS each) and the kernel alone is as fast with 768, but they fill a CU's
  // LDS and keep the side stream's kernels of the training step off it: cfg-5 step 7.21 (768) / 7.18 (512) / 7.45 ms (256)
  const int wgs = 512;
  const int per = yl_cdiv(ntiles, wgs);
  hipLaunchKernelGGL((k_lin64_stream<KH, WT, ACC>), dim3(yl_cdiv(ntiles, per)), dim3(256), 0, st, A, lda, (int)M, a_scale,
                     a_shift, a_relu ? 0.f : -INFINITY, W, ldw, bias, Y, ldy, reinterpret_cast<float2*>(stats), per);
  YL_LAUNCH_CHECK();
  return 0;
}

static bool yl_lin64_stream_ok(const float* A, int64_t lda, int64_t M, int64_t K, const float* a_scale, const float* a_shift,
                               const float* W, int64_t ldw, const float* bias, int64_t Nout, const float* o_scale, int o_relu,
                               const float* Y, int64_t ldy, int accumulate, const float* stats) {
  return K == 64 && Nout == 64 && M >= 65536 && stats != nullptr && bias != nullptr && o_scale == nullptr && !o_relu &&
         !accumulate && lda % 4 == 0 && ldw % 4 == 0 && ldy % 4 == 0 && yl_aligned16(A) && yl_aligned16(W) && yl_aligned16(Y) &&
         (((uintptr_t)stats) & 7) == 0 && (!a_scale || (yl_aligned16(a_scale) && yl_aligned16(a_shift)));
}

extern "C" int yolat_linear_fwd(const float* A, int64_t lda, int64_t M, int64_t K,
                                const float* a_scale, const float* a_shift, int a_relu,
                                const float* W, int64_t ldw, const float* bias, int64_t Nout,
                                const float* o_scale, const float* o_shift, int o_relu, float* Y,
                                int64_t ldy, int accumulate, float* stats, yolat_stream_t stream) {
  if (M < 0 || K <= 0 || Nout <= 0 || (M > 0 && (!A || !Y)) || !W) return YOLAT_E_INVALID;
  if (M >= (1LL << 31) || lda < K || ldw < K || ldy < Nout) return YOLAT_E_INVALID;
  if ((a_scale == nullptr) != (a_shift == nullptr)) return YOLAT_E_INVALID;
  if ((o_scale == nullptr) != (o_shift == nullptr)) return YOLAT_E_INVALID;
  if (a_relu && !a_scale) return YOLAT_E_INVALID;
  if (yl_lin64_stream_ok(A, lda, M, K, a_scale, a_shift, W, ldw, bias, Nout, o_scale, o_relu, Y, ldy, accumulate, stats)) {
    return yl_launch_lin64_stream<1, false, false>(A, (long)lda, (long)M, a_scale, a_shift, a_relu, W, (long)ldw, bias, Y,
                                                   (long)ldy, stats, (hipStream_t)stream);
  }
  DenseOp b = yl_dense(W, ldw, Nout, K);
  Epilogue ep;
  ep.bias = bias; ep.scale = o_scale; ep.shift = o_shift; ep.relu = o_relu;
  ep.Y = Y; ep.ldy = ldy; ep.accumulate = accumulate; ep.stats = stats; ep.seg = nullptr; ep.pool = nullptr; ep.ldpool = 0;
  if (a_scale != nullptr) {
    DenseProOp a = yl_dense_pro(A, lda, M, K, a_scale, a_shift, a_relu);
    return launch_gemm_nt<DenseProOp, DenseOp, false>(a, b, ep, M, Nout, K, (hipStream_t)stream);
  }
  DenseOp a = yl_dense(A, lda, M, K);
  return launch_gemm_nt<DenseOp, DenseOp, false>(a, b, ep, M, Nout, K, (hipStream_t)stream);
}

// Eval-mode fusion block + per-proposal max pooling in one kernel (arch:61-63 + arch:122):
//   pool[p, 0:Nout] = max over rows r of proposal p of relu((A[r] . W^T + bias)*scale + shift)
// The [M,Nout] activation is never written to HBM.  `pool` must be zero-filled beforehand.
extern "C" int yolat_linear_segmax_fwd(const float* A, int64_t lda, int64_t M, int64_t K, const float* W,
                                       int64_t ldw, const float* bias, int64_t Nout, const float* o_scale,
                                       const float* o_shift, const int32_t* node_seg, float* pool,
                                       int64_t ldpool, yolat_stream_t stream) {
  if (M <= 0 || K <= 0 || Nout <= 0 || !A || !W || !node_seg || !pool) return YOLAT_E_INVALID;
  if (M >= (1LL << 31) || lda < K || ldw < K || ldpool < Nout) return YOLAT_E_INVALID;
  if ((o_scale == nullptr) != (o_shift == nullptr)) return YOLAT_E_INVALID;
  DenseOp a = yl_dense(A, lda, M, K), b = yl_dense(W, ldw, Nout, K);
  Epilogue ep;
  ep.bias = bias; ep.scale = o_scale; ep.shift = o_shift; ep.relu = 1;
  ep.Y = nullptr; ep.ldy = 0; ep.accumulate = 0; ep.stats = nullptr;
  ep.seg = node_seg; ep.pool = pool; ep.ldpool = ldpool;
  dim3 grid(yl_cdiv(M, 64), yl_cdiv(Nout, 64));
  hipLaunchKernelGGL((k_gemm_nt<64, 64, 32, DenseOp, DenseOp, false>), grid, dim3(256), 0, (hipStream_t)stream,
                     a, b, ep, (int)M, (int)Nout, (int)K);
  YL_LAUNCH_CHECK();
  return 0;
}

// Node side of an eval-mode AttrRelativeEdgeConvGlobalPool2 layer in one launch (torch_vertex.py:324-327):
//   f_out = mean_{q in CSR row n} H2[q]  +  lin_r(f_in)          (propagate(aggr='mean') ; out += lin_r(x))
//   s_out = relu(sn * (Wn . s_in + bn) + tn)                      (mlp_node, BN folded)
// H2 == NULL or E == 0 skips the aggregation term.
extern "C" int yolat_node_side_eval(const float* f_in, int64_t ld_f, const float* s_in, int64_t ld_s, int64_t N,
                                    int64_t Cin, const float* Wr, const float* br, const float* Wn,
                                    const float* bn, const float* sn, const float* tn, const float* H2,
                                    int64_t ldh, const int32_t* row_ptr, int64_t E, int64_t C, float* f_out,
                                    int64_t ld_fo, float* s_out, int64_t ld_so, yolat_stream_t stream) {
  if (N <= 0 || Cin <= 0 || C <= 0 || !f_in || !s_in || !Wr || !Wn || !f_out || !s_out) return YOLAT_E_INVALID;
  if (C > 64) return YOLAT_E_UNSUPPORTED;
  if (N >= (1LL << 31) || ld_f < Cin || ld_s < Cin || ld_fo < C || ld_so < C) return YOLAT_E_INVALID;
  if ((sn == nullptr) != (tn == nullptr)) return YOLAT_E_INVALID;
  if (H2 != nullptr && E > 0 && (!row_ptr || ldh < C)) return YOLAT_E_INVALID;
  DenseOp a0 = yl_dense(f_in, ld_f, N, Cin), b0 = yl_dense(Wr, Cin, C, Cin);
  DenseOp a1 = yl_dense(s_in, ld_s, N, Cin), b1 = yl_dense(Wn, Cin, C, Cin);
  Epilogue e0, e1;
  e0.bias = br; e0.scale = nullptr; e0.shift = nullptr; e0.relu = 0;
  e0.Y = f_out; e0.ldy = ld_fo; e0.accumulate = 0; e0.stats = nullptr; e0.seg = nullptr; e0.pool = nullptr; e0.ldpool = 0;
  if (H2 != nullptr && E > 0) { e0.agg = H2; e0.ldagg = ldh; e0.agg_ptr = row_ptr; e0.agg_rows = (int)E; }
  e1.bias = bn; e1.scale = sn; e1.shift = tn; e1.relu = 1;
  e1.Y = s_out; e1.ldy = ld_so; e1.accumulate = 0; e1.stats = nullptr; e1.seg = nullptr; e1.pool = nullptr; e1.ldpool = 0;
  const dim3 grid(yl_cdiv(N, 64), 2);
  if (Cin <= 16)
    hipLaunchKernelGGL((k_gemm_nt_pair<64, 64, 16, DenseOp, DenseOp>), grid, dim3(256), 0, (hipStream_t)stream, a0,
                       b0, e0, a1, b1, e1, (int)N, (int)C, (int)Cin);
  else
    hipLaunchKernelGGL((k_gemm_nt_pair<64, 64, 32, DenseOp, DenseOp>), grid, dim3(256), 0, (hipStream_t)stream, a0,
                       b0, e0, a1, b1, e1, (int)N, (int)C, (int)Cin);
  YL_LAUNCH_CHECK();
  return 0;
}

static __global__ void k_conv_split_w1(const float* __restrict__ W1, int Cin, int C, float* Wuv, float* Wc4) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  const int K1 = 2 * Cin + 4;
  if (i < C * Cin) {
    const int c = i / Cin, k = i % Cin;
    const float a = W1[(long)c * K1 + k], b = W1[(long)c * K1 + Cin + k];
    Wuv[(long)c * Cin + k] = a - b;
    Wuv[(long)(C + c) * Cin + k] = b;
  }
  if (i < C * 4) Wc4[i] = W1[(long)(i / 4) * K1 + 2 * Cin + (i % 4)];
}

extern "C" int yolat_conv_split_w1(const float* W1, int64_t Cin, int64_t C, float* Wuv, float* Wc4,
                                   yolat_stream_t stream) {
  if (!W1 || !Wuv || !Wc4 || Cin <= 0 || C <= 0) return YOLAT_E_INVALID;
  const long n = C * (Cin > 4 ? Cin : 4);
  hipLaunchKernelGGL(k_conv_split_w1, dim3(yl_cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, W1, (int)Cin, (int)C,
                     Wuv, Wc4);
  YL_LAUNCH_CHECK();
  return 0;
}

int yl_build_node_uv(NodeUv* a, const float* f_in, int64_t ld_f, const float* s_in, int64_t ld_s, int64_t N,
                     int64_t Cin, const float* Wuv, const float* uv_bias, const float* Wr, const float* br,
                     const float* Wn, const float* bn, const float* sn, const float* tn, int64_t C, float* UV,
                     int64_t ld_uv, float* f_out, int64_t ld_fo, float* s_out, int64_t ld_so) {
  if (N <= 0 || Cin <= 0 || !f_in || !s_in || !Wuv || !Wr || !Wn || !UV || !f_out || !s_out) return YOLAT_E_INVALID;
  if (C != 64) return YOLAT_E_UNSUPPORTED;
  if (N >= (1LL << 31) || ld_f < Cin || ld_s < Cin || ld_uv < 2 * C || ld_fo < C || ld_so < C) return YOLAT_E_INVALID;
  if ((sn == nullptr) != (tn == nullptr)) return YOLAT_E_INVALID;
  a->af = yl_dense(f_in, ld_f, N, Cin); a->as = yl_dense(s_in, ld_s, N, Cin);
  a->wuv = yl_dense(Wuv, Cin, 2 * C, Cin); a->wr = yl_dense(Wr, Cin, C, Cin); a->wn = yl_dense(Wn, Cin, C, Cin);
  Epilogue e;
  e.bias = nullptr; e.scale = nullptr; e.shift = nullptr; e.relu = 0;
  e.Y = UV; e.ldy = ld_uv; e.accumulate = 0; e.stats = nullptr; e.seg = nullptr; e.pool = nullptr; e.ldpool = 0;
  a->euv = e; a->euv.bias = uv_bias;
  a->er = e; a->er.bias = br; a->er.Y = f_out; a->er.ldy = ld_fo;
  a->en = e; a->en.bias = bn; a->en.scale = sn; a->en.shift = tn; a->en.relu = 1; a->en.Y = s_out; a->en.ldy = ld_so;
  a->N = (int)N; a->C = (int)C; a->Cin = (int)Cin;
  return 0;
}

// Node side of the first conv layer of a large graph as an output stream: launcher of common.hpp's node3_smallk_body
namespace {
__global__ void __launch_bounds__(256) k_node3_smallk(NodeUv a) {
  node3_smallk_body(a, blockIdx.x);
}
// the same launch as the fall-back of the one-launch conv stack (conv_local.hip): dead unless the gate word holds its value
// (a few hundred persistent workgroups walking the row blocks: as a dead launch it costs a boundary, not a dispatch of
// thousands of workgroups)
__global__ void __launch_bounds__(256) k_node3_smallk_gated(NodeUv a, YlGate gate, int nblocks) {
  if (yl_gate_dead(gate)) return;
  for (int b = blockIdx.x; b < nblocks; b += gridDim.x) {
    node3_smallk_body(a, b);
    __syncthreads();
  }
}
bool n3_epi_ok(const Epilogue& e) {
  if (e.accumulate || e.stats || e.seg || e.pool || e.key64 || e.agg) return false;
  if ((e.scale == nullptr) != (e.shift == nullptr)) return false;
  if (e.Yh != nullptr) return e.ldy % 4 == 0 && (((uintptr_t)e.Yh) & 7) == 0;
  return e.Y != nullptr && e.ldy % 4 == 0 && yl_aligned16(e.Y);
}
}  // namespace

// the stream kernel applies (large N, tiny K, C = 64, plain operands, 16-byte aligned outputs)
bool yl_node3_smallk_shape_ok(const NodeUv& a) {       // what node3_smallk_body can compute at all (any N)
  return a.C == 64 && a.Cin >= 1 && a.Cin <= N3_KMAX && !a.af.scale && !a.as.scale && n3_epi_ok(a.euv) &&
         n3_epi_ok(a.er) && n3_epi_ok(a.en);
}
bool yl_node3_smallk_ok(const NodeUv& a) {             // ... and where it beats the 64 x 64 MFMA tiles as a launch of its own
  return a.N >= 32768 && yl_node3_smallk_shape_ok(a);
}
int yl_node3_smallk(const NodeUv& a, hipStream_t st, YlGate gate) {
  const int nb = yl_cdiv(a.N, N3_ROWS * N3_ITERS);
  if (gate.p) hipLaunchKernelGGL(k_node3_smallk_gated, dim3(nb < 2048 ? nb : 2048), dim3(256), 0, st, a, gate, nb);
  else hipLaunchKernelGGL(k_node3_smallk, dim3(yl_cdiv(a.N, N3_ROWS * N3_ITERS)), dim3(256), 0, st, a);
  YL_LAUNCH_CHECK();
  return 0;
}

extern "C" int yolat_node_uv_eval(const float* f_in, int64_t ld_f, const float* s_in, int64_t ld_s, int64_t N,
                                  int64_t Cin, const float* Wuv, const float* uv_bias, const float* Wr,
                                  const float* br, const float* Wn, const float* bn, const float* sn, const float* tn,
                                  int64_t C, float* UV, int64_t ld_uv, float* f_out, int64_t ld_fo, float* s_out,
                                  int64_t ld_so, yolat_stream_t stream) {
  NodeUv a;
  const int rc = yl_build_node_uv(&a, f_in, ld_f, s_in, ld_s, N, Cin, Wuv, uv_bias, Wr, br, Wn, bn, sn, tn, C, UV, ld_uv,
                                  f_out, ld_fo, s_out, ld_so);
  if (rc != 0) return rc;
  if (yl_node3_smallk_ok(a)) return yl_node3_smallk(a, (hipStream_t)stream);
  const dim3 grid(yl_cdiv(N, 64), 4);
  if (Cin <= 16) hipLaunchKernelGGL(k_gemm_nt_node3<16>, grid, dim3(256), 0, (hipStream_t)stream, a);
  else hipLaunchKernelGGL(k_gemm_nt_node3<32>, grid, dim3(256), 0, (hipStream_t)stream, a);
  YL_LAUNCH_CHECK();
  return 0;
}

// Eval-mode fusion stage in one launch (arch:61-69,122):
//   pool[p, 0:F]  = max over the rows of proposal p of relu((A.Wf^T + bf)*sf + tf)        (A = feats [N,D])
//   Ys[P, F]      = relu((S.Wfs^T + bfs)*sfs + tfs)                                       (S = mean(fsup) [P,D])
// `pool` must be zero-filled first (yolat_pool_prepare).
extern "C" int yolat_fusion_pair_eval(const float* A, int64_t lda, int64_t N, int64_t D, const float* Wf,
                                      const float* bf, const float* sf, const float* tf, int64_t F,
                                      const int32_t* node_seg, float* pool, int64_t ldpool, const float* S,
                                      int64_t lds, int64_t P, const float* Wfs, const float* bfs, const float* sfs,
                                      const float* tfs, float* Ys, int64_t ldys, yolat_stream_t stream) {
  if (N <= 0 || P <= 0 || D <= 0 || F <= 0 || !A || !Wf || !node_seg || !pool || !S || !Wfs || !Ys) return YOLAT_E_INVALID;
  if (N >= (1LL << 31) || lda < D || lds < D || ldpool < F || ldys < F) return YOLAT_E_INVALID;
  if ((sf == nullptr) != (tf == nullptr) || (sfs == nullptr) != (tfs == nullptr)) return YOLAT_E_INVALID;
  DenseOp a0 = yl_dense(A, lda, N, D), b0 = yl_dense(Wf, D, F, D);
  DenseOp a1 = yl_dense(S, lds, P, D), b1 = yl_dense(Wfs, D, F, D);
  Epilogue e0, e1;
  e0.bias = bf; e0.scale = sf; e0.shift = tf; e0.relu = 1;
  e0.Y = nullptr; e0.ldy = 0; e0.accumulate = 0; e0.stats = nullptr; e0.seg = node_seg; e0.pool = pool; e0.ldpool = ldpool;
  e1.bias = bfs; e1.scale = sfs; e1.shift = tfs; e1.relu = 1;
  e1.Y = Ys; e1.ldy = ldys; e1.accumulate = 0; e1.stats = nullptr; e1.seg = nullptr; e1.pool = nullptr; e1.ldpool = 0;
  const int tm0 = yl_cdiv(N, 64), tn0 = yl_cdiv(F, 64), tm1 = yl_cdiv(P, 64), tn1 = yl_cdiv(F, 64);
  const long total = (long)tm0 * tn0 + (((long)tm1 * tn1 + 7) & ~7L);
  if (total >= (1LL << 31)) return YOLAT_E_UNSUPPORTED;
  hipLaunchKernelGGL(k_gemm_nt_two, dim3((unsigned)total), dim3(256), 0, (hipStream_t)stream, a0, b0, e0, (int)N, (int)F,
                     (int)D, tm0, tn0, a1, b1, e1, (int)P, (int)F, (int)D, tm1, tn1);
  YL_LAUNCH_CHECK();
  return 0;
}

extern "C" int yolat_linear_fwd_wt(const float* A, int64_t lda, int64_t M, int64_t K,
                                   const float* Wt, int64_t ldw, int64_t Nout, float* Y,
                                   int64_t ldy, int accumulate, yolat_stream_t stream) {
  if (M < 0 || K <= 0 || Nout <= 0 || (M > 0 && (!A || !Y)) || !Wt) return YOLAT_E_INVALID;
  if (M >= (1LL << 31) || lda < K || ldw < Nout || ldy < Nout) return YOLAT_E_INVALID;
  if (Nout == 64 && (K == 64 || K == 128) && M >= 65536 && lda % 4 == 0 && ldw % 4 == 0 && ldy % 4 == 0 && yl_aligned16(A) &&
      yl_aligned16(Wt) && yl_aligned16(Y)) {
    // the input-gradient Linears of the training backward over the N nodes ([N, 64 | 128] . [64 | 128, 64], some of them
    // accumulating): row streams like the forward's (k_lin64_stream), same products in the same order as the tiles
    hipStream_t st = (hipStream_t)stream;
    if (K == 64)
      return accumulate ? yl_launch_lin64_stream<1, true, true>(A, (long)lda, (long)M, nullptr, nullptr, 0, Wt, (long)ldw, nullptr, Y, (long)ldy, nullptr, st)
                        : yl_launch_lin64_stream<1, true, false>(A, (long)lda, (long)M, nullptr, nullptr, 0, Wt, (long)ldw, nullptr, Y, (long)ldy, nullptr, st);
    return accumulate ? yl_launch_lin64_stream<2, true, true>(A, (long)lda, (long)M, nullptr, nullptr, 0, Wt, (long)ldw, nullptr, Y, (long)ldy, nullptr, st)
                      : yl_launch_lin64_stream<2, true, false>(A, (long)lda, (long)M, nullptr, nullptr, 0, Wt, (long)ldw, nullptr, Y, (long)ldy, nullptr, st);
  }
  DenseOp a = yl_dense(A, lda, M, K);
  TransOp b;
  b.p = Wt; b.ld = ldw; b.rows = (int)Nout; b.cols = (int)K; b.vec = 1;
  Epilogue ep;
  ep.bias = nullptr; ep.scale = nullptr; ep.shift = nullptr; ep.relu = 0;
  ep.Y = Y; ep.ldy = ldy; ep.accumulate = accumulate; ep.stats = nullptr; ep.seg = nullptr; ep.pool = nullptr; ep.ldpool = 0;
  return launch_gemm_nt<DenseOp, TransOp, true>(a, b, ep, M, Nout, K, (hipStream_t)stream);
}

// gemm_x6.hip: the weight gradient of a wide Linear on the bf16x6 GEMM
bool yl_bwd_w_x6_ok(int64_t M, int64_t Nout, int64_t K);
size_t yl_bwd_w_x6_work_elems(int64_t M, int64_t Nout, int64_t K);
int yl_bwd_w_x6(const float* dY, int64_t lddy, int64_t M, int64_t Nout, const float* A, int64_t lda, int64_t K, float* dW,
                int64_t lddw, float* db, int accumulate_db, float* work, hipStream_t st);

extern "C" size_t yolat_linear_bwd_w_work_elems(int64_t M, int64_t Nout, int64_t K) {
  TnPlan p = yl_tn_plan(M, Nout, K);
  const size_t tn = (size_t)p.S * (size_t)(Nout * K + Nout);
  if (yl_bwd_w_x6_ok(M, Nout, K)) { const size_t x = yl_bwd_w_x6_work_elems(M, Nout, K); return x > tn ? x : tn; }
  return tn;
}

extern "C" int yolat_linear_bwd_w(const float* dY, int64_t lddy, int64_t M, int64_t Nout,
                                  const float* A, int64_t lda, int64_t K, const float* a_scale,
                                  const float* a_shift, int a_relu, float* dW, int64_t lddw,
                                  float* db, int accumulate, float* partial,
                                  yolat_stream_t stream) {
  if (M < 0 || K <= 0 || Nout <= 0 || !dW || !partial || (M > 0 && (!dY || !A)))
    return YOLAT_E_INVALID;
  if (M >= (1LL << 31) || lddy < Nout || lda < K || lddw < K) return YOLAT_E_INVALID;
  if ((a_scale == nullptr) != (a_shift == nullptr)) return YOLAT_E_INVALID;
  if (a_relu && !a_scale) return YOLAT_E_INVALID;
  hipStream_t st = (hipStream_t)stream;
  // a wide weight with many rows (the classifier's first layer): the bf16x6 matrix-core GEMM, 16-byte aligned operands
  if (a_scale == nullptr && !accumulate && yl_bwd_w_x6_ok(M, Nout, K) && lda % 4 == 0 && yl_aligned16(A) &&
      yl_aligned16(partial))
    return yl_bwd_w_x6(dY, lddy, M, Nout, A, lda, K, dW, lddw, db, 0, partial, st);
  TnPlan p = yl_tn_plan(M, Nout, K);
  DenseOp y = yl_dense(dY, lddy, M, Nout);
  float* dbpart = db ? partial + (size_t)p.S * Nout * K : nullptr;
  dim3 grid(yl_cdiv(Nout, 64), yl_cdiv(K, 64), p.S);
  if (a_scale != nullptr) {
    DenseProOp a = yl_dense_pro(A, lda, M, K, a_scale, a_shift, a_relu);
    hipLaunchKernelGGL((k_gemm_tn<DenseOp, DenseProOp>), grid, dim3(256), 0, st, y, a, partial,
                       dbpart, (int)M, (int)Nout, (int)K, p.rows_per_split);
  } else {
    DenseOp a = yl_dense(A, lda, M, K);
    hipLaunchKernelGGL((k_gemm_tn<DenseOp, DenseOp>), grid, dim3(256), 0, st, y, a, partial, dbpart,
                       (int)M, (int)Nout, (int)K, p.rows_per_split);
  }
  YL_LAUNCH_CHECK();
  const long elems = Nout * K;
  yl_reduce_dw_db(st, partial, elems, p.S, dW, (long)lddw, (int)K, dbpart, db, (long)Nout, accumulate);
  YL_LAUNCH_CHECK();
  return 0;
}

// ------------------------------------------------------------------------------------------------
// bfloat16-STORED operands (training with bf16 storage of the [E,*] activations and their gradients, the mode
// BASELINE.json configs[4] names): the same fp32-accumulating tile kernels, elements converted while loading /
// rounded (nearest-even) while storing; BatchNorm statistics are taken from the fp32 accumulators.
// ------------------------------------------------------------------------------------------------
// dW [Nout,K] (+= when accumulate) = dY^T . pro(A),  db = column sums of dY;  dY bf16 [M,Nout];  A bf16 (a_is_half,
// optional BatchNorm+ReLU prologue) or fp32 [M,K].  `partial`: yolat_linear_bwd_w_work_elems(M, Nout, K) floats.
extern "C" int yolat_linear_bwd_w_h(const uint16_t* dY, int64_t lddy, int64_t M, int64_t Nout, const void* A,
                                    int a_is_half, int64_t lda, int64_t K, const float* a_scale, const float* a_shift,
                                    int a_relu, float* dW, int64_t lddw, float* db, int accumulate, float* partial,
                                    yolat_stream_t stream) {
  if (M <= 0 || K <= 0 || Nout <= 0 || !dW || !partial || !dY || !A) return YOLAT_E_INVALID;
  if (M >= (1LL << 31) || lddy < Nout || lda < K || lddw < K) return YOLAT_E_INVALID;
  if ((a_scale == nullptr) != (a_shift == nullptr) || (a_relu && !a_scale) || (a_scale && !a_is_half)) return YOLAT_E_INVALID;
  hipStream_t st = (hipStream_t)stream;
  TnPlan p = yl_tn_plan(M, Nout, K);
  HalfOp y = yl_half(dY, lddy, M, Nout);
  float* dbpart = db ? partial + (size_t)p.S * Nout * K : nullptr;
  dim3 grid(yl_cdiv(Nout, 64), yl_cdiv(K, 64), p.S);
  if (a_is_half && a_scale != nullptr) {
    HalfProOp a = yl_half_pro(reinterpret_cast<const yl_bf16_t*>(A), lda, M, K, a_scale, a_shift, a_relu);
    hipLaunchKernelGGL((k_gemm_tn<HalfOp, HalfProOp>), grid, dim3(256), 0, st, y, a, partial, dbpart, (int)M, (int)Nout,
                       (int)K, p.rows_per_split);
  } else if (a_is_half) {
    HalfOp a = yl_half(reinterpret_cast<const yl_bf16_t*>(A), lda, M, K);
    hipLaunchKernelGGL((k_gemm_tn<HalfOp, HalfOp>), grid, dim3(256), 0, st, y, a, partial, dbpart, (int)M, (int)Nout,
                       (int)K, p.rows_per_split);
  } else {
    DenseOp a = yl_dense(reinterpret_cast<const float*>(A), lda, M, K);
    hipLaunchKernelGGL((k_gemm_tn<HalfOp, DenseOp>), grid, dim3(256), 0, st, y, a, partial, dbpart, (int)M, (int)Nout,
                       (int)K, p.rows_per_split);
  }
  YL_LAUNCH_CHECK();
  const long elems = Nout * K;
  yl_reduce_dw_db(st, partial, elems, p.S, dW, (long)lddw, (int)K, dbpart, db, (long)Nout, accumulate);
  YL_LAUNCH_CHECK();
  return 0;
}

// ------------------------------------------------------------------------------------------------
// BatchNorm1d statistics (training mode) from the 32-row (sum, M2-about-the-group-mean) partials the GEMM epilogues
// write, reduced in fp64.  Level 1: workgroup g reduces a contiguous range of row groups (64 columns x 16 partitions per
// workgroup) to one triple per column.  Level 2: one workgroup adds the level-1 triples in order and emits mean / invstd /
// scale / shift and the running-stat update.  Every summation order is fixed => deterministic.
// The triple is (n, S = sum x, R = sum x^2) with the group's contribution to R = M2_g + S_g^2 / cnt_g, so that reducing
// is three fp64 additions per partial; M2 = R - S^2 / n once at the end.  (The pairwise Chan update it replaces needs
// three fp64 divisions per merge, ~1200 cycles each on this part: a chain of 31 of them made every BatchNorm finalize a
// 9 - 15 us launch.)  In fp64 the cancellation in R - S^2 / n costs a relative error of the variance of about
// 2^-53 (1 + mean^2 / var): below fp32 resolution unless |mean| > 10^4 standard deviations.
// ------------------------------------------------------------------------------------------------
#define BN_L1_MAX 128
struct Chan { double n, mean, m2; };
// (n, S, R = sum x^2) -> (n, mean, M2)
__device__ __forceinline__ Chan bn_chan(double n, double S, double R) {
  Chan a;
  a.n = n;
  a.mean = n > 0.0 ? S / n : 0.0;
  const double m2 = R - S * a.mean;
  a.m2 = m2 > 0.0 ? m2 : 0.0;
  return a;
}

__device__ __forceinline__ void bn_emit(const Chan& a, int c, const float* gamma, const float* beta, float* running_mean,
                                        float* running_var, float momentum, float eps, float* save_mean,
                                        float* save_invstd, float* scale, float* shift) {
  const double var_b = a.n > 0.0 ? a.m2 / a.n : 0.0;                 // biased: used to normalise
  const double var_u = a.n > 1.0 ? a.m2 / (a.n - 1.0) : var_b;       // unbiased: running update
  const float invstd = (float)(1.0 / sqrt(var_b + (double)eps));
  const float meanf = (float)a.mean;
  save_mean[c] = meanf;
  save_invstd[c] = invstd;
  const float sc = gamma[c] * invstd;
  scale[c] = sc;
  shift[c] = beta[c] - meanf * sc;
  if (running_mean != nullptr) {
    running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * meanf;
    running_var[c] = (1.f - momentum) * running_var[c] + momentum * (float)var_u;
  }
}

// FINAL (a single level-1 workgroup per column block, i.e. M <= 8192 rows: the per-proposal BatchNorms): the level-2
// step would merge ONE triple into an empty accumulator — exactly the triple itself — so the coefficients are emitted
// here and the second launch is dropped; same values bit for bit.
template <bool FINAL>
__global__ void __launch_bounds__(1024) k_bn_merge_l1(const float2* stats, long M, int C, long nb,
                                                      long per_wg, double* l1, const float* gamma, const float* beta,
                                                      float* running_mean, float* running_var, float momentum,
                                                      float eps, float* save_mean, float* save_invstd, float* scale,
                                                      float* shift) {
  __shared__ double s_n[16][64], s_mean[16][64], s_m2[16][64];
  const int cl = threadIdx.x & 63, part = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + cl;
  const long g0 = (long)blockIdx.y * per_wg;
  long g1 = g0 + per_wg;
  if (g1 > nb) g1 = nb;
  const long per = (g1 - g0 + 15) / 16;
  long b0 = g0 + part * per, b1 = b0 + per;
  if (b1 > g1) b1 = g1;
  double an = 0.0, as = 0.0, ar = 0.0;
  if (c < C) {
    for (long b = b0; b < b1; b += 8) {                     // 8 loads in flight, added in order
      float2 t[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) t[k] = stats[(b + k < b1 ? b + k : b1 - 1) * C + c];
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        if (b + k < b1) {
          long cnt = M - (b + k) * 32;
          if (cnt > 32) cnt = 32;
          const double sg = (double)t[k].x;
          an += (double)cnt;
          as += sg;
          ar += (double)t[k].y + (cnt == 32 ? sg * sg * 0.03125 : sg * sg / (double)cnt);
        }
      }
    }
  }
  s_n[part][cl] = an; s_mean[part][cl] = as; s_m2[part][cl] = ar;
  __syncthreads();
  if (part == 0 && c < C) {
    for (int p = 1; p < 16; ++p) { an += s_n[p][cl]; as += s_mean[p][cl]; ar += s_m2[p][cl]; }
    if (FINAL) {
      bn_emit(bn_chan(an, as, ar), c, gamma, beta, running_mean, running_var, momentum, eps, save_mean, save_invstd, scale,
              shift);
    } else {
      double* o = l1 + ((long)blockIdx.y * C + c) * 3;
      o[0] = an; o[1] = as; o[2] = ar;
    }
  }
}

// 64 columns x 16 partitions per workgroup: partition p adds a contiguous range of the level-1 triples in order,
// partition 0 then adds the 16 partials in order
__global__ void __launch_bounds__(1024) k_bn_finalize_l2(const double* l1, int G, int C, const float* gamma,
                                                         const float* beta, float* running_mean, float* running_var,
                                                         float momentum, float eps, float* save_mean,
                                                         float* save_invstd, float* scale, float* shift) {
  __shared__ double s_n[16][64], s_mean[16][64], s_m2[16][64];
  const int cl = threadIdx.x & 63, part = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + cl;
  const int per = (G + 15) / 16;
  const int g0 = part * per, g1 = yl_min(G, g0 + per);
  double an = 0.0, as = 0.0, ar = 0.0;
  if (c < C) {
    for (int g = g0; g < g1; g += 4) {                       // 12 loads in flight, added in order
      double v[4][3];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const double* o = l1 + ((long)(g + k < g1 ? g + k : g1 - 1) * C + c) * 3;
        v[k][0] = o[0]; v[k][1] = o[1]; v[k][2] = o[2];
      }
#pragma unroll
      for (int k = 0; k < 4; ++k)
        if (g + k < g1) { an += v[k][0]; as += v[k][1]; ar += v[k][2]; }
    }
  }
  s_n[part][cl] = an; s_mean[part][cl] = as; s_m2[part][cl] = ar;
  __syncthreads();
  if (part != 0 || c >= C) return;
  for (int p = 1; p < 16; ++p) { an += s_n[p][cl]; as += s_mean[p][cl]; ar += s_m2[p][cl]; }
  bn_emit(bn_chan(an, as, ar), c, gamma, beta, running_mean, running_var, momentum, eps, save_mean, save_invstd, scale, shift);
}

static inline void bn_l1_plan(long M, long* nb, long* G, long* per_wg) {
  *nb = (M + 31) / 32;
  // <= 8192 rows: one level-1 workgroup (the fused finalize); above that >= 64 row groups (2048 rows) per level-1
  // workgroup, up to BN_L1_MAX of them (E = 210 k: 26 workgroups of 256 groups took 13 us, 103 of 64 take ~7)
  long g = (*nb <= 256) ? 1 : (*nb + 63) / 64;
  if (g > BN_L1_MAX) g = BN_L1_MAX;
  if (g < 1) g = 1;
  *per_wg = (*nb + g - 1) / g;
  *G = (*nb + *per_wg - 1) / *per_wg;
}

extern "C" size_t yolat_bn_stats_elems(int64_t M, int64_t C) {
  // fp32 elements: float2 partials per 32-row group + (8-byte aligned) level-1 triples in fp64
  long nb, G, per;
  bn_l1_plan(M, &nb, &G, &per);
  return (size_t)(2 * nb * C + 2 + 2 * 3 * BN_L1_MAX * C);
}

extern "C" int yolat_bn_finalize(const float* stats, int64_t M, int64_t C, const float* gamma,
                                 const float* beta, float* running_mean, float* running_var,
                                 float momentum, float eps, float* save_mean, float* save_invstd,
                                 float* scale, float* shift, yolat_stream_t stream) {
  if (!stats || M <= 0 || C <= 0 || !gamma || !beta || !save_mean || !save_invstd || !scale || !shift)
    return YOLAT_E_INVALID;
  if ((running_mean == nullptr) != (running_var == nullptr)) return YOLAT_E_INVALID;
  long nb, G, per;
  bn_l1_plan(M, &nb, &G, &per);
  size_t off = (size_t)(2 * nb * C);
  off = (off + 1) & ~(size_t)1;                       // 8-byte alignment for the fp64 triples
  double* l1 = reinterpret_cast<double*>(const_cast<float*>(stats) + off);
  hipStream_t st = (hipStream_t)stream;
  if (G == 1) {
    hipLaunchKernelGGL(k_bn_merge_l1<true>, dim3(yl_cdiv(C, 64), 1), dim3(1024), 0, st,
                       reinterpret_cast<const float2*>(stats), (long)M, (int)C, nb, per, l1, gamma, beta, running_mean,
                       running_var, momentum, eps, save_mean, save_invstd, scale, shift);
    YL_LAUNCH_CHECK();
    return 0;
  }
  hipLaunchKernelGGL(k_bn_merge_l1<false>, dim3(yl_cdiv(C, 64), (unsigned)G), dim3(1024), 0, st,
                     reinterpret_cast<const float2*>(stats), (long)M, (int)C, nb, per, l1, gamma, beta, running_mean,
                     running_var, momentum, eps, save_mean, save_invstd, scale, shift);
  YL_LAUNCH_CHECK();
  hipLaunchKernelGGL(k_bn_finalize_l2, dim3(yl_cdiv(C, 64)), dim3(1024), 0, st, l1, (int)G, (int)C,
                     gamma, beta, running_mean, running_var, momentum, eps, save_mean, save_invstd,
                     scale, shift);
  YL_LAUNCH_CHECK();
  return 0;
}

__global__ void k_bn_eval_coeffs(const float* gamma, const float* beta, const float* rm,
                                 const float* rv, float eps, int C, float* scale, float* shift) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  const float invstd = 1.f / sqrtf(rv[c] + eps);
  const float sc = gamma[c] * invstd;
  scale[c] = sc;
  shift[c] = beta[c] - rm[c] * sc;
}

extern "C" int yolat_bn_eval_coeffs(const float* gamma, const float* beta,
                                    const float* running_mean, const float* running_var, float eps,
                                    int64_t C, float* scale, float* shift, yolat_stream_t stream) {
  if (!gamma || !beta || !running_mean || !running_var || !scale || !shift || C <= 0)
    return YOLAT_E_INVALID;
  hipLaunchKernelGGL(k_bn_eval_coeffs, dim3(yl_cdiv(C, 256)), dim3(256), 0, (hipStream_t)stream,
                     gamma, beta, running_mean, running_var, eps, (int)C, scale, shift);
  YL_LAUNCH_CHECK();
  return 0;
}

// ------------------------------------------------------------------------------------------------
// Z = relu(Y*scale + shift) — elementwise, grid-stride over rows, threads along columns.
// ------------------------------------------------------------------------------------------------
__global__ void k_scale_shift_relu(const float* Y, long ldy, long M, int C, const float* scale,
                                   const float* shift, int relu, float* Z, long ldz) {
  const int c = blockIdx.x * 64 + (threadIdx.x & 63);
  if (c >= C) return;
  const float sc = scale ? scale[c] : 1.f, sh = scale ? shift[c] : 0.f;
  for (long r = (long)blockIdx.y * 4 + (threadIdx.x >> 6); r < M; r += (long)gridDim.y * 4) {
    float v = fmaf(Y[r * ldy + c], sc, sh);
    if (relu) v = fmaxf(v, 0.f);
    Z[r * ldz + c] = v;
  }
}

extern "C" int yolat_scale_shift_relu(const float* Y, int64_t ldy, int64_t M, int64_t C,
                                      const float* scale, const float* shift, int relu, float* Z,
                                      int64_t ldz, yolat_stream_t stream) {
  if (M < 0 || C <= 0 || (M > 0 && (!Y || !Z))) return YOLAT_E_INVALID;
  if (M == 0) return 0;
  int gy = yl_cdiv(M, 4);
  if (gy > 2048) gy = 2048;
  hipLaunchKernelGGL(k_scale_shift_relu, dim3(yl_cdiv(C, 64), gy), dim3(256), 0,
                     (hipStream_t)stream, Y, (long)ldy, (long)M, (int)C, scale, shift, relu, Z,
                     (long)ldz);
  YL_LAUNCH_CHECK();
  return 0;
}

// ------------------------------------------------------------------------------------------------
// Backward of Z = relu(BN_train(Y)).
//  k_bn_bwd_partial: per 256-row block, per column: (sum dyh, sum dyh*xhat)
//  k_bn_bwd_finalize: fp64 ordered sum over blocks -> dgamma, dbeta, coef = (s1/M, s2/M)
//  k_bn_bwd_apply:   dY = scale*(dyh - c1 - xhat*c2)
// ------------------------------------------------------------------------------------------------
#define BNB_ROWS 256
__global__ void __launch_bounds__(256) k_bn_bwd_partial(const float* dZ, long lddz, const float* Y,
                                                        long ldy, long M, int C, const float* mean,
                                                        const float* invstd, const float* scale,
                                                        const float* shift, int relu,
                                                        float2* part) {
  __shared__ float2 red[4][64];
  const int cl = threadIdx.x & 63, ty = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + cl;
  const long r0 = (long)blockIdx.y * BNB_ROWS;
  float s1 = 0.f, s2 = 0.f;
  if (c < C) {
    const float mu = mean[c], is = invstd[c], sc = scale[c], sh = shift[c];
    long r1 = r0 + BNB_ROWS;
    if (r1 > M) r1 = M;
    for (long r = r0 + ty; r < r1; r += 4) {
      const float y = Y[r * ldy + c];
      float g = dZ[r * lddz + c];
      if (relu && !(fmaf(y, sc, sh) > 0.f)) g = 0.f;
      s1 += g;
      s2 += g * ((y - mu) * is);
    }
  }
  red[ty][cl] = make_float2(s1, s2);
  __syncthreads();
  if (ty == 0 && c < C) {
    float2 a = red[0][cl];
    for (int t = 1; t < 4; ++t) { a.x += red[t][cl].x; a.y += red[t][cl].y; }
    part[(long)blockIdx.y * C + c] = a;
  }
}

// Same partial sums for the common case (C % 4 == 0, 16-byte aligned rows): float4 columns per lane (a wave covers
// 4 rows = 1 KiB per load), 4 independent row loads in flight per thread, 512-row blocks (half the partials for
// the finalize kernel).  The one-float-per-lane kernel above ran at 3 TB/s (36 us for 2 x 54 MB at E = 212k).
#define BNB_ROWS_V4 512
template <class T>
__global__ void __launch_bounds__(256) k_bn_bwd_partial_v4(const T* __restrict__ dZ, long lddz,
                                                           const T* __restrict__ Y, long ldy, long M, int C,
                                                           const float* __restrict__ mean,
                                                           const float* __restrict__ invstd,
                                                           const float* __restrict__ scale,
                                                           const float* __restrict__ shift, int relu, float2* part) {
  __shared__ float4 red1[16][16], red2[16][16];
  const int q = threadIdx.x & 15, rg = threadIdx.x >> 4;     // 4 columns 4q.. of the 64-column slab, row group
  const int c = blockIdx.x * 64 + 4 * q;
  const long r0 = (long)blockIdx.y * BNB_ROWS_V4;
  long r1 = r0 + BNB_ROWS_V4;
  if (r1 > M) r1 = M;
  float4 s1 = make_float4(0.f, 0.f, 0.f, 0.f), s2 = s1;
  if (c < C) {
    const float4 mu = *reinterpret_cast<const float4*>(mean + c), is = *reinterpret_cast<const float4*>(invstd + c);
    const float4 sc = *reinterpret_cast<const float4*>(scale + c), sh = *reinterpret_cast<const float4*>(shift + c);
    auto acc1 = [&](float y, float g, float m, float i, float a, float b, float& t1, float& t2) {
      if (relu && !(fmaf(y, a, b) > 0.f)) g = 0.f;
      t1 += g;
      t2 += g * ((y - m) * i);
    };
    for (long r = r0 + rg; r < r1; r += 64) {                 // rows r, r+16, r+32, r+48 in flight together
      float4 y[4], g[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const long rr = (r + 16 * k < r1) ? r + 16 * k : r1 - 1;
        y[k] = yl_ld4(Y + rr * ldy + c);
        g[k] = yl_ld4(dZ + rr * lddz + c);
      }
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        if (r + 16 * k < r1) {
          acc1(y[k].x, g[k].x, mu.x, is.x, sc.x, sh.x, s1.x, s2.x);
          acc1(y[k].y, g[k].y, mu.y, is.y, sc.y, sh.y, s1.y, s2.y);
          acc1(y[k].z, g[k].z, mu.z, is.z, sc.z, sh.z, s1.z, s2.z);
          acc1(y[k].w, g[k].w, mu.w, is.w, sc.w, sh.w, s1.w, s2.w);
        }
      }
    }
  }
  red1[rg][q] = s1; red2[rg][q] = s2;
  __syncthreads();
  if (rg == 0 && c < C) {
    float4 a = red1[0][q], b = red2[0][q];
    for (int t = 1; t < 16; ++t) {
      a.x += red1[t][q].x; a.y += red1[t][q].y; a.z += red1[t][q].z; a.w += red1[t][q].w;
      b.x += red2[t][q].x; b.y += red2[t][q].y; b.z += red2[t][q].z; b.w += red2[t][q].w;
    }
    float2* o = part + (long)blockIdx.y * C + c;
    o[0] = make_float2(a.x, b.x); o[1] = make_float2(a.y, b.y); o[2] = make_float2(a.z, b.z); o[3] = make_float2(a.w, b.w);
  }
}

__global__ void __launch_bounds__(1024) k_bn_bwd_finalize(const float2* part, long nb, long M,
                                                          int C, float* dgamma, float* dbeta,
                                                          int accumulate, float* coef) {
  __shared__ double s1s[16][64], s2s[16][64];
  const int cl = threadIdx.x & 63, p = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + cl;
  const long per = (nb + 15) / 16;
  long b0 = p * per, b1 = b0 + per;
  if (b1 > nb) b1 = nb;
  double a = 0.0, b = 0.0;
  if (c < C)
    for (long i = b0; i < b1; i += 8) {                    // 8 independent loads per step, summed in order
      float2 t[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) t[k] = part[(i + k < b1 ? i + k : b1 - 1) * C + c];
#pragma unroll
      for (int k = 0; k < 8; ++k)
        if (i + k < b1) { a += (double)t[k].x; b += (double)t[k].y; }
    }
  s1s[p][cl] = a; s2s[p][cl] = b;
  __syncthreads();
  if (p == 0 && c < C) {
    for (int t = 1; t < 16; ++t) { a += s1s[t][cl]; b += s2s[t][cl]; }
    float dg = (float)b, dbt = (float)a;
    if (accumulate) { dg += dgamma[c]; dbt += dbeta[c]; }
    dgamma[c] = dg; dbeta[c] = dbt;
    coef[c] = (float)(a / (double)M);
    coef[C + c] = (float)(b / (double)M);
  }
}

__global__ void k_bn_bwd_apply(const float* dZ, long lddz, const float* Y, long ldy, long M, int C,
                               const float* mean, const float* invstd, const float* scale,
                               const float* shift, int relu, const float* coef, float* dY,
                               long lddy) {
  const int c = blockIdx.x * 64 + (threadIdx.x & 63);
  if (c >= C) return;
  const float mu = mean[c], is = invstd[c], sc = scale[c], sh = shift[c];
  const float c1 = coef[c], c2 = coef[C + c];
  for (long r = (long)blockIdx.y * 4 + (threadIdx.x >> 6); r < M; r += (long)gridDim.y * 4) {
    const float y = Y[r * ldy + c];
    float g = dZ[r * lddz + c];
    if (relu && !(fmaf(y, sc, sh) > 0.f)) g = 0.f;
    const float xh = (y - mu) * is;
    dY[r * lddy + c] = sc * (g - c1 - xh * c2);
  }
}

// dY = scale*(dyh - c1 - xhat*c2) with 4 columns per lane and 4 rows in flight per thread (bf16-stored tensors:
// 8-byte accesses; also used for fp32 when the rows are 16-byte aligned)
template <class T>
__global__ void __launch_bounds__(256) k_bn_bwd_apply_v4(const T* __restrict__ dZ, long lddz, const T* __restrict__ Y,
                                                         long ldy, long M, int C, const float* __restrict__ mean,
                                                         const float* __restrict__ invstd, const float* __restrict__ scale,
                                                         const float* __restrict__ shift, int relu,
                                                         const float* __restrict__ coef, T* dY, long lddy) {
  const int q = threadIdx.x & 15, rg = threadIdx.x >> 4;
  const int c = blockIdx.x * 64 + 4 * q;
  if (c >= C) return;
  const float4 mu = *reinterpret_cast<const float4*>(mean + c), is = *reinterpret_cast<const float4*>(invstd + c);
  const float4 sc = *reinterpret_cast<const float4*>(scale + c), sh = *reinterpret_cast<const float4*>(shift + c);
  const float4 c1 = *reinterpret_cast<const float4*>(coef + c), c2 = *reinterpret_cast<const float4*>(coef + C + c);
  auto one = [&](float y, float g, float m, float i, float a, float b, float k1, float k2) {
    if (relu && !(fmaf(y, a, b) > 0.f)) g = 0.f;
    return a * (g - k1 - ((y - m) * i) * k2);
  };
  for (long r = (long)blockIdx.y * 64 + rg; r < M; r += (long)gridDim.y * 64) {
    float4 y[4], g[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const long rr = (r + 16 * k < M) ? r + 16 * k : M - 1;
      y[k] = yl_ld4(Y + rr * ldy + c);
      g[k] = yl_ld4(dZ + rr * lddz + c);
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      if (r + 16 * k < M)
        yl_st4(dY + (r + 16 * k) * lddy + c,
               make_float4(one(y[k].x, g[k].x, mu.x, is.x, sc.x, sh.x, c1.x, c2.x),
                           one(y[k].y, g[k].y, mu.y, is.y, sc.y, sh.y, c1.y, c2.y),
                           one(y[k].z, g[k].z, mu.z, is.z, sc.z, sh.z, c1.z, c2.z),
                           one(y[k].w, g[k].w, mu.w, is.w, sc.w, sh.w, c1.w, c2.w)));
    }
  }
}

extern "C" size_t yolat_bn_bwd_work_elems(int64_t M, int64_t C) {
  return (size_t)(2 * yl_cdiv(M, BNB_ROWS) * C + 2 * C);
}

extern "C" int yolat_bn_relu_bwd(const float* dZ, int64_t lddz, const float* Y, int64_t ldy,
                                 int64_t M, int64_t C, const float* gamma, const float* save_mean,
                                 const float* save_invstd, const float* scale, const float* shift,
                                 int relu, float* dgamma, float* dbeta, int accumulate, float* dY,
                                 int64_t lddy, float* work, yolat_stream_t stream) {
  (void)gamma;
  if (M <= 0 || C <= 0 || !dZ || !Y || !save_mean || !save_invstd || !scale || !shift || !dgamma ||
      !dbeta || !dY || !work)
    return YOLAT_E_INVALID;
  hipStream_t st = (hipStream_t)stream;
  const bool v4 = (C % 4 == 0) && (lddz % 4 == 0) && (ldy % 4 == 0) && yl_aligned16(dZ) && yl_aligned16(Y) &&
                  yl_aligned16(save_mean) && yl_aligned16(save_invstd) && yl_aligned16(scale) && yl_aligned16(shift);
  const long nb = v4 ? yl_cdiv(M, BNB_ROWS_V4) : yl_cdiv(M, BNB_ROWS);
  float2* part = reinterpret_cast<float2*>(work);
  float* coef = work + 2 * (long)yl_cdiv(M, BNB_ROWS) * C;          // after the (larger) scalar-layout partial area
  if (v4)
    hipLaunchKernelGGL(k_bn_bwd_partial_v4<float>, dim3(yl_cdiv(C, 64), (unsigned)nb), dim3(256), 0, st, dZ, (long)lddz, Y,
                       (long)ldy, (long)M, (int)C, save_mean, save_invstd, scale, shift, relu, part);
  else
  hipLaunchKernelGGL(k_bn_bwd_partial, dim3(yl_cdiv(C, 64), (unsigned)nb), dim3(256), 0, st, dZ,
                     (long)lddz, Y, (long)ldy, (long)M, (int)C, save_mean, save_invstd, scale,
                     shift, relu, part);
  YL_LAUNCH_CHECK();
  hipLaunchKernelGGL(k_bn_bwd_finalize, dim3(yl_cdiv(C, 64)), dim3(1024), 0, st, part, nb, (long)M,
                     (int)C, dgamma, dbeta, accumulate, coef);
  YL_LAUNCH_CHECK();
  if (v4 && lddy % 4 == 0 && yl_aligned16(dY) && yl_aligned16(coef)) {
    // float4 columns, 4 rows in flight per thread (the one-float-per-lane kernel below ran at ~3 TB/s)
    int gy = yl_cdiv(M, 64);
    if (gy > 4096) gy = 4096;
    hipLaunchKernelGGL(k_bn_bwd_apply_v4<float>, dim3(yl_cdiv(C, 64), gy), dim3(256), 0, st, dZ, (long)lddz, Y, (long)ldy,
                       (long)M, (int)C, save_mean, save_invstd, scale, shift, relu, coef, dY, (long)lddy);
    YL_LAUNCH_CHECK();
    return 0;
  }
  int gy = yl_cdiv(M, 4);
  if (gy > 2048) gy = 2048;
  hipLaunchKernelGGL(k_bn_bwd_apply, dim3(yl_cdiv(C, 64), gy), dim3(256), 0, st, dZ, (long)lddz, Y,
                     (long)ldy, (long)M, (int)C, save_mean, save_invstd, scale, shift, relu, coef,
                     dY, (long)lddy);
  YL_LAUNCH_CHECK();
  return 0;
}

// The apply pass alone: dY = scale * (g - c1 - xhat * c2) with coef = (c1 | c2) [2C] computed elsewhere
// (yolat_bn_csr_l2_bwd's next_coef).  half != 0: dZ / Y / dY bfloat16-stored.  C % 4 == 0, 16-byte (8-byte) aligned rows.
extern "C" int yolat_bn_relu_bwd_apply(const void* dZ, int64_t lddz, const void* Y, int64_t ldy, int64_t M, int64_t C,
                                       const float* save_mean, const float* save_invstd, const float* scale,
                                       const float* shift, int relu, const float* coef, void* dY, int64_t lddy, int half,
                                       yolat_stream_t stream) {
  if (M <= 0 || C <= 0 || !dZ || !Y || !save_mean || !save_invstd || !scale || !shift || !coef || !dY) return YOLAT_E_INVALID;
  const uintptr_t al = half ? 7 : 15;
  if (C % 4 != 0 || lddz % 4 != 0 || ldy % 4 != 0 || lddy % 4 != 0 || (((uintptr_t)dZ | (uintptr_t)Y | (uintptr_t)dY) & al) ||
      !yl_aligned16(save_mean) || !yl_aligned16(save_invstd) || !yl_aligned16(scale) || !yl_aligned16(shift) ||
      !yl_aligned16(coef))
    return YOLAT_E_UNSUPPORTED;
  int gy = yl_cdiv(M, 64);
  if (gy > 4096) gy = 4096;
  hipStream_t st = (hipStream_t)stream;
  if (half)
    hipLaunchKernelGGL(k_bn_bwd_apply_v4<yl_bf16_t>, dim3(yl_cdiv(C, 64), gy), dim3(256), 0, st,
                       reinterpret_cast<const yl_bf16_t*>(dZ), (long)lddz, reinterpret_cast<const yl_bf16_t*>(Y), (long)ldy,
                       (long)M, (int)C, save_mean, save_invstd, scale, shift, relu, coef, reinterpret_cast<yl_bf16_t*>(dY),
                       (long)lddy);
  else
    hipLaunchKernelGGL(k_bn_bwd_apply_v4<float>, dim3(yl_cdiv(C, 64), gy), dim3(256), 0, st,
                       reinterpret_cast<const float*>(dZ), (long)lddz, reinterpret_cast<const float*>(Y), (long)ldy, (long)M,
                       (int)C, save_mean, save_invstd, scale, shift, relu, coef, reinterpret_cast<float*>(dY), (long)lddy);
  YL_LAUNCH_CHECK();
  return 0;
}

// The same backward on bfloat16-stored dZ / Y / dY ([M,C], C % 4 == 0, 8-byte aligned rows); sums in fp32 / fp64.
extern "C" int yolat_bn_relu_bwd_h(const uint16_t* dZ, int64_t lddz, const uint16_t* Y, int64_t ldy, int64_t M,
                                   int64_t C, const float* save_mean, const float* save_invstd, const float* scale,
                                   const float* shift, int relu, float* dgamma, float* dbeta, int accumulate,
                                   uint16_t* dY, int64_t lddy, float* work, yolat_stream_t stream) {
  if (M <= 0 || C <= 0 || !dZ || !Y || !save_mean || !save_invstd || !scale || !shift || !dgamma || !dbeta || !dY || !work)
    return YOLAT_E_INVALID;
  if (C % 4 != 0 || lddz % 4 != 0 || ldy % 4 != 0 || lddy % 4 != 0 || (((uintptr_t)dZ | (uintptr_t)Y | (uintptr_t)dY) & 7) ||
      !yl_aligned16(save_mean) || !yl_aligned16(save_invstd) || !yl_aligned16(scale) || !yl_aligned16(shift))
    return YOLAT_E_UNSUPPORTED;
  hipStream_t st = (hipStream_t)stream;
  const long nb = yl_cdiv(M, BNB_ROWS_V4);
  float2* part = reinterpret_cast<float2*>(work);
  float* coef = work + 2 * (long)yl_cdiv(M, BNB_ROWS) * C;
  if ((((uintptr_t)coef) & 15) != 0) return YOLAT_E_UNSUPPORTED;
  hipLaunchKernelGGL(k_bn_bwd_partial_v4<yl_bf16_t>, dim3(yl_cdiv(C, 64), (unsigned)nb), dim3(256), 0, st, dZ, (long)lddz,
                     Y, (long)ldy, (long)M, (int)C, save_mean, save_invstd, scale, shift, relu, part);
  YL_LAUNCH_CHECK();
  hipLaunchKernelGGL(k_bn_bwd_finalize, dim3(yl_cdiv(C, 64)), dim3(1024), 0, st, part, nb, (long)M, (int)C, dgamma, dbeta,
                     accumulate, coef);
  YL_LAUNCH_CHECK();
  int gy = yl_cdiv(M, 64);
  if (gy > 4096) gy = 4096;
  hipLaunchKernelGGL(k_bn_bwd_apply_v4<yl_bf16_t>, dim3(yl_cdiv(C, 64), gy), dim3(256), 0, st, dZ, (long)lddz, Y, (long)ldy,
                     (long)M, (int)C, save_mean, save_invstd, scale, shift, relu, coef, dY, (long)lddy);
  YL_LAUNCH_CHECK();
  return 0;
}
