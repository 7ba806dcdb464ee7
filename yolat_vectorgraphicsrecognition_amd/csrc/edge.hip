// edge.hip — AttrRelativeEdgeConvGlobalPool2 (gcn_lib/sparse/torch_vertex.py:288-341) on gfx950:
// gathered edge-feature GEMM, CSR mean aggregation and their backward.  All [E,*] tensors are in
// destination-sorted (CSR) order, so aggregation reads contiguous rows and needs no atomics.
#include "common.hpp"
#include <stdlib.h>

extern "C" int yolat_edge_lin1_fwd(const float* x, int64_t ldx, int64_t N, int64_t Cin,
                                   const int32_t* src_csr, const int32_t* dst_csr,
                                   const float* attr_csr, int64_t E, const float* W1, int64_t ldw,
                                   const float* b1, int64_t C, const float* o_scale,
                                   const float* o_shift, int o_relu, float* H1, int64_t ldh,
                                   float* stats, yolat_stream_t stream) {
  if (E < 0 || N <= 0 || Cin <= 0 || C <= 0 || !x || !W1) return YOLAT_E_INVALID;
  if (E == 0) return 0;
  if (!src_csr || !dst_csr || !attr_csr || !H1 || E >= (1LL << 31)) return YOLAT_E_INVALID;
  const long K = 2 * Cin + 4;
  if (ldw < K || ldh < C || ldx < Cin) return YOLAT_E_INVALID;
  if ((o_scale == nullptr) != (o_shift == nullptr)) return YOLAT_E_INVALID;
  EdgeOp a = yl_edge(x, ldx, Cin, src_csr, dst_csr, attr_csr, E);
  DenseOp b = yl_dense(W1, ldw, C, K);
  Epilogue ep;
  ep.bias = b1; ep.scale = o_scale; ep.shift = o_shift; ep.relu = o_relu;
  ep.Y = H1; ep.ldy = ldh; ep.accumulate = 0; ep.stats = stats; ep.seg = nullptr; ep.pool = nullptr; ep.ldpool = 0;
  hipStream_t st = (hipStream_t)stream;
  dim3 grid(yl_cdiv(E, 64), yl_cdiv(C, 64));
  if (K <= 16)
    hipLaunchKernelGGL((k_gemm_nt<64, 64, 16, EdgeOp, DenseOp, false>), grid, dim3(256), 0, st, a,
                       b, ep, (int)E, (int)C, (int)K);
  else
    hipLaunchKernelGGL((k_gemm_nt<64, 64, 32, EdgeOp, DenseOp, false>), grid, dim3(256), 0, st, a,
                       b, ep, (int)E, (int)C, (int)K);
  YL_LAUNCH_CHECK();
  return 0;
}

// ------------------------------------------------------------------------------------------------
// Eval-mode edge MLP, both layers in one kernel (torch_vertex.py:311,331-335 `self.nn`, BN folded):
//   H2[q] = relu(s2*(W2 . relu(s1*(W1 . [x[dst] | x[src]-x[dst] | attr](q) + b1) + t1) + b2) + t2)
// One 64-edge tile per workgroup.  GEMM1 is the k_gemm_nt<64,64,*> loop on the gathered operand; its
// activated accumulators go to LDS (never to HBM: saves the E x 64 write + read and one launch), GEMM2
// reads them back as MFMA A-fragments with W2 staged in LDS during GEMM1.  Same k order and the same
// epilogue arithmetic as the two-kernel path -> bit-identical H2.
//   BLOCK = true : Cin % 32 == 0, vector loads; the 4 attr columns are a final 2-MFMA step instead of a
//                  mostly-zero 32-wide k-step (K = 132 costs 66 MFMAs per wave, not 80)
//   BLOCK = false: any Cin (the Cin = 5 head layer), 16-wide generic k-steps
// ------------------------------------------------------------------------------------------------
template <bool BLOCK>
__global__ void __launch_bounds__(256) k_edge_mlp2(EdgeOp A, DenseOp W1, const float* __restrict__ b1,
                                                   const float* __restrict__ s1, const float* __restrict__ t1,
                                                   DenseOp W2, Epilogue ep2, int E) {
  constexpr int BK = BLOCK ? 32 : 16, LD = BK + 1, KQ = BK / 4, NL = (64 * KQ) / 256, LDH = 65;
  __shared__ float As[64 * LD];
  __shared__ float Bs[64 * LD];
  __shared__ float Hs[64 * LDH];
  __shared__ float W2s[64 * LDH];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int wm = wave >> 1, wn = wave & 1, l31 = lane & 31, lhi = lane >> 5;
  const int row0 = blockIdx.x * 64;
  const int col = wn * 32 + l31;
  const int K1 = A.cols, KX = BLOCK ? 2 * A.Cin : K1;     // KX: extent covered by the BK-wide steps

  // epilogue constants + W2 (64x64) prefetched now, consumed after GEMM1
  const float bias1 = b1[col], sc1 = s1 ? s1[col] : 1.f, sh1 = s1 ? t1[col] : 0.f;
  const EpiPre pre2 = epi_prefetch(ep2, row0 + wm * 32, col, E, 64);
  float rw2[4][4];
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    const int i = tid + t * 256;
    W2.template load4<false>(i >> 4, 4 * (i & 15), rw2[t]);
  }

  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  float ra[NL][4], rb[NL][4];
  auto fetch = [&](int k0) {
#pragma unroll
    for (int t = 0; t < NL; ++t) {
      const int i = tid + t * 256;
      if (BLOCK) {
        A.template load4<true>(row0 + i / KQ, k0 + 4 * (i % KQ), ra[t]);
        W1.template load4<true>(i / KQ, k0 + 4 * (i % KQ), rb[t]);
      } else {
        A.template load4<false>(row0 + i / KQ, k0 + 4 * (i % KQ), ra[t]);
        W1.template load4<false>(i / KQ, k0 + 4 * (i % KQ), rb[t]);
      }
    }
  };
  auto stage = [&]() {
#pragma unroll
    for (int t = 0; t < NL; ++t) {
      const int i = tid + t * 256;
      float* d = As + (i / KQ) * LD + 4 * (i % KQ);
      d[0] = ra[t][0]; d[1] = ra[t][1]; d[2] = ra[t][2]; d[3] = ra[t][3];
      float* e = Bs + (i / KQ) * LD + 4 * (i % KQ);
      e[0] = rb[t][0]; e[1] = rb[t][1]; e[2] = rb[t][2]; e[3] = rb[t][3];
    }
  };
  // the attr columns (BLOCK): 64 rows x one float4 of A (threads 0..63) and of W1 (threads 64..127)
  float rattr[4] = {0.f, 0.f, 0.f, 0.f};
  if (BLOCK) {
    if (tid < 64) A.template load4<true>(row0 + tid, KX, rattr);
    else if (tid < 128) W1.template load4<true>(tid - 64, KX, rattr);
  }
  fetch(0);
#pragma unroll
  for (int t = 0; t < 4; ++t) {       // W2 -> LDS (first barrier below publishes it)
    const int i = tid + t * 256;
    float* d = W2s + (i >> 4) * LDH + 4 * (i & 15);
    d[0] = rw2[t][0]; d[1] = rw2[t][1]; d[2] = rw2[t][2]; d[3] = rw2[t][3];
  }
  for (int k0 = 0; k0 < KX; k0 += BK) {
    stage();
    __syncthreads();
    if (k0 + BK < KX) fetch(k0 + BK);
#pragma unroll
    for (int kk = 0; kk < BK; kk += 2) {
      const float a = As[(wm * 32 + l31) * LD + kk + lhi];
      const float b = Bs[(wn * 32 + l31) * LD + kk + lhi];
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
    }
    __syncthreads();
  }
  if (BLOCK) {
    if (tid < 128) {
      float* d = (tid < 64 ? As + tid * LD : Bs + (tid - 64) * LD);
      d[0] = rattr[0]; d[1] = rattr[1]; d[2] = rattr[2]; d[3] = rattr[3];
    }
    __syncthreads();
#pragma unroll
    for (int kk = 0; kk < 4; kk += 2) {
      const float a = As[(wm * 32 + l31) * LD + kk + lhi];
      const float b = Bs[(wn * 32 + l31) * LD + kk + lhi];
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
    }
  }
  // layer-1 epilogue -> LDS (same arithmetic as wave_epilogue: +bias, fma(scale, shift), relu)
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int row = wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhi;
    Hs[row * LDH + col] = fmaxf(fmaf(acc[r] + bias1, sc1, sh1), 0.f);
  }
  __syncthreads();
  f32x16 acc2;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc2[r] = 0.f;
#pragma unroll 8
  for (int kk = 0; kk < 64; kk += 2) {
    const float a = Hs[(wm * 32 + l31) * LDH + kk + lhi];
    const float b = W2s[(wn * 32 + l31) * LDH + kk + lhi];
    acc2 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc2, 0, 0, 0);
  }
  wave_epilogue(acc2, row0 + wm * 32, col, lhi, ep2, E, 64, pre2);
}

extern "C" int yolat_edge_mlp2_eval(const float* x, int64_t ldx, int64_t N, int64_t Cin,
                                    const int32_t* src_csr, const int32_t* dst_csr, const float* attr_csr,
                                    int64_t E, const float* W1, const float* b1, const float* s1,
                                    const float* t1, const float* W2, const float* b2, const float* s2,
                                    const float* t2, int64_t C, float* H2, int64_t ldh,
                                    yolat_stream_t stream) {
  if (E < 0 || N <= 0 || Cin <= 0 || !x || !W1 || !W2 || !b1) return YOLAT_E_INVALID;
  if (C != 64) return YOLAT_E_UNSUPPORTED;
  if (E == 0) return 0;
  if (!src_csr || !dst_csr || !attr_csr || !H2 || E >= (1LL << 31) || ldh < C || ldx < Cin)
    return YOLAT_E_INVALID;
  if ((s1 == nullptr) != (t1 == nullptr) || (s2 == nullptr) != (t2 == nullptr)) return YOLAT_E_INVALID;
  const long K1 = 2 * Cin + 4;
  EdgeOp a = yl_edge(x, ldx, Cin, src_csr, dst_csr, attr_csr, E);
  DenseOp w1 = yl_dense(W1, K1, C, K1), w2 = yl_dense(W2, C, C, C);
  Epilogue ep;
  ep.bias = b2; ep.scale = s2; ep.shift = t2; ep.relu = 1;
  ep.Y = H2; ep.ldy = ldh; ep.accumulate = 0; ep.stats = nullptr; ep.seg = nullptr; ep.pool = nullptr; ep.ldpool = 0;
  hipStream_t st = (hipStream_t)stream;
  const dim3 grid(yl_cdiv(E, 64));
  if (Cin % 32 == 0 && a.vec && w1.vec)
    hipLaunchKernelGGL(k_edge_mlp2<true>, grid, dim3(256), 0, st, a, w1, b1, s1, t1, w2, ep, (int)E);
  else
    hipLaunchKernelGGL(k_edge_mlp2<false>, grid, dim3(256), 0, st, a, w1, b1, s1, t1, w2, ep, (int)E);
  YL_LAUNCH_CHECK();
  return 0;
}

// ------------------------------------------------------------------------------------------------
// Factorised eval-mode edge MLP: layer 1 is a gather-add of per-node products (see yolat_hip.h):
//   h1[q] = relu(s1*(U[dst_q] + V[src_q] + Wc4.attr_q + b1) + t1)   (VALU, 64 floats per edge)
//   H2[q] = relu(s2*(W2.h1[q] + b2) + t2)                            (MFMA, 32 per wave per 64-edge tile)
// The K = 2*Cin part of the per-edge GEMM (64 of the 98 MFMAs of k_edge_mlp2) is gone: it was computed
// once per node by k_gemm_nt_node3.  Thread (row r, float4 column q): the 16 threads of a row share its
// two index loads; all 8 index loads and then all 12 row gathers of a thread are issued together.
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_edge_uv_mlp2(const float* __restrict__ UV, long ld_uv,
                                                      const int* __restrict__ src, const int* __restrict__ dst,
                                                      const float* __restrict__ attr,
                                                      const float* __restrict__ Wc4, const float* __restrict__ b1,
                                                      const float* __restrict__ s1, const float* __restrict__ t1,
                                                      DenseOp W2, Epilogue ep2, int E) {
  constexpr int LDH = 65;
  __shared__ float Hs[64 * LDH];
  __shared__ float W2s[64 * LDH];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int wm = wave >> 1, wn = wave & 1, l31 = lane & 31, lhi = lane >> 5;
  const int row0 = blockIdx.x * 64;
  const int q = tid & 15, rb = tid >> 4;             // this thread: columns 4q..4q+3 of rows rb, rb+16, rb+32, rb+48
  const EpiPre pre2 = epi_prefetch(ep2, row0 + wm * 32, wn * 32 + l31, E, 64);
  // W2 -> registers -> LDS, per-column constants of layer 1
  float rw2[4][4];
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    const int i = tid + t * 256;
    W2.template load4<false>(i >> 4, 4 * (i & 15), rw2[t]);
  }
  float4 wc[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) wc[j] = *reinterpret_cast<const float4*>(Wc4 + (4 * q + j) * 4);
  const float4 bb = *reinterpret_cast<const float4*>(b1 + 4 * q);
  float4 sc = make_float4(1.f, 1.f, 1.f, 1.f), sh = make_float4(0.f, 0.f, 0.f, 0.f);
  if (s1) { sc = *reinterpret_cast<const float4*>(s1 + 4 * q); sh = *reinterpret_cast<const float4*>(t1 + 4 * q); }
  int di[4], si[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    const int e = yl_min(row0 + rb + 16 * t, E - 1);
    di[t] = dst[e]; si[t] = src[e];
  }
  float4 u[4], v[4], a[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    const int e = yl_min(row0 + rb + 16 * t, E - 1);
    u[t] = *reinterpret_cast<const float4*>(UV + (long)di[t] * ld_uv + 4 * q);
    v[t] = *reinterpret_cast<const float4*>(UV + (long)si[t] * ld_uv + 64 + 4 * q);
    a[t] = *reinterpret_cast<const float4*>(attr + (long)e * 4);
  }
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    const int i = tid + t * 256;
    float* d = W2s + (i >> 4) * LDH + 4 * (i & 15);
    d[0] = rw2[t][0]; d[1] = rw2[t][1]; d[2] = rw2[t][2]; d[3] = rw2[t][3];
  }
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    auto one = [&](float uu, float vv, const float4& w, float b, float s, float h) {
      float z = uu + vv;
      z = fmaf(a[t].x, w.x, z); z = fmaf(a[t].y, w.y, z); z = fmaf(a[t].z, w.z, z); z = fmaf(a[t].w, w.w, z);
      return fmaxf(fmaf(z + b, s, h), 0.f);
    };
    float* hrow = Hs + (rb + 16 * t) * LDH + 4 * q;
    hrow[0] = one(u[t].x, v[t].x, wc[0], bb.x, sc.x, sh.x);
    hrow[1] = one(u[t].y, v[t].y, wc[1], bb.y, sc.y, sh.y);
    hrow[2] = one(u[t].z, v[t].z, wc[2], bb.z, sc.z, sh.z);
    hrow[3] = one(u[t].w, v[t].w, wc[3], bb.w, sc.w, sh.w);
  }
  __syncthreads();
  f32x16 acc2;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc2[r] = 0.f;
#pragma unroll 8
  for (int kk = 0; kk < 64; kk += 2) {
    const float av = Hs[(wm * 32 + l31) * LDH + kk + lhi];
    const float bv = W2s[(wn * 32 + l31) * LDH + kk + lhi];
    acc2 = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc2, 0, 0, 0);
  }
  wave_epilogue(acc2, row0 + wm * 32, wn * 32 + l31, lhi, ep2, E, 64, pre2);
}

// ------------------------------------------------------------------------------------------------
// Factorised edge MLP + mean aggregation in one kernel (eval): the [E,64] message matrix never reaches HBM.
// One workgroup = `npt` consecutive destination nodes (<= 16) = the contiguous CSR edge range
// [row_ptr[n0], row_ptr[n0+npt]) processed in passes of 64 edges (npt is chosen by the host so that one pass
// is the common case).  Per pass: gather-add layer 1 -> LDS -> MFMA layer 2 -> BN+ReLU -> LDS; then thread
// (node j, float4 column q) adds the pass's rows of node j in ascending edge order to its running sum.  At
// the end  f_out[n] += sum / deg  on top of the root Linear the node-side launch wrote.  No atomics; the
// per-node summation order is the CSR order, bit-identical to k_edge_uv_mlp2 + k_csr_mean_fwd.
// ------------------------------------------------------------------------------------------------
// NG = 16-node groups per tile: 1 (<= 16 nodes, the dense-graph case) or 4 (<= 64 nodes: graphs with ~1 edge per
// node — the Floorplans shape — would otherwise fill a 64-edge pass to a third).
template <int NG>
__global__ void __launch_bounds__(256) k_edge_uv_mlp2_mean(const float* __restrict__ UV, long ld_uv,
                                                           const int* __restrict__ src,
                                                           const int* __restrict__ dst,
                                                           const float* __restrict__ attr,
                                                           const int* __restrict__ row_ptr, int N, int npt,
                                                           const float* __restrict__ Wc4,
                                                           const float* __restrict__ b1,
                                                           const float* __restrict__ s1,
                                                           const float* __restrict__ t1, DenseOp W2,
                                                           const float* __restrict__ b2,
                                                           const float* __restrict__ s2,
                                                           const float* __restrict__ t2, float* f_out, long ld_fo,
                                                           int E) {
  constexpr int LDH = 65;
  __shared__ float Hs[64 * LDH];      // layer-1 activations of the pass, then the layer-2 messages
  __shared__ float W2s[64 * LDH];
  __shared__ int rp[16 * NG + 1];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int wm = wave >> 1, wn = wave & 1, l31 = lane & 31, lhi = lane >> 5;
  const int n0 = blockIdx.x * npt;
  const int nn = yl_min(npt, N - n0);                 // nodes of this tile
  if (tid <= 16 * NG) rp[tid] = row_ptr[yl_min(n0 + tid, n0 + nn)];
  const int q = tid & 15, rb = tid >> 4;              // gather role: columns 4q..4q+3 of rows rb + 16t
  const int col = wn * 32 + l31;                      // MFMA role: output column of this lane
  float rw2[4][4];
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    const int i = tid + t * 256;
    W2.template load4<false>(i >> 4, 4 * (i & 15), rw2[t]);
  }
  float4 wc[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) wc[j] = *reinterpret_cast<const float4*>(Wc4 + (4 * q + j) * 4);
  const float4 bb = *reinterpret_cast<const float4*>(b1 + 4 * q);
  float4 sc = make_float4(1.f, 1.f, 1.f, 1.f), sh = make_float4(0.f, 0.f, 0.f, 0.f);
  if (s1) { sc = *reinterpret_cast<const float4*>(s1 + 4 * q); sh = *reinterpret_cast<const float4*>(t1 + 4 * q); }
  const float bias2 = b2 ? b2[col] : 0.f, sc2 = s2 ? s2[col] : 1.f, sh2 = s2 ? t2[col] : 0.f;
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    const int i = tid + t * 256;
    float* d = W2s + (i >> 4) * LDH + 4 * (i & 15);
    d[0] = rw2[t][0]; d[1] = rw2[t][1]; d[2] = rw2[t][2]; d[3] = rw2[t][3];
  }
  __syncthreads();
  const int e0 = rp[0], e1 = rp[nn];
  int my_b[NG], my_e[NG];                             // aggregation role: nodes rb + 16 j, columns 4q..
  float4 sum[NG];
#pragma unroll
  for (int j = 0; j < NG; ++j) {
    my_b[j] = rp[yl_min(rb + 16 * j, nn)]; my_e[j] = rp[yl_min(rb + 16 * j + 1, nn)];
    sum[j] = make_float4(0.f, 0.f, 0.f, 0.f);
  }
  for (int c0 = e0; c0 < e1; c0 += 64) {
    int di[4], si[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const int e = yl_min(c0 + rb + 16 * t, E - 1);
      di[t] = dst[e]; si[t] = src[e];
    }
    float4 u[4], v[4], a[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const int e = yl_min(c0 + rb + 16 * t, E - 1);
      u[t] = *reinterpret_cast<const float4*>(UV + (long)di[t] * ld_uv + 4 * q);
      v[t] = *reinterpret_cast<const float4*>(UV + (long)si[t] * ld_uv + 64 + 4 * q);
      a[t] = *reinterpret_cast<const float4*>(attr + (long)e * 4);
    }
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      auto one = [&](float uu, float vv, const float4& w, float b, float s, float h) {
        float z = uu + vv;
        z = fmaf(a[t].x, w.x, z); z = fmaf(a[t].y, w.y, z); z = fmaf(a[t].z, w.z, z); z = fmaf(a[t].w, w.w, z);
        return fmaxf(fmaf(z + b, s, h), 0.f);
      };
      float* hrow = Hs + (rb + 16 * t) * LDH + 4 * q;
      hrow[0] = one(u[t].x, v[t].x, wc[0], bb.x, sc.x, sh.x);
      hrow[1] = one(u[t].y, v[t].y, wc[1], bb.y, sc.y, sh.y);
      hrow[2] = one(u[t].z, v[t].z, wc[2], bb.z, sc.z, sh.z);
      hrow[3] = one(u[t].w, v[t].w, wc[3], bb.w, sc.w, sh.w);
    }
    __syncthreads();
    f32x16 acc2;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc2[r] = 0.f;
#pragma unroll 8
    for (int kk = 0; kk < 64; kk += 2) {
      const float av = Hs[(wm * 32 + l31) * LDH + kk + lhi];
      const float bv = W2s[(wn * 32 + l31) * LDH + kk + lhi];
      acc2 = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc2, 0, 0, 0);
    }
    __syncthreads();                      // every wave is done reading Hs as the layer-1 tile
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhi;
      Hs[row * LDH + col] = fmaxf(fmaf(acc2[r] + bias2, sc2, sh2), 0.f);
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < NG; ++j) {
      if (rb + 16 * j < nn) {             // rows of node rb + 16 j inside this pass, ascending edge order
        const int lo = my_b[j] > c0 ? my_b[j] : c0;
        const int hi = my_e[j] < c0 + 64 ? my_e[j] : c0 + 64;
        for (int e = lo; e < hi; ++e) {
          const float* m = Hs + (e - c0) * LDH + 4 * q;
          sum[j].x += m[0]; sum[j].y += m[1]; sum[j].z += m[2]; sum[j].w += m[3];
        }
      }
    }
    __syncthreads();
  }
#pragma unroll
  for (int j = 0; j < NG; ++j) {
    const int deg = my_e[j] - my_b[j];
    if (rb + 16 * j < nn && deg > 0) {
      const float inv = 1.f / (float)deg;
      float4* o = reinterpret_cast<float4*>(f_out + (long)(n0 + rb + 16 * j) * ld_fo + 4 * q);
      float4 d = *o;
      // explicit mul then add (no fma contraction): the same two roundings as k_csr_mean_fwd*
      d.x = yl_mul_rn(sum[j].x, inv) + d.x; d.y = yl_mul_rn(sum[j].y, inv) + d.y;
      d.z = yl_mul_rn(sum[j].z, inv) + d.z; d.w = yl_mul_rn(sum[j].w, inv) + d.w;
      *o = d;
    }
  }
}

extern "C" int yolat_edge_uv_mlp2_mean_eval(const float* UV, int64_t ld_uv, const int32_t* src_csr,
                                            const int32_t* dst_csr, const float* attr_csr,
                                            const int32_t* row_ptr, int64_t N, int64_t E, const float* Wc4,
                                            const float* b1, const float* s1, const float* t1, const float* W2,
                                            const float* b2, const float* s2, const float* t2, int64_t C,
                                            float* f_out, int64_t ld_fo, yolat_stream_t stream) {
  if (E < 0 || N <= 0 || !UV || !Wc4 || !b1 || !W2 || !row_ptr || !f_out) return YOLAT_E_INVALID;
  if (C != 64) return YOLAT_E_UNSUPPORTED;
  if (E == 0) return 0;
  if (!src_csr || !dst_csr || !attr_csr || E >= (1LL << 31) || ld_fo < C || ld_uv < 2 * C) return YOLAT_E_INVALID;
  if ((s1 == nullptr) != (t1 == nullptr) || (s2 == nullptr) != (t2 == nullptr)) return YOLAT_E_INVALID;
  if (ld_uv % 4 != 0 || ld_fo % 4 != 0 || !yl_aligned16(UV) || !yl_aligned16(attr_csr) || !yl_aligned16(Wc4) ||
      !yl_aligned16(b1) || !yl_aligned16(f_out) || (s1 && (!yl_aligned16(s1) || !yl_aligned16(t1))))
    return YOLAT_E_UNSUPPORTED;
  // nodes per workgroup: ~56 edges on average so that a single 64-edge pass is the common case
  // (measured at cfg 5: 9 nodes / one pass 208 us, 12 nodes / a second mostly-empty pass 242 us, 16 nodes /
  // two full passes 194 us — on big graphs two passes halve the per-workgroup W2 staging)
  long npt = (56 * N) / E;
  const long npt2 = (112 * N) / E < 16 ? (112 * N) / E : 16;
  if (npt2 >= 2 * npt - 2 && N / (npt2 > 0 ? npt2 : 1) >= 8192) npt = npt2;
  if (const char* e = getenv("YOLAT_EDGE_NPT")) npt = atol(e);     // tuning hook
  if (npt < 1) npt = 1;
  if (npt > 64) npt = 64;
  DenseOp w2 = yl_dense(W2, C, C, C);
  if (npt <= 16)
    hipLaunchKernelGGL(k_edge_uv_mlp2_mean<1>, dim3(yl_cdiv(N, npt)), dim3(256), 0, (hipStream_t)stream, UV, (long)ld_uv,
                       src_csr, dst_csr, attr_csr, row_ptr, (int)N, (int)npt, Wc4, b1, s1, t1, w2, b2, s2, t2, f_out,
                       (long)ld_fo, (int)E);
  else
    hipLaunchKernelGGL(k_edge_uv_mlp2_mean<4>, dim3(yl_cdiv(N, npt)), dim3(256), 0, (hipStream_t)stream, UV, (long)ld_uv,
                       src_csr, dst_csr, attr_csr, row_ptr, (int)N, (int)npt, Wc4, b1, s1, t1, w2, b2, s2, t2, f_out,
                       (long)ld_fo, (int)E);
  YL_LAUNCH_CHECK();
  return 0;
}

// ------------------------------------------------------------------------------------------------
// Factorised FIRST edge Linear for the training forward:  H1[q] = U[dst_q] + V[src_q] + W1c.attr_q + b1
// (UV = x.[W1a-W1b | W1b]^T computed once per node by a dense GEMM), plus the BatchNorm partial statistics in the
// GEMM epilogue's format (float2 (sum, M2 about the group mean) per 32-row group and column).  Replaces the
// gathered K = 2 Cin + 4 GEMM when E >> N: at E = 1.2 M / N = 200 k that GEMM ran at 17 TFLOP/s (1.19 ms) because
// its A operand is two random 256-B row gathers per edge.  One workgroup = 64 edges: thread (row rb + 16 t,
// columns 4q..) writes its 16 bytes of H1 and parks them in LDS; 128 threads then reduce the two 32-row groups.
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_edge_uv_lin1(const float* __restrict__ UV, long ld_uv,
                                                      const int* __restrict__ src, const int* __restrict__ dst,
                                                      const float* __restrict__ attr, int E,
                                                      const float* __restrict__ Wc4, const float* __restrict__ b1,
                                                      float* __restrict__ H1, long ldh, float2* __restrict__ stats) {
  constexpr int LDT = 65;
  __shared__ float T[64 * LDT];
  const int tid = threadIdx.x, q = tid & 15, rb = tid >> 4;
  const int row0 = blockIdx.x * 64;
  float4 wc[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) wc[j] = *reinterpret_cast<const float4*>(Wc4 + (4 * q + j) * 4);
  float4 bb = make_float4(0.f, 0.f, 0.f, 0.f);
  if (b1) bb = *reinterpret_cast<const float4*>(b1 + 4 * q);
  int di[4], si[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    const int e = yl_min(row0 + rb + 16 * t, E - 1);
    di[t] = dst[e]; si[t] = src[e];
  }
  float4 u[4], v[4], a[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    const int e = yl_min(row0 + rb + 16 * t, E - 1);
    u[t] = *reinterpret_cast<const float4*>(UV + (long)di[t] * ld_uv + 4 * q);
    v[t] = *reinterpret_cast<const float4*>(UV + (long)si[t] * ld_uv + 64 + 4 * q);
    a[t] = *reinterpret_cast<const float4*>(attr + (long)e * 4);
  }
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    auto one = [&](float uu, float vv, const float4& w, float b) {
      float z = uu + vv;
      z = fmaf(a[t].x, w.x, z); z = fmaf(a[t].y, w.y, z); z = fmaf(a[t].z, w.z, z); z = fmaf(a[t].w, w.w, z);
      return z + b;
    };
    const float4 h = make_float4(one(u[t].x, v[t].x, wc[0], bb.x), one(u[t].y, v[t].y, wc[1], bb.y),
                                 one(u[t].z, v[t].z, wc[2], bb.z), one(u[t].w, v[t].w, wc[3], bb.w));
    const int r = rb + 16 * t;
    if (row0 + r < E) *reinterpret_cast<float4*>(H1 + (long)(row0 + r) * ldh + 4 * q) = h;
    float* tr = T + r * LDT + 4 * q;
    tr[0] = h.x; tr[1] = h.y; tr[2] = h.z; tr[3] = h.w;
  }
  if (stats == nullptr) return;
  __syncthreads();
  if (tid < 128) {
    const int c = tid & 63, grp = tid >> 6;
    const int base = row0 + 32 * grp;
    int cnt = E - base;
    cnt = cnt > 32 ? 32 : cnt;
    if (cnt > 0) {
      float sum = 0.f;
      for (int r = 0; r < cnt; ++r) sum += T[(32 * grp + r) * LDT + c];
      const float mu = sum / (float)cnt;
      float m2 = 0.f;
      for (int r = 0; r < cnt; ++r) { const float d = T[(32 * grp + r) * LDT + c] - mu; m2 += d * d; }
      stats[(long)(base >> 5) * 64 + c] = make_float2(sum, m2);
    }
  }
}

extern "C" int yolat_edge_uv_lin1_fwd(const float* UV, int64_t ld_uv, const int32_t* src_csr, const int32_t* dst_csr,
                                      const float* attr_csr, int64_t E, const float* Wc4, const float* b1, int64_t C,
                                      float* H1, int64_t ldh, float* stats, yolat_stream_t stream) {
  if (E < 0 || !UV || !Wc4) return YOLAT_E_INVALID;
  if (C != 64) return YOLAT_E_UNSUPPORTED;
  if (E == 0) return 0;
  if (!src_csr || !dst_csr || !attr_csr || !H1 || E >= (1LL << 31) || ldh < C || ld_uv < 2 * C) return YOLAT_E_INVALID;
  if (ld_uv % 4 != 0 || ldh % 4 != 0 || !yl_aligned16(UV) || !yl_aligned16(attr_csr) || !yl_aligned16(Wc4) ||
      !yl_aligned16(H1) || (b1 && !yl_aligned16(b1)) || (stats && (((uintptr_t)stats) & 7) != 0))
    return YOLAT_E_UNSUPPORTED;
  hipLaunchKernelGGL(k_edge_uv_lin1, dim3(yl_cdiv(E, 64)), dim3(256), 0, (hipStream_t)stream, UV, (long)ld_uv, src_csr,
                     dst_csr, attr_csr, (int)E, Wc4, b1, H1, (long)ldh, reinterpret_cast<float2*>(stats));
  YL_LAUNCH_CHECK();
  return 0;
}

extern "C" int yolat_edge_uv_mlp2_eval(const float* UV, int64_t ld_uv, const int32_t* src_csr,
                                       const int32_t* dst_csr, const float* attr_csr, int64_t E, const float* Wc4,
                                       const float* b1, const float* s1, const float* t1, const float* W2,
                                       const float* b2, const float* s2, const float* t2, int64_t C, float* H2,
                                       int64_t ldh, yolat_stream_t stream) {
  if (E < 0 || !UV || !Wc4 || !b1 || !W2) return YOLAT_E_INVALID;
  if (C != 64) return YOLAT_E_UNSUPPORTED;
  if (E == 0) return 0;
  if (!src_csr || !dst_csr || !attr_csr || !H2 || E >= (1LL << 31) || ldh < C || ld_uv < 2 * C) return YOLAT_E_INVALID;
  if ((s1 == nullptr) != (t1 == nullptr) || (s2 == nullptr) != (t2 == nullptr)) return YOLAT_E_INVALID;
  if (ld_uv % 4 != 0 || !yl_aligned16(UV) || !yl_aligned16(attr_csr) || !yl_aligned16(Wc4) || !yl_aligned16(b1) ||
      (s1 && (!yl_aligned16(s1) || !yl_aligned16(t1))))
    return YOLAT_E_UNSUPPORTED;
  DenseOp w2 = yl_dense(W2, C, C, C);
  Epilogue ep;
  ep.bias = b2; ep.scale = s2; ep.shift = t2; ep.relu = 1;
  ep.Y = H2; ep.ldy = ldh; ep.accumulate = 0; ep.stats = nullptr; ep.seg = nullptr; ep.pool = nullptr; ep.ldpool = 0;
  hipLaunchKernelGGL(k_edge_uv_mlp2, dim3(yl_cdiv(E, 64)), dim3(256), 0, (hipStream_t)stream, UV, (long)ld_uv, src_csr,
                     dst_csr, attr_csr, Wc4, b1, s1, t1, w2, ep, (int)E);
  YL_LAUNCH_CHECK();
  return 0;
}

extern "C" int yolat_edge_lin1_bwd_w(const float* dH1, int64_t lddh, int64_t E, int64_t C,
                                     const float* x, int64_t ldx, int64_t N, int64_t Cin,
                                     const int32_t* src_csr, const int32_t* dst_csr,
                                     const float* attr_csr, float* dW1, int64_t lddw, float* db1,
                                     int accumulate, float* partial, yolat_stream_t stream) {
  if (E < 0 || N <= 0 || Cin <= 0 || C <= 0 || !x || !dW1 || !partial) return YOLAT_E_INVALID;
  if (E > 0 && (!dH1 || !src_csr || !dst_csr || !attr_csr)) return YOLAT_E_INVALID;
  const long K = 2 * Cin + 4;
  if (lddw < K || lddh < C || E >= (1LL << 31)) return YOLAT_E_INVALID;
  hipStream_t st = (hipStream_t)stream;
  TnPlan p = yl_tn_plan(E, C, K);
  DenseOp y = yl_dense(dH1, lddh, E, C);
  EdgeOp a = yl_edge(x, ldx, Cin, src_csr, dst_csr, attr_csr, E);
  float* dbpart = db1 ? partial + (size_t)p.S * C * K : nullptr;
  dim3 grid(yl_cdiv(C, 64), yl_cdiv(K, 64), p.S);
  hipLaunchKernelGGL((k_gemm_tn<DenseOp, EdgeOp>), grid, dim3(256), 0, st, y, a, partial, dbpart,
                     (int)E, (int)C, (int)K, p.rows_per_split);
  YL_LAUNCH_CHECK();
  const long elems = C * K;
  hipLaunchKernelGGL(k_reduce_splits, dim3(yl_cdiv(elems, 32)), dim3(256), 0, st, partial, elems,
                     p.S, dW1, (long)lddw, (int)K, accumulate);
  YL_LAUNCH_CHECK();
  if (db1) {
    hipLaunchKernelGGL(k_reduce_splits, dim3(yl_cdiv(C, 32)), dim3(256), 0, st, dbpart, (long)C,
                       p.S, db1, (long)C, (int)C, accumulate);
    YL_LAUNCH_CHECK();
  }
  return 0;
}

extern "C" int yolat_edge_lin1_bwd_x(const float* dH1, int64_t lddh, int64_t E, int64_t C,
                                     const float* W1, int64_t ldw, int64_t Cin, float* dG,
                                     int64_t lddg, yolat_stream_t stream) {
  if (E < 0 || C <= 0 || Cin <= 0 || !W1) return YOLAT_E_INVALID;
  if (E == 0) return 0;
  if (!dH1 || !dG || lddh < C || lddg < 2 * Cin || ldw < 2 * Cin + 4 || E >= (1LL << 31))
    return YOLAT_E_INVALID;
  DenseOp a = yl_dense(dH1, lddh, E, C);
  EdgeWcOp b;
  b.W1 = W1; b.ldw = ldw; b.Cin = (int)Cin; b.C = (int)C; b.vec = 1;
  Epilogue ep;
  ep.bias = nullptr; ep.scale = nullptr; ep.shift = nullptr; ep.relu = 0;
  ep.Y = dG; ep.ldy = lddg; ep.accumulate = 0; ep.stats = nullptr; ep.seg = nullptr; ep.pool = nullptr; ep.ldpool = 0;
  const long Nn = 2 * Cin;
  dim3 grid(yl_cdiv(E, 64), yl_cdiv(Nn, 64));
  hipLaunchKernelGGL((k_gemm_nt<64, 64, 32, DenseOp, EdgeWcOp, true>), grid, dim3(256), 0,
                     (hipStream_t)stream, a, b, ep, (int)E, (int)Nn, (int)C);
  YL_LAUNCH_CHECK();
  return 0;
}

// ------------------------------------------------------------------------------------------------
// CSR mean aggregation.  One wave per destination node, lanes across channels (256-B rows are
// read fully coalesced); rows of a node are consecutive CSR slots, summed in ascending edge order.
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_csr_mean_fwd(const float* H, long ldh, int C,
                                                      const float* hs, const float* hb, int relu,
                                                      const int* row_ptr, int N, float* out,
                                                      long ldo, int accumulate) {
  const int lane = threadIdx.x & 63;
  const int n = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (n >= N) return;
  const int q0 = row_ptr[n], q1 = row_ptr[n + 1];
  const float inv = 1.f / (float)((q1 - q0) > 1 ? (q1 - q0) : 1);
  for (int c = lane; c < C; c += 64) {
    const float sc = hs ? hs[c] : 1.f, sh = hs ? hb[c] : 0.f;
    const float floor = relu ? 0.f : -INFINITY;
    float s = 0.f;
    int q = q0;
    for (; q + 4 <= q1; q += 4) {      // 4 independent row loads in flight; summation stays in edge order
      float v[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) v[j] = H[(long)(q + j) * ldh + c];
#pragma unroll
      for (int j = 0; j < 4; ++j) s += fmaxf(fmaf(v[j], sc, sh), floor);
    }
    for (; q < q1; ++q) s += fmaxf(fmaf(H[(long)q * ldh + c], sc, sh), floor);
    s = yl_mul_rn(s, inv);
    float* o = out + (long)n * ldo + c;
    if (accumulate) s += *o;
    *o = s;
  }
}

// C == 4*LPN, 16-byte aligned rows: LPN lanes own one node (one float4 of columns each), 64/LPN nodes per
// wave, up to 8 rows (8 x 16 B per lane) in flight per lane.  Per-column summation order is unchanged
// (ascending CSR slot), so results are bit-identical to the scalar kernel above.
template <int LPN>
__global__ void __launch_bounds__(256) k_csr_mean_fwd_v4(const float* __restrict__ H, long ldh,
                                                         const float* hs, const float* hb, int relu,
                                                         const int* __restrict__ row_ptr, int N,
                                                         float* out, long ldo, int accumulate) {
  const int sub = threadIdx.x % LPN;
  const int n = blockIdx.x * (256 / LPN) + threadIdx.x / LPN;
  if (n >= N) return;
  const int q0 = row_ptr[n], q1 = row_ptr[n + 1];
  const float inv = 1.f / (float)((q1 - q0) > 1 ? (q1 - q0) : 1);
  float4 sc = make_float4(1.f, 1.f, 1.f, 1.f), sh = make_float4(0.f, 0.f, 0.f, 0.f);
  if (hs) {
    sc = *reinterpret_cast<const float4*>(hs + 4 * sub);
    sh = *reinterpret_cast<const float4*>(hb + 4 * sub);
  }
  const float floor = relu ? 0.f : -INFINITY;
  const float* hp = H + 4 * sub;
  float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
  auto add = [&](const float4& v) {
    s.x += fmaxf(fmaf(v.x, sc.x, sh.x), floor);
    s.y += fmaxf(fmaf(v.y, sc.y, sh.y), floor);
    s.z += fmaxf(fmaf(v.z, sc.z, sh.z), floor);
    s.w += fmaxf(fmaf(v.w, sc.w, sh.w), floor);
  };
  int q = q0;
  for (; q + 8 <= q1; q += 8) {
    float4 v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = *reinterpret_cast<const float4*>(hp + (long)(q + j) * ldh);
#pragma unroll
    for (int j = 0; j < 8; ++j) add(v[j]);
  }
  if (q < q1) {   // 1..7 remaining rows: clamped (re-read) addresses keep the loads unconditional
    float4 v[7];
#pragma unroll
    for (int j = 0; j < 7; ++j) v[j] = *reinterpret_cast<const float4*>(hp + (long)yl_min(q + j, q1 - 1) * ldh);
#pragma unroll
    for (int j = 0; j < 7; ++j)
      if (q + j < q1) add(v[j]);
  }
  s.x = yl_mul_rn(s.x, inv); s.y = yl_mul_rn(s.y, inv); s.z = yl_mul_rn(s.z, inv); s.w = yl_mul_rn(s.w, inv);
  float4* o = reinterpret_cast<float4*>(out + (long)n * ldo + 4 * sub);
  if (accumulate) {
    const float4 p = *o;
    s.x += p.x; s.y += p.y; s.z += p.z; s.w += p.w;
  }
  *o = s;
}

extern "C" int yolat_csr_mean_fwd(const float* H, int64_t ldh, int64_t C, const float* h_scale,
                                  const float* h_shift, int h_relu, const int32_t* row_ptr,
                                  int64_t N, float* out, int64_t ldo, int accumulate,
                                  yolat_stream_t stream) {
  if (N <= 0 || C <= 0 || !row_ptr || !out || ldo < C) return YOLAT_E_INVALID;
  if ((h_scale == nullptr) != (h_shift == nullptr)) return YOLAT_E_INVALID;
  const bool al16 = ((uintptr_t)H % 16 == 0) && ((uintptr_t)out % 16 == 0) && ldh % 4 == 0 && ldo % 4 == 0 &&
                    (!h_scale || ((uintptr_t)h_scale % 16 == 0 && (uintptr_t)h_shift % 16 == 0));
  if (C == 64 && al16) {
    hipLaunchKernelGGL(k_csr_mean_fwd_v4<16>, dim3(yl_cdiv(N, 16)), dim3(256), 0, (hipStream_t)stream, H,
                       (long)ldh, h_scale, h_shift, h_relu, row_ptr, (int)N, out, (long)ldo, accumulate);
    YL_LAUNCH_CHECK();
    return 0;
  }
  hipLaunchKernelGGL(k_csr_mean_fwd, dim3(yl_cdiv(N, 4)), dim3(256), 0, (hipStream_t)stream, H,
                     (long)ldh, (int)C, h_scale, h_shift, h_relu, row_ptr, (int)N, out, (long)ldo,
                     accumulate);
  YL_LAUNCH_CHECK();
  return 0;
}

__global__ void __launch_bounds__(256) k_csr_mean_bwd(const float* dOut, long lddo, int C,
                                                      const int* row_ptr, const int* dst, int E,
                                                      float* dM, long lddm) {
  const int lane = threadIdx.x & 63;
  const int q = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (q >= E) return;
  const int n = dst[q];
  const int deg = row_ptr[n + 1] - row_ptr[n];
  const float inv = 1.f / (float)(deg > 1 ? deg : 1);
  for (int c = lane; c < C; c += 64) dM[(long)q * lddm + c] = dOut[(long)n * lddo + c] * inv;
}

// C == 64, aligned rows: 16 lanes x float4 per edge, a wave writes 4 rows (1 KiB) per store instead of 256 B
__global__ void __launch_bounds__(256) k_csr_mean_bwd_v4(const float* __restrict__ dOut, long lddo,
                                                         const int* __restrict__ row_ptr, const int* __restrict__ dst,
                                                         int E, float* __restrict__ dM, long lddm) {
  const int l16 = threadIdx.x & 15;
  const int q = blockIdx.x * 16 + (threadIdx.x >> 4);
  if (q >= E) return;
  const int n = dst[q];
  const int deg = row_ptr[n + 1] - row_ptr[n];
  const float inv = 1.f / (float)(deg > 1 ? deg : 1);
  const float4 g = *reinterpret_cast<const float4*>(dOut + (long)n * lddo + 4 * l16);
  *reinterpret_cast<float4*>(dM + (long)q * lddm + 4 * l16) = make_float4(g.x * inv, g.y * inv, g.z * inv, g.w * inv);
}

extern "C" int yolat_csr_mean_bwd(const float* dOut, int64_t lddo, int64_t C,
                                  const int32_t* row_ptr, const int32_t* dst_csr, int64_t E,
                                  float* dM, int64_t lddm, yolat_stream_t stream) {
  if (E < 0 || C <= 0 || !dOut || !row_ptr) return YOLAT_E_INVALID;
  if (E == 0) return 0;
  if (!dst_csr || !dM || lddm < C) return YOLAT_E_INVALID;
  if (C == 64 && lddo % 4 == 0 && lddm % 4 == 0 && yl_aligned16(dOut) && yl_aligned16(dM))
    hipLaunchKernelGGL(k_csr_mean_bwd_v4, dim3(yl_cdiv(E, 16)), dim3(256), 0, (hipStream_t)stream, dOut, (long)lddo,
                       row_ptr, dst_csr, (int)E, dM, (long)lddm);
  else
  hipLaunchKernelGGL(k_csr_mean_bwd, dim3(yl_cdiv(E, 4)), dim3(256), 0, (hipStream_t)stream, dOut,
                     (long)lddo, (int)C, row_ptr, dst_csr, (int)E, dM, (long)lddm);
  YL_LAUNCH_CHECK();
  return 0;
}

// ------------------------------------------------------------------------------------------------
// Backward of the two gathers: dX[n] (+)= sum_{q in CSR row n} dG[q, 0:Cin]
//                                       + sum_{q in CSC col n} dG[q, Cin:2Cin]   (ascending slots)
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_edge_scatter_bwd(const float* dG, long lddg, int Cin,
                                                          const int* row_ptr, const int* col_ptr,
                                                          const int* slots, int N, float* dX,
                                                          long lddx, int accumulate) {
  const int lane = threadIdx.x & 63;
  const int n = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (n >= N) return;
  const int q0 = row_ptr[n], q1 = row_ptr[n + 1];
  const int t0 = col_ptr[n], t1 = col_ptr[n + 1];
  for (int c = lane; c < Cin; c += 64) {
    float s = 0.f;
    for (int q = q0; q < q1; ++q) s += dG[(long)q * lddg + c];
    for (int t = t0; t < t1; ++t) s += dG[(long)slots[t] * lddg + Cin + c];
    float* o = dX + (long)n * lddx + c;
    if (accumulate) s += *o;
    *o = s;
  }
}

// Cin == 64, aligned rows: 16 lanes x float4 per node (a wave covers 4 nodes), 4 row loads in flight per thread;
// same summation order as the scalar kernel (CSR slots ascending, then CSC slots ascending)
__global__ void __launch_bounds__(256) k_edge_scatter_bwd_v4(const float* __restrict__ dG, long lddg,
                                                             const int* __restrict__ row_ptr,
                                                             const int* __restrict__ col_ptr,
                                                             const int* __restrict__ slots, int N, float* dX,
                                                             long lddx, int accumulate) {
  const int l16 = threadIdx.x & 15;
  const int n = blockIdx.x * 16 + (threadIdx.x >> 4);
  if (n >= N) return;
  const int q0 = row_ptr[n], q1 = row_ptr[n + 1];
  const int t0 = col_ptr[n], t1 = col_ptr[n + 1];
  float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
  const float* gd = dG + 4 * l16;
  for (int q = q0; q < q1; q += 4) {
    float4 v[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) v[k] = *reinterpret_cast<const float4*>(gd + (long)yl_min(q + k, q1 - 1) * lddg);
#pragma unroll
    for (int k = 0; k < 4; ++k)
      if (q + k < q1) { s.x += v[k].x; s.y += v[k].y; s.z += v[k].z; s.w += v[k].w; }
  }
  const float* gs = dG + 64 + 4 * l16;
  for (int t = t0; t < t1; t += 4) {
    int sl[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) sl[k] = slots[yl_min(t + k, t1 - 1)];
    float4 v[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) v[k] = *reinterpret_cast<const float4*>(gs + (long)sl[k] * lddg);
#pragma unroll
    for (int k = 0; k < 4; ++k)
      if (t + k < t1) { s.x += v[k].x; s.y += v[k].y; s.z += v[k].z; s.w += v[k].w; }
  }
  float4* o = reinterpret_cast<float4*>(dX + (long)n * lddx + 4 * l16);
  if (accumulate) { const float4 d = *o; s.x += d.x; s.y += d.y; s.z += d.z; s.w += d.w; }
  *o = s;
}

extern "C" int yolat_edge_scatter_bwd(const float* dG, int64_t lddg, int64_t Cin,
                                      const int32_t* row_ptr, const int32_t* col_ptr,
                                      const int32_t* slots, int64_t N, float* dX, int64_t lddx,
                                      int accumulate, yolat_stream_t stream) {
  if (N <= 0 || Cin <= 0 || !row_ptr || !col_ptr || !dX || lddx < Cin) return YOLAT_E_INVALID;
  if (Cin == 64 && lddg % 4 == 0 && lddx % 4 == 0 && yl_aligned16(dG) && yl_aligned16(dX))
    hipLaunchKernelGGL(k_edge_scatter_bwd_v4, dim3(yl_cdiv(N, 16)), dim3(256), 0, (hipStream_t)stream, dG, (long)lddg,
                       row_ptr, col_ptr, slots, (int)N, dX, (long)lddx, accumulate);
  else
  hipLaunchKernelGGL(k_edge_scatter_bwd, dim3(yl_cdiv(N, 4)), dim3(256), 0, (hipStream_t)stream,
                     dG, (long)lddg, (int)Cin, row_ptr, col_ptr, slots, (int)N, dX, (long)lddx,
                     accumulate);
  YL_LAUNCH_CHECK();
  return 0;
}
