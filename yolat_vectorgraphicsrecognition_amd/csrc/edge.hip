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
  const float4 bb = b1 ? *reinterpret_cast<const float4*>(b1 + 4 * q) : make_float4(0.f, 0.f, 0.f, 0.f);
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
#ifdef YOLAT_EDGE_STAMPS
// debug build only (tools/exp/r06_edge_stamps.sh): wall-clock stamps (100 MHz) of thread 0 of every node-tile workgroup.
// The stamped build is for the STRUCTURE of a workgroup's time; a stamp is a scalar memory read + a wait + a store, and the
// build that carries them schedules differently (round 6: two load re-orderings that cut the stamped workgroup from 13.4 to
// 9.2 us left the product build's launch where it was, by rocprof) — changes are judged by A/B of product builds.
__device__ long long edge_stamps_d[2][4096 * 16];
#define EDGE_STAMP(k) do { if (threadIdx.x == 0 && blockIdx.x < 4096) edge_stamps_d[nx.Wp != nullptr ? 0 : 1][blockIdx.x * 16 + (k)] = wall_clock64(); } while (0)
extern "C" int yolat_debug_edge_stamps(long long* out, int n) {
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(edge_stamps_d), sizeof(long long) * (size_t)n);
}
#else
#define EDGE_STAMP(k) do { } while (0)
#endif
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
                                                           int E, int tiles, PoolRider rider, EdgeNext nx) {
  // workgroups past the node tiles: the next layer's node-branch tiles (EdgeNext), then the pooling-prologue rider
  // (common.hpp) — both independent of this layer's messages
  if ((int)blockIdx.x >= tiles + nx.s_tiles) {
    yl_pool_rider(rider, blockIdx.x - tiles - nx.s_tiles, rider.blocks, threadIdx.x, 256);
    return;
  }
  constexpr int LDH = 65;
  __shared__ float Hs[64 * LDH];      // layer-1 activations of the pass, then the layer-2 messages
  __shared__ float W2s[64 * LDH];
  __shared__ int rp[16 * NG + 1];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int wm = wave >> 1, wn = wave & 1, l31 = lane & 31, lhi = lane >> 5;
  if ((int)blockIdx.x >= tiles) {
    // s'[r0 .. r0+63] = relu(((s . Wn'^T) + bn') * sn' + tn'): one 64 x 64 x 64 tile on the same LDS tiles and the same
    // MFMA loop (k ascending in pairs) as layer 2 below — bit-identical to k_gemm_nt_node3's tile
    const int r0 = (blockIdx.x - tiles) * 64;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const int i = tid + t * 256, row = i >> 4, c4 = 4 * (i & 15);
      const float4 av = *reinterpret_cast<const float4*>(nx.s_in + (long)yl_min(r0 + row, N - 1) * nx.ld_si + c4);
      const float4 wv = *reinterpret_cast<const float4*>(nx.Wn + row * 64 + c4);
      float* da = Hs + row * LDH + c4;
      float* dw = W2s + row * LDH + c4;
      da[0] = av.x; da[1] = av.y; da[2] = av.z; da[3] = av.w;
      dw[0] = wv.x; dw[1] = wv.y; dw[2] = wv.z; dw[3] = wv.w;
    }
    const int colS = wn * 32 + l31;
    const float bS = nx.bn ? nx.bn[colS] : 0.f, scS = nx.sn[colS], shS = nx.tn[colS];
    __syncthreads();
    f32x16 accS;
#pragma unroll
    for (int r = 0; r < 16; ++r) accS[r] = 0.f;
#pragma unroll 8
    for (int kk = 0; kk < 64; kk += 2) {
      const float av = Hs[(wm * 32 + l31) * LDH + kk + lhi];
      const float bv = W2s[(wn * 32 + l31) * LDH + kk + lhi];
      accS = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, accS, 0, 0, 0);
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = r0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhi;
      if (row < N) nx.s_out[(long)row * nx.ld_so + colS] = fmaxf(fmaf(accS[r] + bS, scS, shS), 0.f);
    }
    return;
  }
  const int n0 = blockIdx.x * npt;
  const int nn = yl_min(npt, N - n0);                 // nodes of this tile
  EDGE_STAMP(0);
  if (tid <= 16 * NG) rp[tid] = row_ptr[yl_min(n0 + tid, n0 + nn)];
  const int q = tid & 15, rb = tid >> 4;              // gather role: columns 4q..4q+3 of rows rb + 16t
  const int col = wn * 32 + l31;                      // MFMA role: output column of this lane
  // the tile's edge range straight from global (same two addresses in every lane: one broadcast load each), so that
  // the first pass's index loads go out before the barrier below instead of after it, and the f_out rows this thread
  // finishes with at the very end: both shorten the workgroup's chain of dependent global round trips, which is what
  // a small graph's launch time consists of (715 workgroups, all resident at once, at E = 40 k)
  const int e0g = row_ptr[n0], e1g = row_ptr[n0 + nn];
  int di0[4], si0[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    const int e = yl_min(e0g + rb + 16 * t, E - 1);
    di0[t] = dst[e]; si0[t] = src[e];
  }
  float4 fo[NG];
#pragma unroll
  for (int j = 0; j < NG; ++j)
    fo[j] = *reinterpret_cast<const float4*>(f_out + (long)yl_min(n0 + rb + 16 * j, N - 1) * ld_fo + 4 * q);
  float rw2[4][4];
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    const int i = tid + t * 256;
    W2.template load4<false>(i >> 4, 4 * (i & 15), rw2[t]);
  }
  float4 wc[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) wc[j] = *reinterpret_cast<const float4*>(Wc4 + (4 * q + j) * 4);
  const float4 bb = b1 ? *reinterpret_cast<const float4*>(b1 + 4 * q) : make_float4(0.f, 0.f, 0.f, 0.f);
  float4 sc = make_float4(1.f, 1.f, 1.f, 1.f), sh = make_float4(0.f, 0.f, 0.f, 0.f);
  if (s1) { sc = *reinterpret_cast<const float4*>(s1 + 4 * q); sh = *reinterpret_cast<const float4*>(t1 + 4 * q); }
  const float bias2 = b2 ? b2[col] : 0.f, sc2 = s2 ? s2[col] : 1.f, sh2 = s2 ? t2[col] : 0.f;
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    const int i = tid + t * 256;
    float* d = W2s + (i >> 4) * LDH + 4 * (i & 15);
    d[0] = rw2[t][0]; d[1] = rw2[t][1]; d[2] = rw2[t][2]; d[3] = rw2[t][3];
  }
  // next layer's node side (NG == 1): the first column tile's B fragments travel with the loads above, the other two
  // are fetched under the previous tile's MFMAs (see the end of the kernel)
  float bf0[16];
  if constexpr (NG == 1) {
    if (nx.Wp != nullptr) {
#pragma unroll
      for (int ks = 0; ks < 16; ++ks) bf0[ks] = nx.Wp[(wave * 16 + ks) * 64 + lane];
    }
  }
  EDGE_STAMP(1);
  __syncthreads();
  EDGE_STAMP(2);
  const int e0 = e0g, e1 = e1g;
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
      if (c0 == e0) { di[t] = di0[t]; si[t] = si0[t]; }          // loaded before the barrier
      else {
        const int e = yl_min(c0 + rb + 16 * t, E - 1);
        di[t] = dst[e]; si[t] = src[e];
      }
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
    if (c0 == e0) EDGE_STAMP(3);
    __syncthreads();
    if (c0 == e0) EDGE_STAMP(4);
    f32x16 acc2;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc2[r] = 0.f;
#pragma unroll 8
    for (int kk = 0; kk < 64; kk += 2) {
      const float av = Hs[(wm * 32 + l31) * LDH + kk + lhi];
      const float bv = W2s[(wn * 32 + l31) * LDH + kk + lhi];
      acc2 = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc2, 0, 0, 0);
    }
    if (c0 == e0) EDGE_STAMP(5);
    __syncthreads();                      // every wave is done reading Hs as the layer-1 tile
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhi;
      Hs[row * LDH + col] = fmaxf(fmaf(acc2[r] + bias2, sc2, sh2), 0.f);
    }
    __syncthreads();
    if (c0 == e0) EDGE_STAMP(6);
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
    if (c0 == e0) EDGE_STAMP(7);
    __syncthreads();
  }
  EDGE_STAMP(8);
#pragma unroll
  for (int j = 0; j < NG; ++j) {
    const int deg = my_e[j] - my_b[j];
    if (rb + 16 * j < nn && deg > 0) {
      const float inv = 1.f / (float)deg;
      float4* o = reinterpret_cast<float4*>(f_out + (long)(n0 + rb + 16 * j) * ld_fo + 4 * q);
      float4 d = fo[j];                                // this workgroup is the only writer of its nodes' rows
      // explicit mul then add (no fma contraction): the same two roundings as k_csr_mean_fwd*
      d.x = yl_mul_rn(sum[j].x, inv) + d.x; d.y = yl_mul_rn(sum[j].y, inv) + d.y;
      d.z = yl_mul_rn(sum[j].z, inv) + d.z; d.w = yl_mul_rn(sum[j].w, inv) + d.w;
      *o = d;
      fo[j] = d;
    }
  }
  EDGE_STAMP(9);
  // ---- node side of the NEXT layer for this tile's nodes (EdgeNext, common.hpp): [nn <= 16, 64] x [192, 64]^T
  if constexpr (NG == 1) {
    if (nx.Wp != nullptr) {
      // the loop above ended with a barrier (or never ran): Hs is free; rows >= nn hold the clamped load of row N-1 or
      // stale sums — finite or not, their products only reach output rows that are never stored
      float* frow = Hs + rb * LDH + 4 * q;
      const float4 fz = (rb < nn) ? fo[0] : make_float4(0.f, 0.f, 0.f, 0.f);
      frow[0] = fz.x; frow[1] = fz.y; frow[2] = fz.z; frow[3] = fz.w;
      __syncthreads();
      const int fr = lane & 15, fk = lane >> 4;
      float bfa[16], bfb[16];
#pragma unroll
      for (int ks = 0; ks < 16; ++ks) bfa[ks] = bf0[ks];
#pragma unroll
      for (int t = 0; t < 3; ++t) {
        const int ct = wave + 4 * t;                   // 12 column tiles of 16: UV' 0..7, root' 8..11
        if (t < 2) {
#pragma unroll
          for (int ks = 0; ks < 16; ++ks) bfb[ks] = nx.Wp[((ct + 4) * 16 + ks) * 64 + lane];
        }
        const int col = ct * 16 + fr;
        const float bias = nx.bias[col];
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < 16; ++ks)
          acc = __builtin_amdgcn_mfma_f32_16x16x4f32(Hs[fr * LDH + 4 * ks + fk], bfa[ks], acc, 0, 0, 0);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int row = 4 * fk + r;
          if (row < nn) {
            const float v = acc[r] + bias;
            if (col < 128) nx.UV[(long)(n0 + row) * nx.ld_uv + col] = v;
            else nx.root[(long)(n0 + row) * nx.ld_root + (col - 128)] = v;
          }
        }
#pragma unroll
        for (int ks = 0; ks < 16; ++ks) bfa[ks] = bfb[ks];
      }
    }
  }
  EDGE_STAMP(10);
#ifdef YOLAT_EDGE_STAMPS
  if (threadIdx.x == 0 && blockIdx.x < 4096) edge_stamps_d[nx.Wp != nullptr ? 0 : 1][blockIdx.x * 16 + 11] = (long long)((e1g - e0g + 63) / 64);
#endif
}

// ------------------------------------------------------------------------------------------------
// Wave-specialised factorised edge MLP + mean aggregation (round 2) — the large-graph path.
// Why (profiles/r02_base_fwd_cfg5_pmc_sq_*.txt and the phase ablation in profiles/r02_edge_ablation.txt, cfg 5): the
// node-tiled kernel above runs its phases — gather, layer-1 VALU, layer-2 MFMA, aggregation — one after the other
// in every wave, and because all resident workgroups start together and contend for the same pipes they stay in
// lockstep: the phase times ADD (87 us without the MFMAs + 105 us of MFMA phase = 192 us) although the MFMA pipe
// needs only ~75 us of it; its 64-row passes are also only 75 % full and W2 is re-staged for every 96 edges.
// Here a 512-thread workgroup owns the edge range [w*chunk, (w+1)*chunk) moved to node boundaries (the node holding
// edge w*chunk is dst[w*chunk]: no search) and walks it in FULL 64-edge passes with two kinds of waves:
//   waves 0-3 "producers": per pass the row gathers U[dst] + V[src] + attr (issued one pass ahead of their use, the
//              indices two passes ahead), layer 1 on the VALU -> LDS tile Hs[(p+1)&1], and the per-node mean of the
//              messages of pass p-1 (segment table built from dst with one ballot per pass; the running sum of a
//              node that straddles two passes travels through 65 floats of LDS);
//   waves 4-7 "consumers": W2's MFMA B fragments in 32 registers for the whole kernel, per pass 8 ds_read_b128 (the
//              tile is stored k-parity-interleaved, row stride 68: one conflict-free read feeds four MFMAs) +
//              32 v_mfma_f32_32x32x2_f32 on Hs[p&1], BN+ReLU epilogue -> Ms[p&1].
// One s_barrier per pass; every SIMD hosts one producer and one consumer wave of each of the two resident
// workgroups.
//
// X6 = false: layer 2 on v_mfma_f32_32x32x2_f32 — the same arithmetic in the same order as k_edge_uv_mlp2 +
//   k_csr_mean_fwd (bit-identical results).  Measured 180 us at cfg 5 against 213 us for the node tiles: on gfx950 the
//   fp32-input MFMA executes on the SIMD's VECTOR ALUs (tools/exp/pipe_overlap.hip: the MFMA time and the VALU time
//   of two waves on one SIMD ADD, 124 + 104 -> 222 us), so specialising waves cannot overlap the two, and a single
//   VALU-heavy wave per SIMD issues one instruction per ~7 cycles (tools/exp/valu_rate.hip).
// X6 = true (the product path for large graphs): layer 2 as an fp32 GEMM EMULATED on the bf16 matrix cores, which
//   do run beside the vector ALUs: a = a_h + a_m + a_l with three bfloat16 terms (truncation splits, 8 + 8 + 8 = 24
//   significand bits: the split is exact), a.b = a_h b_h + (a_h b_m + a_m b_h + a_h b_l + a_l b_h + a_m b_m) + O(2^-24):
//   six v_mfma_f32_32x32x16_bf16 (32 cycles each) per 16 k instead of eight fp32 MFMAs (64 cycles each), exact
//   products, fp32 accumulation.  The result differs from the fp32-MFMA kernels only by the summation order
//   (measured ~1e-7 relative); tests/test_gpu_ops.py compares the two at 2e-6 of scale.
// FOLD: layer 1's bias / BatchNorm live in UV and Wc4 (node-side epilogue; b1 = s1 = t1 = NULL) and layer 2's bias
//   in t2 (b2 = NULL): h1 = relu(U + V + Wc4.attr), message = relu(s2 * (W2.h1) + t2) — 6 instead of 8 VALU
//   operations per hidden activation (packed fp32 math) and 2 instead of 3 per message.
// ------------------------------------------------------------------------------------------------
typedef __bf16 yl_bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned yl_u32x4 __attribute__((ext_vector_type(4)));
// exact 3-way bfloat16 split of 8 fp32 values (truncation: every term keeps the next 8 significand bits)
__device__ __forceinline__ void yl_split8(const float x[8], yl_bf16x8& h, yl_bf16x8& m, yl_bf16x8& l) {
  yl_u32x4 ph, pm, pl;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const unsigned x0 = __float_as_uint(x[2 * i]), x1 = __float_as_uint(x[2 * i + 1]);
    // h = top 8 significand bits, hm = top 16: m = hm - h and l = x - hm are exact and need 8 bits each
    const yl_f32x2 xv = {x[2 * i], x[2 * i + 1]};
    const yl_f32x2 hv = {__uint_as_float(x0 & 0xffff0000u), __uint_as_float(x1 & 0xffff0000u)};
    const yl_f32x2 hmv = {__uint_as_float(x0 & 0xffffff00u), __uint_as_float(x1 & 0xffffff00u)};
    const yl_f32x2 mv = hmv - hv, lv = xv - hmv;     // v_pk_add_f32 with negated operand
    ph[i] = __builtin_amdgcn_perm(x1, x0, 0x07060302u);
    pm[i] = __builtin_amdgcn_perm(__float_as_uint(mv.y), __float_as_uint(mv.x), 0x07060302u);
    pl[i] = __builtin_amdgcn_perm(__float_as_uint(lv.y), __float_as_uint(lv.x), 0x07060302u);
  }
  h = *reinterpret_cast<yl_bf16x8*>(&ph);
  m = *reinterpret_cast<yl_bf16x8*>(&pm);
  l = *reinterpret_cast<yl_bf16x8*>(&pl);
}

template <bool FOLD, bool X6>
__global__ void __launch_bounds__(512, 4) k_edge_uv_mlp2_mean_ws(const float* __restrict__ UV, long ld_uv,
                                                                 const int* __restrict__ src,
                                                                 const int* __restrict__ dst,
                                                                 const float* __restrict__ attr,
                                                                 const int* __restrict__ row_ptr, int N, int E, int chunk,
                                                                 const float* __restrict__ Wc4,
                                                                 const float* __restrict__ b1,
                                                                 const float* __restrict__ s1,
                                                                 const float* __restrict__ t1,
                                                                 const float* __restrict__ W2,
                                                                 const float* __restrict__ b2,
                                                                 const float* __restrict__ s2,
                                                                 const float* __restrict__ t2, float* f_out, long ld_fo) {
  constexpr int LDH = 68, TILE = 64 * LDH;
  __shared__ __attribute__((aligned(16))) float Hs[2 * TILE];   // layer-1 activations, k-parity-interleaved
  __shared__ __attribute__((aligned(16))) float Ms[2 * TILE];   // layer-2 messages, row-major
  __shared__ int seg_start[2][66];                              // first row of the k-th node of the pass (+ end)
  __shared__ int seg_node[2][64];
  __shared__ int seg_info[2][4];                                // #nodes, first continues from / last continues into
  __shared__ __attribute__((aligned(16))) float carry_s[2][64];
  __shared__ int carry_cnt_s[2];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  // ---- this workgroup's edge range, moved to node boundaries (uniform)
  const long c_lo = (long)blockIdx.x * chunk;
  if (c_lo >= E) return;
  const long c_hi = c_lo + chunk;
  int eS = 0, eE = E;
  if (c_lo > 0) {
    const int n = dst[c_lo];
    eS = (row_ptr[n] == (int)c_lo) ? (int)c_lo : row_ptr[n + 1];
  }
  if (c_hi < E) {
    const int n = dst[c_hi];
    eE = (row_ptr[n] == (int)c_hi) ? (int)c_hi : row_ptr[n + 1];
  }
  if (eS >= eE) return;
  const int np = (eE - eS + 63) >> 6;

  if (wave >= 4) {
    // ================================ consumers: layer 2 on the matrix cores ================================
    const int cw = wave - 4, wm = cw >> 1, wn = cw & 1, l31 = lane & 31, lhi = lane >> 5;
    const int col = wn * 32 + l31;
    const float bias2 = (!FOLD && b2) ? b2[col] : 0.f, sc2 = s2 ? s2[col] : 1.f, sh2 = s2 ? t2[col] : 0.f;
    auto store_messages = [&](const f32x16& acc, int p) {
      float* ms = Ms + (p & 1) * TILE;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhi;
        ms[row * LDH + col] = FOLD ? fmaxf(fmaf(acc[r], sc2, sh2), 0.f) : fmaxf(fmaf(acc[r] + bias2, sc2, sh2), 0.f);
      }
    };
    if (X6) {
      // W2[col][16 ks + 8 lhi + 0..7] as three bfloat16 fragments per k step, resident for the whole kernel.
      // FOLD: the BatchNorm scale s2[col] is multiplied into W2's row before the split and the shift t2[col] is the
      // accumulator's initial value, so the epilogue is one add (the two accumulators) and the ReLU.
      yl_bf16x8 Bh[4], Bm[4], Bl[4];
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        const float4* wp = reinterpret_cast<const float4*>(W2 + (long)col * 64 + 16 * ks + 8 * lhi);
        const float4 w0 = wp[0], w1 = wp[1];
        float x[8] = {w0.x, w0.y, w0.z, w0.w, w1.x, w1.y, w1.z, w1.w};
        if (FOLD) {
#pragma unroll
          for (int i = 0; i < 8; ++i) x[i] *= sc2;
        }
        yl_split8(x, Bh[ks], Bm[ks], Bl[ks]);
      }
      __syncthreads();                                // barrier 0: Hs[0] is complete
      for (int p = 0; p < np; ++p) {
        const float* arow = Hs + (p & 1) * TILE + (wm * 32 + l31) * LDH + 8 * lhi;
        f32x16 accM, accS;
#pragma unroll
        for (int r = 0; r < 16; ++r) { accM[r] = FOLD ? sh2 : 0.f; accS[r] = 0.f; }
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
          const float4 a0 = *reinterpret_cast<const float4*>(arow + 16 * ks);
          const float4 a1 = *reinterpret_cast<const float4*>(arow + 16 * ks + 4);
          const float x[8] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w};
          yl_bf16x8 Ah, Am, Al;
          yl_split8(x, Ah, Am, Al);
          // two accumulators, three products each, alternating: no MFMA waits on the one issued just before it
          accS = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Al, Bh[ks], accS, 0, 0, 0);
          accM = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Ah, Bl[ks], accM, 0, 0, 0);
          accS = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Am, Bh[ks], accS, 0, 0, 0);
          accM = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Am, Bm[ks], accM, 0, 0, 0);
          accS = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Ah, Bm[ks], accS, 0, 0, 0);
          accM = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Ah, Bh[ks], accM, 0, 0, 0);
        }
        float* ms = Ms + (p & 1) * TILE;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhi;
          const float z = accM[r] + accS[r];
          ms[row * LDH + col] = FOLD ? fmaxf(z, 0.f) : fmaxf(fmaf(z + bias2, sc2, sh2), 0.f);
        }
        __syncthreads();                              // barrier p+1: Ms[p&1] complete, Hs[p&1] free
      }
      return;
    }
    float bfrag[32];                                  // bfrag[m] = W2[col][2m + lhi]
    {
      const float4* wrow = reinterpret_cast<const float4*>(W2 + (long)col * 64);
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        const float4 f = wrow[j];
        bfrag[2 * j] = lhi ? f.y : f.x;
        bfrag[2 * j + 1] = lhi ? f.w : f.z;
      }
    }
    __syncthreads();                                  // barrier 0: Hs[0] is complete
    for (int p = 0; p < np; ++p) {
      // fp32 path: Hs rows are k-parity-interleaved (position(k) = (k & 1) * 32 + (k >> 1)), so the A operands of four
      // consecutive MFMAs (k = 2m + lhi) are one ds_read_b128
      const float* arow = Hs + (p & 1) * TILE + (wm * 32 + l31) * LDH + lhi * 32;
      f32x16 acc2;
#pragma unroll
      for (int r = 0; r < 16; ++r) acc2[r] = 0.f;
#pragma unroll
      for (int m4 = 0; m4 < 8; ++m4) {
        const float4 av = *reinterpret_cast<const float4*>(arow + 4 * m4);
        acc2 = __builtin_amdgcn_mfma_f32_32x32x2f32(av.x, bfrag[4 * m4 + 0], acc2, 0, 0, 0);
        acc2 = __builtin_amdgcn_mfma_f32_32x32x2f32(av.y, bfrag[4 * m4 + 1], acc2, 0, 0, 0);
        acc2 = __builtin_amdgcn_mfma_f32_32x32x2f32(av.z, bfrag[4 * m4 + 2], acc2, 0, 0, 0);
        acc2 = __builtin_amdgcn_mfma_f32_32x32x2f32(av.w, bfrag[4 * m4 + 3], acc2, 0, 0, 0);
      }
      store_messages(acc2, p);
      __syncthreads();                                // barrier p+1: Ms[p&1] complete, Hs[p&1] free
    }
    return;
  }

  // ================================== producers: gathers, layer 1, mean ==================================
  const int q = tid & 15, rb = tid >> 4;              // gather role: columns 4q..4q+3 of rows rb + 16t;
                                                      // aggregation role: nodes rb, rb+16, .. of the pass, same columns
  float4 wc[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) wc[j] = *reinterpret_cast<const float4*>(Wc4 + (4 * q + j) * 4);
  float4 bb = make_float4(0.f, 0.f, 0.f, 0.f), sc = make_float4(1.f, 1.f, 1.f, 1.f), sh = make_float4(0.f, 0.f, 0.f, 0.f);
  if (!FOLD) {
    if (b1) bb = *reinterpret_cast<const float4*>(b1 + 4 * q);
    if (s1) { sc = *reinterpret_cast<const float4*>(s1 + 4 * q); sh = *reinterpret_cast<const float4*>(t1 + 4 * q); }
  }
  int di[4], si[4];
  float4 u[4], v[4], a[4];
  // every load below is unconditional (addresses clamped into the range), so the compiler can count the loads in
  // flight: a pass beyond the range re-reads the last edge and its tile is never consumed
  // (uniform base + 32-bit byte offset: one VALU operation per address; the host checks that UV / attr / the index
  // arrays are smaller than 4 GiB)
  const unsigned ldb = (unsigned)ld_uv * 4u, qb = 16u * q;
  auto load_idx = [&](int c0) {
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const unsigned eb = 4u * (unsigned)yl_min(c0 + rb + 16 * t, eE - 1);
      di[t] = *reinterpret_cast<const int*>(reinterpret_cast<const char*>(dst) + eb);
      si[t] = *reinterpret_cast<const int*>(reinterpret_cast<const char*>(src) + eb);
    }
  };
  auto gather = [&](int c0) {
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const unsigned e = (unsigned)yl_min(c0 + rb + 16 * t, eE - 1);
      u[t] = *reinterpret_cast<const float4*>(reinterpret_cast<const char*>(UV) + ((unsigned)di[t] * ldb + qb));
      v[t] = *reinterpret_cast<const float4*>(reinterpret_cast<const char*>(UV) + ((unsigned)si[t] * ldb + qb + 256u));
      a[t] = *reinterpret_cast<const float4*>(reinterpret_cast<const char*>(attr) + 16u * e);
    }
  };
  auto layer1 = [&](float* hs) {
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      auto one = [&](float uu, float vv, const float4& w, float b, float s, float h) {
        float z = uu + vv;
        z = fmaf(a[t].x, w.x, z); z = fmaf(a[t].y, w.y, z); z = fmaf(a[t].z, w.z, z); z = fmaf(a[t].w, w.w, z);
        return fmaxf(fmaf(z + b, s, h), 0.f);
      };
      float h0, h1, h2, h3;
      if (FOLD) {
        // packed fp32 math (v_pk_add_f32 / v_pk_fma_f32: two IEEE operations per instruction, same roundings)
        yl_f32x2 z02 = {u[t].x + v[t].x, u[t].z + v[t].z}, z13 = {u[t].y + v[t].y, u[t].w + v[t].w};
        const yl_f32x2 ax = {a[t].x, a[t].x}, ay = {a[t].y, a[t].y}, az = {a[t].z, a[t].z}, aw = {a[t].w, a[t].w};
        z02 = __builtin_elementwise_fma(ax, (yl_f32x2){wc[0].x, wc[2].x}, z02);
        z13 = __builtin_elementwise_fma(ax, (yl_f32x2){wc[1].x, wc[3].x}, z13);
        z02 = __builtin_elementwise_fma(ay, (yl_f32x2){wc[0].y, wc[2].y}, z02);
        z13 = __builtin_elementwise_fma(ay, (yl_f32x2){wc[1].y, wc[3].y}, z13);
        z02 = __builtin_elementwise_fma(az, (yl_f32x2){wc[0].z, wc[2].z}, z02);
        z13 = __builtin_elementwise_fma(az, (yl_f32x2){wc[1].z, wc[3].z}, z13);
        z02 = __builtin_elementwise_fma(aw, (yl_f32x2){wc[0].w, wc[2].w}, z02);
        z13 = __builtin_elementwise_fma(aw, (yl_f32x2){wc[1].w, wc[3].w}, z13);
        h0 = fmaxf(z02.x, 0.f); h2 = fmaxf(z02.y, 0.f); h1 = fmaxf(z13.x, 0.f); h3 = fmaxf(z13.y, 0.f);
      } else {
        h0 = one(u[t].x, v[t].x, wc[0], bb.x, sc.x, sh.x);
        h1 = one(u[t].y, v[t].y, wc[1], bb.y, sc.y, sh.y);
        h2 = one(u[t].z, v[t].z, wc[2], bb.z, sc.z, sh.z);
        h3 = one(u[t].w, v[t].w, wc[3], bb.w, sc.w, sh.w);
      }
      if (X6) {                                       // natural k order: the consumers read 8 consecutive k per lane
        *reinterpret_cast<float4*>(hs + (rb + 16 * t) * LDH + 4 * q) = make_float4(h0, h1, h2, h3);
      } else {
        float* hrow = hs + (rb + 16 * t) * LDH + 2 * q;
        *reinterpret_cast<float2*>(hrow) = make_float2(h0, h2);
        *reinterpret_cast<float2*>(hrow + 32) = make_float2(h1, h3);
      }
    }
  };
  // segment table of pass P from its 64 destination ids (lane = row); every producer wave writes the same values
  int dm_cur = dst[yl_min(eS + lane, eE - 1)], dm_nxt = dst[yl_min(eS + 64 + lane, eE - 1)], prev_last = -1;
  auto meta = [&](int P) {
    const int c0 = eS + 64 * P, bsel = P & 1;
    const int nrows = yl_min(64, eE - c0);
    const int dprev = __shfl_up(dm_cur, 1);
    const bool start = lane < nrows && (lane == 0 || dm_cur != dprev);
    const unsigned long long mask = __ballot(start);
    const int k = __popcll(mask & ((1ull << lane) - 1ull));
    if (start) { seg_start[bsel][k] = lane; seg_node[bsel][k] = dm_cur; }
    if (lane == 0) {
      const int nseg = __popcll(mask);
      seg_start[bsel][nseg] = nrows;
      seg_info[bsel][0] = nseg;
      seg_info[bsel][1] = (P > 0 && prev_last == dm_cur) ? 1 : 0;
    }
    const int first_next = __shfl(dm_nxt, 0);
    if (lane == 63) seg_info[bsel][2] = (c0 + 64 < eE && first_next == dm_cur) ? 1 : 0;
    prev_last = __shfl(dm_cur, 63);
    dm_cur = dm_nxt;
    dm_nxt = dst[yl_min(c0 + 128 + lane, eE - 1)];
  };
  // f_out row of this thread's FIRST node of pass P (the common case: <= 16 nodes per pass; further nodes load on
  // demand).  Issued one whole interval before aggregate(P) adds to it, unconditionally and clamped so that the
  // compiler can count the loads in flight: the wait in aggregate() then leaves the younger prefetches (next row
  // gathers, indices) in flight instead of draining them.  seg_node was written by this wave itself in meta(P).
  float4 fo0 = make_float4(0.f, 0.f, 0.f, 0.f);
  auto fo_load = [&](int P) {
    const int bsel = P & 1;
    const int k = yl_min(rb, seg_info[bsel][0] - 1);
    fo0 = *reinterpret_cast<const float4*>(f_out + (long)seg_node[bsel][k] * ld_fo + 4 * q);
  };
  auto aggregate = [&](int P) {
    const int bsel = P & 1;
    const float* ms = Ms + bsel * TILE + 4 * q;
    const int nseg = seg_info[bsel][0], cont_in = seg_info[bsel][1], cont_out = seg_info[bsel][2];
    for (int k = rb; k < nseg; k += 16) {
      const int lo = seg_start[bsel][k], hi = seg_start[bsel][k + 1];
      float4 sm = make_float4(0.f, 0.f, 0.f, 0.f);
      int cn = 0;
      if (k == 0 && cont_in) {
        sm = *reinterpret_cast<const float4*>(&carry_s[bsel ^ 1][4 * q]);
        cn = carry_cnt_s[bsel ^ 1];
      }
      for (int r = lo; r < hi; ++r) {
        const float4 m = *reinterpret_cast<const float4*>(ms + r * LDH);
        sm.x += m.x; sm.y += m.y; sm.z += m.z; sm.w += m.w;
      }
      cn += hi - lo;
      if (k == nseg - 1 && cont_out) {                // the node continues in the next pass
        *reinterpret_cast<float4*>(&carry_s[bsel][4 * q]) = sm;
        if (q == 0) carry_cnt_s[bsel] = cn;
      } else {
        const float inv = 1.f / (float)cn;
        float4* o = reinterpret_cast<float4*>(f_out + (long)seg_node[bsel][k] * ld_fo + 4 * q);
        float4 d = fo0;
        if (k != rb) d = *o;
        // explicit mul then add (no fma contraction): the same two roundings as k_csr_mean_fwd*
        d.x = yl_mul_rn(sm.x, inv) + d.x; d.y = yl_mul_rn(sm.y, inv) + d.y;
        d.z = yl_mul_rn(sm.z, inv) + d.z; d.w = yl_mul_rn(sm.w, inv) + d.w;
        *o = d;
      }
    }
  };
  load_idx(eS);
  gather(eS);
  load_idx(eS + 64);
  layer1(Hs);
  gather(eS + 64);
  load_idx(eS + 128);
  __syncthreads();                                    // barrier 0
  for (int p = 0; p < np; ++p) {
    layer1(Hs + ((p + 1) & 1) * TILE);                // pass p+1 (gathered during pass p-1 / the barrier)
    gather(eS + 64 * (p + 2));
    load_idx(eS + 64 * (p + 3));
    meta(p);
    if (p >= 1) aggregate(p - 1);                     // adds onto fo0 = rows loaded at the end of the last interval
    fo_load(p);
    __syncthreads();                                  // barrier p+1
  }
  aggregate(np - 1);
}

// nodes per node-tile workgroup: ~56 edges on average so that a single 64-edge pass is the common case
// (measured at cfg 5: 9 nodes / one pass 208 us, 12 nodes / a second mostly-empty pass 242 us, 16 nodes /
// two full passes 194 us — on big graphs two passes halve the per-workgroup W2 staging)
static long edge_tile_npt(long N, long E) {
  long npt = (56 * N) / E;
  const long npt2 = (112 * N) / E < 16 ? (112 * N) / E : 16;
  if (npt2 >= 2 * npt - 2 && N / (npt2 > 0 ? npt2 : 1) >= 8192) npt = npt2;
  if (npt < 1) npt = 1;
  if (npt > 64) npt = 64;
  return npt;
}
int yl_edge_tile_groups(int64_t N, int64_t E) {
  if (N <= 0 || E <= 0) return 0;
  const int variant = E >= 131072 ? (yl_strict_fp32() ? YOLAT_EDGE_WS_F32 : YOLAT_EDGE_WS_X6) : YOLAT_EDGE_TILES;
  if (variant != YOLAT_EDGE_TILES) return 0;
  return edge_tile_npt(N, E) <= 16 ? 1 : 4;
}

int yl_edge_uv_mlp2_mean_eval_impl(const float* UV, int64_t ld_uv, const int32_t* src_csr, const int32_t* dst_csr,
                                   const float* attr_csr, const int32_t* row_ptr, int64_t N, int64_t E, const float* Wc4,
                                   const float* b1, const float* s1, const float* t1, const float* W2, const float* b2,
                                   const float* s2, const float* t2, int64_t C, float* f_out, int64_t ld_fo, int variant,
                                   const PoolRider* rider, int* rode, const EdgeNext* next, int* did_next,
                                   yolat_stream_t stream) {
  if (rode) *rode = 0;
  if (did_next) *did_next = 0;
  if (E < 0 || N <= 0 || !UV || !Wc4 || !W2 || !row_ptr || !f_out) return YOLAT_E_INVALID;
  if (variant < YOLAT_EDGE_AUTO || variant > YOLAT_EDGE_WS_X6) return YOLAT_E_INVALID;
  if (C != 64) return YOLAT_E_UNSUPPORTED;
  if (E == 0) return 0;
  if (!src_csr || !dst_csr || !attr_csr || E >= (1LL << 31) || ld_fo < C || ld_uv < 2 * C) return YOLAT_E_INVALID;
  if ((s1 == nullptr) != (t1 == nullptr) || (s2 == nullptr) != (t2 == nullptr)) return YOLAT_E_INVALID;
  if (ld_uv % 4 != 0 || ld_fo % 4 != 0 || !yl_aligned16(UV) || !yl_aligned16(attr_csr) || !yl_aligned16(Wc4) ||
      (b1 && !yl_aligned16(b1)) || !yl_aligned16(f_out) || (s1 && (!yl_aligned16(s1) || !yl_aligned16(t1))))
    return YOLAT_E_UNSUPPORTED;
  // folded form: layer 1's bias / BatchNorm already applied to UV and Wc4 by the caller, layer 2's bias inside t2
  const bool fold = b1 == nullptr && s1 == nullptr && b2 == nullptr;
  const long ws_wgs = 512;            // persistent workgroups of the wave-specialised kernels: two per CU
  // the persistent kernel addresses UV / attr / the index arrays with 32-bit byte offsets
  const bool ws_ok = yl_aligned16(W2) && E >= 64 && N * ld_uv * 4 < (1LL << 32) && E * 16 < (1LL << 32);
  if (variant == YOLAT_EDGE_AUTO) {
    variant = E >= 131072 ? (yl_strict_fp32() ? YOLAT_EDGE_WS_F32 : YOLAT_EDGE_WS_X6) : YOLAT_EDGE_TILES;
    if (!ws_ok) variant = YOLAT_EDGE_TILES;
  } else if (variant != YOLAT_EDGE_TILES && !ws_ok) {
    return YOLAT_E_UNSUPPORTED;
  }
  if (variant != YOLAT_EDGE_TILES) {
    // 2 workgroups of 512 threads per CU, each walking a contiguous edge range in full 64-edge passes
    long chunk = ((E + ws_wgs - 1) / ws_wgs + 63) / 64 * 64;
    if (chunk < 64) chunk = 64;
    const unsigned grid = (unsigned)((E + chunk - 1) / chunk);
#define YL_WS_LAUNCH(F, X)                                                                                          \
  hipLaunchKernelGGL((k_edge_uv_mlp2_mean_ws<F, X>), dim3(grid), dim3(512), 0, (hipStream_t)stream, UV, (long)ld_uv,   \
                     src_csr, dst_csr, attr_csr, row_ptr, (int)N, (int)E, (int)chunk, Wc4, b1, s1, t1, W2, b2, s2, t2, \
                     f_out, (long)ld_fo)
    if (variant == YOLAT_EDGE_WS_X6) { if (fold) YL_WS_LAUNCH(true, true); else YL_WS_LAUNCH(false, true); }
    else { if (fold) YL_WS_LAUNCH(true, false); else YL_WS_LAUNCH(false, false); }
#undef YL_WS_LAUNCH
    YL_LAUNCH_CHECK();
    return 0;
  }
  const long npt = edge_tile_npt(N, E);
  DenseOp w2 = yl_dense(W2, C, C, C);
  const int tiles = yl_cdiv(N, npt);
  PoolRider pr{};
  if (rider && rider->blocks > 0) { pr = *rider; if (rode) *rode = 1; }
  EdgeNext nx{};
  if (next && next->Wp && npt <= 16) {
    if (!next->bias || !next->UV || !next->root || next->UV == UV) return YOLAT_E_INVALID;
    if (next->s_tiles > 0 && (!next->s_in || !next->Wn || !next->sn || !next->tn || !next->s_out || next->ld_si % 4 != 0 ||
                              !yl_aligned16(next->s_in) || !yl_aligned16(next->Wn)))
      return YOLAT_E_INVALID;
    nx = *next;
    if (did_next) *did_next = 1;
  }
  const unsigned grid = (unsigned)tiles + (unsigned)nx.s_tiles + (unsigned)pr.blocks;
  if (npt <= 16)
    hipLaunchKernelGGL(k_edge_uv_mlp2_mean<1>, dim3(grid), dim3(256), 0, (hipStream_t)stream, UV, (long)ld_uv,
                       src_csr, dst_csr, attr_csr, row_ptr, (int)N, (int)npt, Wc4, b1, s1, t1, w2, b2, s2, t2, f_out,
                       (long)ld_fo, (int)E, tiles, pr, nx);
  else
    hipLaunchKernelGGL(k_edge_uv_mlp2_mean<4>, dim3(grid), dim3(256), 0, (hipStream_t)stream, UV, (long)ld_uv,
                       src_csr, dst_csr, attr_csr, row_ptr, (int)N, (int)npt, Wc4, b1, s1, t1, w2, b2, s2, t2, f_out,
                       (long)ld_fo, (int)E, tiles, pr, nx);
  YL_LAUNCH_CHECK();
  return 0;
}

extern "C" int yolat_edge_uv_mlp2_mean_eval_variant(const float* UV, int64_t ld_uv, const int32_t* src_csr,
                                                    const int32_t* dst_csr, const float* attr_csr,
                                                    const int32_t* row_ptr, int64_t N, int64_t E, const float* Wc4,
                                                    const float* b1, const float* s1, const float* t1, const float* W2,
                                                    const float* b2, const float* s2, const float* t2, int64_t C,
                                                    float* f_out, int64_t ld_fo, int variant, yolat_stream_t stream) {
  return yl_edge_uv_mlp2_mean_eval_impl(UV, ld_uv, src_csr, dst_csr, attr_csr, row_ptr, N, E, Wc4, b1, s1, t1, W2, b2, s2,
                                        t2, C, f_out, ld_fo, variant, nullptr, nullptr, nullptr, nullptr, stream);
}

extern "C" int yolat_edge_uv_mlp2_mean_eval(const float* UV, int64_t ld_uv, const int32_t* src_csr,
                                            const int32_t* dst_csr, const float* attr_csr,
                                            const int32_t* row_ptr, int64_t N, int64_t E, const float* Wc4,
                                            const float* b1, const float* s1, const float* t1, const float* W2,
                                            const float* b2, const float* s2, const float* t2, int64_t C,
                                            float* f_out, int64_t ld_fo, yolat_stream_t stream) {
  return yolat_edge_uv_mlp2_mean_eval_variant(UV, ld_uv, src_csr, dst_csr, attr_csr, row_ptr, N, E, Wc4, b1, s1, t1, W2,
                                              b2, s2, t2, C, f_out, ld_fo, YOLAT_EDGE_AUTO, stream);
}

// ------------------------------------------------------------------------------------------------
// Factorised FIRST edge Linear for the training forward:  H1[q] = U[dst_q] + V[src_q] + W1c.attr_q + b1
// (UV = x.[W1a-W1b | W1b]^T computed once per node by a dense GEMM), plus the BatchNorm partial statistics in the
// GEMM epilogue's format (float2 (sum, M2 about the group mean) per 32-row group and column).  Replaces the
// gathered K = 2 Cin + 4 GEMM when E >> N: at E = 1.2 M / N = 200 k that GEMM ran at 17 TFLOP/s (1.19 ms) because
// its A operand is two random 256-B row gathers per edge.  One workgroup = 64 edges: thread (row rb + 16 t,
// columns 4q..) writes its 16 bytes of H1 and parks them in LDS; 128 threads then reduce the two 32-row groups.
// ------------------------------------------------------------------------------------------------
template <class TO>
__global__ void __launch_bounds__(256) k_edge_uv_lin1(const float* __restrict__ UV, long ld_uv,
                                                      const int* __restrict__ src, const int* __restrict__ dst,
                                                      const float* __restrict__ attr, int E,
                                                      const float* __restrict__ Wc4, const float* __restrict__ b1,
                                                      TO* __restrict__ H1, long ldh, float2* __restrict__ stats) {
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
    if (row0 + r < E) yl_st4(H1 + (long)(row0 + r) * ldh + 4 * q, h);
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
    if (cnt == 32) {
      // full group: the 32 values in registers (reads in flight together; the rolled loops below were 2 x 32 dependent
      // LDS round trips per workgroup), same order of additions
      float v[32];
#pragma unroll
      for (int r = 0; r < 32; ++r) v[r] = T[(32 * grp + r) * LDT + c];
      float sum = 0.f;
#pragma unroll
      for (int r = 0; r < 32; ++r) sum += v[r];
      const float mu = sum / 32.f;
      float m2 = 0.f;
#pragma unroll
      for (int r = 0; r < 32; ++r) { const float d = v[r] - mu; m2 += d * d; }
      stats[(long)(base >> 5) * 64 + c] = make_float2(sum, m2);
    } else if (cnt > 0) {
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
  hipLaunchKernelGGL(k_edge_uv_lin1<float>, dim3(yl_cdiv(E, 64)), dim3(256), 0, (hipStream_t)stream, UV, (long)ld_uv, src_csr,
                     dst_csr, attr_csr, (int)E, Wc4, b1, H1, (long)ldh, reinterpret_cast<float2*>(stats));
  YL_LAUNCH_CHECK();
  return 0;
}

// The same with H1 stored as bfloat16 (round-to-nearest-even); the BatchNorm partial statistics are those of the
// fp32 values before rounding.
extern "C" int yolat_edge_uv_lin1_fwd_h(const float* UV, int64_t ld_uv, const int32_t* src_csr, const int32_t* dst_csr,
                                        const float* attr_csr, int64_t E, const float* Wc4, const float* b1, int64_t C,
                                        uint16_t* H1, int64_t ldh, float* stats, yolat_stream_t stream) {
  if (E < 0 || !UV || !Wc4) return YOLAT_E_INVALID;
  if (C != 64) return YOLAT_E_UNSUPPORTED;
  if (E == 0) return 0;
  if (!src_csr || !dst_csr || !attr_csr || !H1 || E >= (1LL << 31) || ldh < C || ld_uv < 2 * C) return YOLAT_E_INVALID;
  if (ld_uv % 4 != 0 || ldh % 4 != 0 || !yl_aligned16(UV) || !yl_aligned16(attr_csr) || !yl_aligned16(Wc4) ||
      (((uintptr_t)H1) & 7) != 0 || (b1 && !yl_aligned16(b1)) || (stats && (((uintptr_t)stats) & 7) != 0))
    return YOLAT_E_UNSUPPORTED;
  hipLaunchKernelGGL(k_edge_uv_lin1<yl_bf16_t>, dim3(yl_cdiv(E, 64)), dim3(256), 0, (hipStream_t)stream, UV, (long)ld_uv,
                     src_csr, dst_csr, attr_csr, (int)E, Wc4, b1, H1, (long)ldh, reinterpret_cast<float2*>(stats));
  YL_LAUNCH_CHECK();
  return 0;
}

extern "C" int yolat_edge_uv_mlp2_eval(const float* UV, int64_t ld_uv, const int32_t* src_csr,
                                       const int32_t* dst_csr, const float* attr_csr, int64_t E, const float* Wc4,
                                       const float* b1, const float* s1, const float* t1, const float* W2,
                                       const float* b2, const float* s2, const float* t2, int64_t C, float* H2,
                                       int64_t ldh, yolat_stream_t stream) {
  if (E < 0 || !UV || !Wc4 || !W2) return YOLAT_E_INVALID;
  if (C != 64) return YOLAT_E_UNSUPPORTED;
  if (E == 0) return 0;
  if (!src_csr || !dst_csr || !attr_csr || !H2 || E >= (1LL << 31) || ldh < C || ld_uv < 2 * C) return YOLAT_E_INVALID;
  if ((s1 == nullptr) != (t1 == nullptr) || (s2 == nullptr) != (t2 == nullptr)) return YOLAT_E_INVALID;
  if (ld_uv % 4 != 0 || !yl_aligned16(UV) || !yl_aligned16(attr_csr) || !yl_aligned16(Wc4) || (b1 && !yl_aligned16(b1)) ||
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
  yl_reduce_dw_db(st, partial, elems, p.S, dW1, (long)lddw, (int)K, dbpart, db1, (long)C, accumulate);
  YL_LAUNCH_CHECK();
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
template <int LPN, class T = float>
__global__ void __launch_bounds__(256) k_csr_mean_fwd_v4(const T* __restrict__ H, long ldh,
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
  const T* hp = H + 4 * sub;
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
    for (int j = 0; j < 8; ++j) v[j] = yl_ld4(hp + (long)(q + j) * ldh);
#pragma unroll
    for (int j = 0; j < 8; ++j) add(v[j]);
  }
  if (q < q1) {   // 1..7 remaining rows: clamped (re-read) addresses keep the loads unconditional
    float4 v[7];
#pragma unroll
    for (int j = 0; j < 7; ++j) v[j] = yl_ld4(hp + (long)yl_min(q + j, q1 - 1) * ldh);
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

// bfloat16-stored message matrix (training with bf16 storage): C = 64, 8-byte aligned rows
extern "C" int yolat_csr_mean_fwd_h(const uint16_t* H, int64_t ldh, int64_t C, const float* h_scale,
                                    const float* h_shift, int h_relu, const int32_t* row_ptr, int64_t N, float* out,
                                    int64_t ldo, int accumulate, yolat_stream_t stream) {
  if (N <= 0 || !H || !row_ptr || !out || ldo < C || ldh < C) return YOLAT_E_INVALID;
  if ((h_scale == nullptr) != (h_shift == nullptr)) return YOLAT_E_INVALID;
  if (C != 64 || ldh % 4 != 0 || ldo % 4 != 0 || (((uintptr_t)H) & 7) != 0 || !yl_aligned16(out) ||
      (h_scale && (!yl_aligned16(h_scale) || !yl_aligned16(h_shift))))
    return YOLAT_E_UNSUPPORTED;
  hipLaunchKernelGGL((k_csr_mean_fwd_v4<16, yl_bf16_t>), dim3(yl_cdiv(N, 16)), dim3(256), 0, (hipStream_t)stream, H,
                     (long)ldh, h_scale, h_shift, h_relu, row_ptr, (int)N, out, (long)ldo, accumulate);
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
template <class T>
__global__ void __launch_bounds__(256) k_csr_mean_bwd_v4(const float* __restrict__ dOut, long lddo,
                                                         const int* __restrict__ row_ptr, const int* __restrict__ dst,
                                                         int E, T* __restrict__ dM, long lddm) {
  const int l16 = threadIdx.x & 15;
  const int q = blockIdx.x * 16 + (threadIdx.x >> 4);
  if (q >= E) return;
  const int n = dst[q];
  const int deg = row_ptr[n + 1] - row_ptr[n];
  const float inv = 1.f / (float)(deg > 1 ? deg : 1);
  const float4 g = *reinterpret_cast<const float4*>(dOut + (long)n * lddo + 4 * l16);
  yl_st4(dM + (long)q * lddm + 4 * l16, make_float4(g.x * inv, g.y * inv, g.z * inv, g.w * inv));
}

extern "C" int yolat_csr_mean_bwd(const float* dOut, int64_t lddo, int64_t C,
                                  const int32_t* row_ptr, const int32_t* dst_csr, int64_t E,
                                  float* dM, int64_t lddm, yolat_stream_t stream) {
  if (E < 0 || C <= 0 || !dOut || !row_ptr) return YOLAT_E_INVALID;
  if (E == 0) return 0;
  if (!dst_csr || !dM || lddm < C) return YOLAT_E_INVALID;
  if (C == 64 && lddo % 4 == 0 && lddm % 4 == 0 && yl_aligned16(dOut) && yl_aligned16(dM))
    hipLaunchKernelGGL(k_csr_mean_bwd_v4<float>, dim3(yl_cdiv(E, 16)), dim3(256), 0, (hipStream_t)stream, dOut, (long)lddo,
                       row_ptr, dst_csr, (int)E, dM, (long)lddm);
  else
  hipLaunchKernelGGL(k_csr_mean_bwd, dim3(yl_cdiv(E, 4)), dim3(256), 0, (hipStream_t)stream, dOut,
                     (long)lddo, (int)C, row_ptr, dst_csr, (int)E, dM, (long)lddm);
  YL_LAUNCH_CHECK();
  return 0;
}

extern "C" int yolat_csr_mean_bwd_h(const float* dOut, int64_t lddo, int64_t C, const int32_t* row_ptr,
                                    const int32_t* dst_csr, int64_t E, uint16_t* dM, int64_t lddm,
                                    yolat_stream_t stream) {
  if (E < 0 || !dOut || !row_ptr) return YOLAT_E_INVALID;
  if (E == 0) return 0;
  if (!dst_csr || !dM || lddm < C) return YOLAT_E_INVALID;
  if (C != 64 || lddo % 4 != 0 || lddm % 4 != 0 || !yl_aligned16(dOut) || (((uintptr_t)dM) & 7) != 0)
    return YOLAT_E_UNSUPPORTED;
  hipLaunchKernelGGL(k_csr_mean_bwd_v4<yl_bf16_t>, dim3(yl_cdiv(E, 16)), dim3(256), 0, (hipStream_t)stream, dOut,
                     (long)lddo, row_ptr, dst_csr, (int)E, dM, (long)lddm);
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

// ------------------------------------------------------------------------------------------------
// Backward of the FACTORISED first edge Linear (training, E >> N).  Forward:  H1[q] = U[dst_q] + V[src_q] + W1c.attr_q + b1
// with UV = x.[W1a - W1b | W1b]^T, so the gradients w.r.t. the per-node products are two per-node sums of dH1 rows,
//   dU[n] = sum_{q in CSR row n} dH1[q]   (contiguous rows)      dV[n] = sum_{t in CSC col n} dH1[slots[t]]   (gathered)
// and everything else is N-row dense algebra (dWuv = dUV^T.x, dx += dUV.Wuv) plus dW1c = dH1^T.attr, db1 = colsum(dH1).
// Replaces the gathered K = 2Cin+4 TN GEMM (dW1), the E x 64 -> E x 2Cin NT GEMM (dG) and the gather-scatter of dG:
// at cfg 5 (E = 1.2 M, N = 200 k) 595 + 347 + 145 us per block layer.  One 16-lane group per node, one float4 of
// columns per lane, 8 rows in flight; ascending slot order -> deterministic.  C = 64.
// ------------------------------------------------------------------------------------------------
template <class T>
__global__ void __launch_bounds__(256) k_edge_uv_sums(const T* __restrict__ dH, long ldh,
                                                      const int* __restrict__ row_ptr, const int* __restrict__ col_ptr,
                                                      const int* __restrict__ slots, int N, float* __restrict__ dUV,
                                                      long ldo, int with_u) {
  const int sub = threadIdx.x & 15;
  const int n = blockIdx.x * 16 + (threadIdx.x >> 4);
  if (n >= N) return;
  const T* hp = dH + 4 * sub;
  auto acc = [](float4& s, const float4& v) { s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w; };
  // ---- dU: the node's own CSR rows (with_u == 0: written by yolat_bn_apply_edge_sums)
  int q0 = 0, q1 = 0;
  if (with_u) { q0 = row_ptr[n]; q1 = row_ptr[n + 1]; }
  float4 su = make_float4(0.f, 0.f, 0.f, 0.f);
  int q = q0;
  for (; q + 8 <= q1; q += 8) {
    float4 v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = yl_ld4(hp + (long)(q + j) * ldh);
#pragma unroll
    for (int j = 0; j < 8; ++j) acc(su, v[j]);
  }
  if (q < q1) {
    float4 v[7];
#pragma unroll
    for (int j = 0; j < 7; ++j) v[j] = yl_ld4(hp + (long)yl_min(q + j, q1 - 1) * ldh);
#pragma unroll
    for (int j = 0; j < 7; ++j)
      if (q + j < q1) acc(su, v[j]);
  }
  // ---- dV: rows of the edges that leave this node (CSC by source; slot list ascending)
  const int t0 = col_ptr[n], t1 = col_ptr[n + 1];
  float4 sv = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int t = t0; t < t1; t += 8) {
    int sl[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) sl[j] = slots[yl_min(t + j, t1 - 1)];
    float4 v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = yl_ld4(hp + (long)sl[j] * ldh);
#pragma unroll
    for (int j = 0; j < 8; ++j)
      if (t + j < t1) acc(sv, v[j]);
  }
  float* o = dUV + (long)n * ldo + 4 * sub;
  if (with_u) *reinterpret_cast<float4*>(o) = su;
  *reinterpret_cast<float4*>(o + 64) = sv;
}

extern "C" int yolat_edge_uv_sums(const float* dH1, int64_t ldh, const int32_t* row_ptr, const int32_t* col_ptr,
                                  const int32_t* slots, int64_t N, int64_t C, float* dUV, int64_t ld_uv,
                                  yolat_stream_t stream) {
  if (N <= 0 || !dH1 || !row_ptr || !col_ptr || !slots || !dUV || ldh < C || ld_uv < 2 * C) return YOLAT_E_INVALID;
  if (C != 64 || ldh % 4 != 0 || ld_uv % 4 != 0 || !yl_aligned16(dH1) || !yl_aligned16(dUV)) return YOLAT_E_UNSUPPORTED;
  hipLaunchKernelGGL(k_edge_uv_sums<float>, dim3(yl_cdiv(N, 16)), dim3(256), 0, (hipStream_t)stream, dH1, (long)ldh, row_ptr,
                     col_ptr, slots, (int)N, dUV, (long)ld_uv, 1);
  YL_LAUNCH_CHECK();
  return 0;
}

// The dV half alone (dUV columns [64, 128): sums over the CSC column of every node, gathered by slot), for callers that
// got the dU half from yolat_bn_apply_edge_sums.  dH1 fp32 (half = 0) or bfloat16.
extern "C" int yolat_edge_uv_sums_v(const void* dH1, int64_t ldh, int half, const int32_t* col_ptr, const int32_t* slots,
                                    int64_t N, int64_t C, float* dUV, int64_t ld_uv, yolat_stream_t stream) {
  if (N <= 0 || !dH1 || !col_ptr || !slots || !dUV || ldh < C || ld_uv < 2 * C) return YOLAT_E_INVALID;
  if (C != 64 || ldh % 4 != 0 || ld_uv % 4 != 0 || (((uintptr_t)dH1) & (half ? 7 : 15)) != 0 || !yl_aligned16(dUV))
    return YOLAT_E_UNSUPPORTED;
  if (half)
    hipLaunchKernelGGL(k_edge_uv_sums<yl_bf16_t>, dim3(yl_cdiv(N, 16)), dim3(256), 0, (hipStream_t)stream,
                       reinterpret_cast<const yl_bf16_t*>(dH1), (long)ldh, col_ptr, col_ptr, slots, (int)N, dUV, (long)ld_uv, 0);
  else
    hipLaunchKernelGGL(k_edge_uv_sums<float>, dim3(yl_cdiv(N, 16)), dim3(256), 0, (hipStream_t)stream,
                       reinterpret_cast<const float*>(dH1), (long)ldh, col_ptr, col_ptr, slots, (int)N, dUV, (long)ld_uv, 0);
  YL_LAUNCH_CHECK();
  return 0;
}

extern "C" int yolat_edge_uv_sums_h(const uint16_t* dH1, int64_t ldh, const int32_t* row_ptr, const int32_t* col_ptr,
                                    const int32_t* slots, int64_t N, int64_t C, float* dUV, int64_t ld_uv,
                                    yolat_stream_t stream) {
  if (N <= 0 || !dH1 || !row_ptr || !col_ptr || !slots || !dUV || ldh < C || ld_uv < 2 * C) return YOLAT_E_INVALID;
  if (C != 64 || ldh % 4 != 0 || ld_uv % 4 != 0 || (((uintptr_t)dH1) & 7) != 0 || !yl_aligned16(dUV))
    return YOLAT_E_UNSUPPORTED;
  hipLaunchKernelGGL(k_edge_uv_sums<yl_bf16_t>, dim3(yl_cdiv(N, 16)), dim3(256), 0, (hipStream_t)stream, dH1, (long)ldh,
                     row_ptr, col_ptr, slots, (int)N, dUV, (long)ld_uv, 1);
  YL_LAUNCH_CHECK();
  return 0;
}

// ------------------------------------------------------------------------------------------------
// Weight gradient of the attr columns of the factorised first edge Linear + its bias gradient (training backward,
// torch_vertex.py:331 nn.0 restricted to the 4 edge-attribute inputs):
//   dWc4[c][j] = sum_e dH1[e][c] * attr[e][j]      (64 x 4)        db1[c] = sum_e dH1[e][c]
// A pure streaming reduction over the [E, 64] gradient (307 MB fp32 / 154 MB bf16 at E = 1.2 M).  It used to run on the
// general TN GEMM (k_gemm_tn split-row: 64 x 64 tiles for a [64 x 4] result): 279 us fp32 / 319 us bf16 per call = 0.15 /
// 0.13 of HBM, 14 - 18 % of the cfg-5 train step.  Here: one pass, 16 bytes (fp32) / 8 bytes (bf16) per
// lane, a 16-lane group per row (the row's attr quad is one broadcast 16-byte load), 64 rows per workgroup and step with
// all loads of the step in flight, 16 + 4 register accumulators per lane; the 16 row groups of a workgroup are summed
// through LDS in fixed order, the workgroups' partials by a second small launch in fixed order: deterministic.
// ------------------------------------------------------------------------------------------------
constexpr int ADW_WG_MAX = 2048;
template <class T>
__global__ void __launch_bounds__(256) k_attr_dw(const T* __restrict__ dH, long ldh, const float4* __restrict__ attr, int E,
                                                 int rows_wg, float* __restrict__ part) {
  __shared__ float red[16][16 * 20 + 4];
  const int tid = threadIdx.x, sub = tid & 15, rg = tid >> 4;
  const int b0 = blockIdx.x * rows_wg, b1 = yl_min(b0 + rows_wg, E);
  float w[4][4], sb[4];
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    sb[c] = 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j) w[c][j] = 0.f;
  }
  const T* hp = dH + 4 * sub;
  for (int r0 = b0; r0 < b1; r0 += 64) {
    float4 h[4], a[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const int r = yl_min(r0 + 16 * t + rg, E - 1);
      h[t] = yl_ld4(hp + (long)r * ldh);
      a[t] = attr[r];
    }
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      if (r0 + 16 * t + rg < b1) {
        const float hv[4] = {h[t].x, h[t].y, h[t].z, h[t].w}, av[4] = {a[t].x, a[t].y, a[t].z, a[t].w};
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          sb[c] += hv[c];
#pragma unroll
          for (int j = 0; j < 4; ++j) w[c][j] = fmaf(hv[c], av[j], w[c][j]);
        }
      }
    }
  }
  // this lane's 20 sums: columns 4 sub + c -> [c][j] at (4 sub + c) * 4 + j, bias at 256 + 4 sub + c
  float* mine = red[rg];
#pragma unroll
  for (int c = 0; c < 4; ++c) {
#pragma unroll
    for (int j = 0; j < 4; ++j) mine[(4 * sub + c) * 4 + j] = w[c][j];
    mine[256 + 4 * sub + c] = sb[c];
  }
  __syncthreads();
  for (int i = tid; i < 320; i += 256) {
    float s = red[0][i];
#pragma unroll
    for (int g2 = 1; g2 < 16; ++g2) s += red[g2][i];
    part[(long)blockIdx.x * 320 + i] = s;
  }
}
// Sum of the workgroups' partials: ONE 256-thread workgroup per output (320 of them), thread t takes partial rows t, t + 256,
// ... (eight loads in flight), the 256 thread sums meet in a fixed tree (shuffles inside a wave, then the four wave sums
// in order) -> deterministic.  (32 lanes per output in 40 workgroups walked 64 partials per lane: 15 us per call for 2.6 MB,
// four calls per cfg-5 step; one workgroup walking all 2048 partials per output was 57 us.)
static __global__ void __launch_bounds__(256) k_attr_dw_reduce(const float* __restrict__ part, int nwg, float* __restrict__ dWc4,
                                                             float* __restrict__ db) {
  __shared__ float ws[4];
  const int o = blockIdx.x, t = threadIdx.x;                     // o < 320 by the launch
  float s = 0.f;
  for (int g0 = t; g0 < nwg; g0 += 8 * 256) {
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = part[(long)yl_min(g0 + 256 * j, nwg - 1) * 320 + o];
#pragma unroll
    for (int j = 0; j < 8; ++j)
      if (g0 + 256 * j < nwg) s += v[j];
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off);
  if ((t & 63) == 0) ws[t >> 6] = s;
  __syncthreads();
  if (t == 0) {
    s = ((ws[0] + ws[1]) + ws[2]) + ws[3];
    if (o < 256) dWc4[o] = s;
    else if (db != nullptr) db[o - 256] = s;
  }
}

extern "C" size_t yolat_edge_attr_dw_work_elems(int64_t E) { (void)E; return (size_t)ADW_WG_MAX * 320; }

extern "C" int yolat_edge_attr_dw(const void* dH1, int64_t ldh, int half, const float* attr_csr, int64_t E, int64_t C,
                                  float* dWc4, float* db1, float* work, yolat_stream_t stream) {
  if (E <= 0 || !dH1 || !attr_csr || !dWc4 || !work || ldh < C) return YOLAT_E_INVALID;
  if (C != 64 || ldh % 4 != 0 || !yl_aligned16(attr_csr) || E >= (1LL << 31) - 64 ||
      (((uintptr_t)dH1) & (half ? 7 : 15)) != 0)
    return YOLAT_E_UNSUPPORTED;
  hipStream_t st = (hipStream_t)stream;
  long rows_wg = (yl_cdiv(E, ADW_WG_MAX) + 63) / 64 * 64;
  if (rows_wg < 256) rows_wg = 256;
  const int nwg = yl_cdiv(E, rows_wg);
  if (half)
    hipLaunchKernelGGL(k_attr_dw<yl_bf16_t>, dim3(nwg), dim3(256), 0, st, reinterpret_cast<const yl_bf16_t*>(dH1), (long)ldh,
                       reinterpret_cast<const float4*>(attr_csr), (int)E, (int)rows_wg, work);
  else
    hipLaunchKernelGGL(k_attr_dw<float>, dim3(nwg), dim3(256), 0, st, reinterpret_cast<const float*>(dH1), (long)ldh,
                       reinterpret_cast<const float4*>(attr_csr), (int)E, (int)rows_wg, work);
  YL_LAUNCH_CHECK();
  hipLaunchKernelGGL(k_attr_dw_reduce, dim3(320), dim3(256), 0, st, work, nwg, dWc4, db1);
  YL_LAUNCH_CHECK();
  return 0;
}

// ------------------------------------------------------------------------------------------------
// BatchNorm-1 backward apply + the CONTIGUOUS consumers of its result in one pass (round 4; training backward of the
// factorised conv layer, torch_vertex.py:331-332 nn.1/nn.2 backward feeding nn.0's):
//   dH1[q] = scale * (relu'(.) dA1[q] - c1 - xhat[q] c2)                        (k_bn_bwd_apply_v4's arithmetic, stored)
//   dU[n]  = sum_{q in CSR row n} dH1[q]                                        (k_edge_uv_sums' first half, ascending q)
//   dWc4   = dH1^T . attr,  db1 = column sums of dH1                            (k_attr_dw)
// dH1 used to be written by the apply pass and read three times (dU rows, dV gather, attr gradient): 921 + 614 + 307 MB
// per layer at E = 1.2 M.  Here the rows of a node are formed, stored (the dV gather of yolat_edge_uv_sums_v still
// needs them) and summed while they are in registers: 921 MB + the gather.  One 16-lane group per node (a float4 of
// columns per lane), 4 rows in flight, a workgroup owns a contiguous range of nodes = a contiguous range of rows;
// 16 + 4 attr accumulators per lane over all the nodes of the group, reduced like k_attr_dw (LDS in fixed order, then
// k_attr_dw_reduce over the workgroups): deterministic.  bf16 storage: the sums take the ROUNDED values, i.e. what the
// gather reads back.
// ------------------------------------------------------------------------------------------------
constexpr int BA_NODES_MAX = 1024;
__device__ __forceinline__ float ba_round(float v, float) { return v; }
__device__ __forceinline__ float ba_round(float v, yl_bf16_t) { return __uint_as_float((yl_pack_bf16(v, 0.f) & 0xffffu) << 16); }
__device__ __forceinline__ float ba_ld1(const float* p) { return *p; }
__device__ __forceinline__ float ba_ld1(const yl_bf16_t* p) { return __uint_as_float((unsigned)(*p) << 16); }
__device__ __forceinline__ void ba_st1(float* p, float v) { *p = v; }
__device__ __forceinline__ void ba_st1(yl_bf16_t* p, float v) { *p = (yl_bf16_t)(__float_as_uint(v) >> 16); }   // v is rounded already
// Mapping: one WAVE per row, lane = column.  (The first version used a 16-lane group per node with a float4 of columns
// per lane, like k_edge_uv_sums: 16 attr accumulators + 24 constants + the rows in flight = 124 VGPRs = 4 waves per SIMD,
// and it ran at 184 us per call at E = 1.2 M against 144 for the apply pass alone.)  With the row wave-uniform the
// attr quad is a scalar load, a lane holds 4 + 1 accumulators and 6 constants, eight rows are in flight with 16
// registers, and the node boundaries are scalar branches.
template <class T>
__global__ void __launch_bounds__(256) k_bn_apply_edge_sums(const T* __restrict__ dZ, long lddz, const T* __restrict__ Y,
                                                            long ldy, T* dY, long lddy,
                                                            const float* __restrict__ mean, const float* __restrict__ invstd,
                                                            const float* __restrict__ scale, const float* __restrict__ shift,
                                                            int relu, const float* __restrict__ coef,
                                                            const int* __restrict__ row_ptr, const float4* __restrict__ attr,
                                                            int N, int nodes_wg, float* __restrict__ dU, long ldo,
                                                            float* __restrict__ part) {
  __shared__ float red[4][320];
  __shared__ int rp[BA_NODES_MAX + 1];
  const int tid = threadIdx.x, c = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const float mu = mean[c], is = invstd[c], sc = scale[c], sh = shift[c], k1 = coef[c], k2 = coef[64 + c];
  auto one = [&](float y, float g) {
    if (relu && !(fmaf(y, sc, sh) > 0.f)) g = 0.f;
    // the product is a stored value: the sums below must add exactly what the gather reads back — the empty asm keeps
    // the compiler from contracting "sum + a * (...)" into an fma of its factors
    float p = sc * (g - k1 - ((y - mu) * is) * k2);
    asm volatile("" : "+v"(p));
    return ba_round(p, T());
  };
  float w[4] = {0.f, 0.f, 0.f, 0.f}, sb = 0.f;
  // The workgroup's nodes [n0, n1) are dealt to its 4 waves as 4 contiguous node ranges of (nearly) equal ROW counts:
  // wave k owns the nodes whose first row lies in the k-th quarter of the workgroup's row range (the last wave also the
  // trailing nodes without rows).  A function of the graph alone, so the order of every sum is fixed.
  const int n0 = blockIdx.x * nodes_wg, n1 = yl_min(n0 + nodes_wg, N);
  const int cnt = n1 - n0;                                  // <= BA_NODES_MAX by the launch
  for (int i = tid; i <= cnt; i += 256) rp[i] = row_ptr[n0 + i];
  __syncthreads();
  const long Q0 = rp[0], QR = (long)rp[cnt] - Q0;
  auto first_at_or_after = [&](long v) {                    // lowest i in [0, cnt] with rp[i] >= v
    int lo = 0, hi = cnt;
    while (lo < hi) {
      const int mid = (lo + hi) >> 1;
      if (rp[mid] >= v) hi = mid; else lo = mid + 1;
    }
    return lo;
  };
  int i = __builtin_amdgcn_readfirstlane(first_at_or_after(Q0 + QR * wave / 4));
  const int iend = wave == 3 ? cnt : __builtin_amdgcn_readfirstlane(first_at_or_after(Q0 + QR * (wave + 1) / 4));
  // the wave's nodes [i, iend) own the contiguous rows [rp[i], rp[iend]): streamed eight at a time whatever the node
  // boundaries are; a row that starts a new node first flushes the finished nodes' sums (nodes without rows: zeros)
  if (i < iend) {
    const int qa = __builtin_amdgcn_readfirstlane(rp[i]), qb = __builtin_amdgcn_readfirstlane(rp[iend]);
    int qn = __builtin_amdgcn_readfirstlane(rp[i + 1]);     // first row that is NOT node i's
    float su = 0.f;
    for (int q = qa; q < qb; q += 8) {
      float y[8], g[8];
      float4 av[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const long r = yl_min(q + k, qb - 1);               // wave-uniform
        y[k] = ba_ld1(Y + r * ldy + c);
        g[k] = ba_ld1(dZ + r * lddz + c);
        av[k] = attr[r];                                    // scalar load
      }
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        if (q + k < qb) {
          while (q + k >= qn) {
            dU[(long)(n0 + i) * ldo + c] = su;
            su = 0.f;
            ++i;
            qn = __builtin_amdgcn_readfirstlane(rp[i + 1]);
          }
          const float hv = one(y[k], g[k]);
          ba_st1(dY + (long)(q + k) * lddy + c, hv);
          su += hv;
          sb += hv;
          w[0] = fmaf(hv, av[k].x, w[0]); w[1] = fmaf(hv, av[k].y, w[1]);
          w[2] = fmaf(hv, av[k].z, w[2]); w[3] = fmaf(hv, av[k].w, w[3]);
        }
      }
    }
    for (; i < iend; ++i) {                                 // the last node with rows + trailing nodes without
      dU[(long)(n0 + i) * ldo + c] = su;
      su = 0.f;
    }
  }
  float* mine = red[wave];
#pragma unroll
  for (int j = 0; j < 4; ++j) mine[c * 4 + j] = w[j];
  mine[256 + c] = sb;
  __syncthreads();
  for (int e = tid; e < 320; e += 256) part[(long)blockIdx.x * 320 + e] = ((red[0][e] + red[1][e]) + red[2][e]) + red[3][e];
}

static long ba_nodes_wg(int64_t N) {
  // ~ADW_WG_MAX workgroups: 124 VGPRs = 4 workgroups per CU = 1024 resident on 256 CUs, so 2048 is two full rounds (the
  // first version rounded the node count up to a multiple of 16: 1786 workgroups at cfg 5 = 1.74 rounds)
  long nodes_wg = yl_cdiv(N, ADW_WG_MAX);
  if (nodes_wg < 16) nodes_wg = 16;
  if (nodes_wg > BA_NODES_MAX) nodes_wg = BA_NODES_MAX;      // N > 2 M nodes: more than ADW_WG_MAX workgroups
  return nodes_wg;
}
extern "C" size_t yolat_bn_apply_edge_sums_work_elems(int64_t N) {
  return (size_t)(N > 0 ? yl_cdiv(N, ba_nodes_wg(N)) : 1) * 320;
}

// dA1 / H1 / dH1 [E, 64] fp32 (half = 0) or bfloat16 (half != 0), rows in CSR order (dH1 may alias dA1); coef [128] =
// (c1 | c2) of the BatchNorm backward (yolat_bn_csr_l2_bwd's next_coef); dUV [N, ld_uv]: columns [0, 64) are written
// (the dV half: yolat_edge_uv_sums_v); dWc4 [64, 4], db1 [64] (nullable).  work: yolat_bn_apply_edge_sums_work_elems(N).
extern "C" int yolat_bn_apply_edge_sums(const void* dA1, int64_t ldda, const void* H1, int64_t ldh, void* dH1, int64_t lddh,
                                        int half, int64_t E, const float* save_mean, const float* save_invstd,
                                        const float* scale, const float* shift, int relu, const float* coef,
                                        const int32_t* row_ptr, const float* attr_csr, int64_t N, float* dUV, int64_t ld_uv,
                                        float* dWc4, float* db1, float* work, yolat_stream_t stream) {
  if (E <= 0 || N <= 0 || !dA1 || !H1 || !dH1 || !save_mean || !save_invstd || !scale || !shift || !coef || !row_ptr ||
      !attr_csr || !dUV || !dWc4 || !work || ldda < 64 || ldh < 64 || lddh < 64 || ld_uv < 64)
    return YOLAT_E_INVALID;
  const uintptr_t al = half ? 7 : 15;
  if (ldda % 4 != 0 || ldh % 4 != 0 || lddh % 4 != 0 || ld_uv % 4 != 0 || E >= (1LL << 31) - 64 || N >= (1LL << 31) - 64 ||
      (((uintptr_t)dA1 | (uintptr_t)H1 | (uintptr_t)dH1) & al) || !yl_aligned16(save_mean) || !yl_aligned16(save_invstd) ||
      !yl_aligned16(scale) || !yl_aligned16(shift) || !yl_aligned16(coef) || !yl_aligned16(attr_csr) || !yl_aligned16(dUV))
    return YOLAT_E_UNSUPPORTED;
  hipStream_t st = (hipStream_t)stream;
  const long nodes_wg = ba_nodes_wg(N);
  const int nwg = yl_cdiv(N, nodes_wg);
  if (half)
    hipLaunchKernelGGL(k_bn_apply_edge_sums<yl_bf16_t>, dim3(nwg), dim3(256), 0, st, reinterpret_cast<const yl_bf16_t*>(dA1),
                       (long)ldda, reinterpret_cast<const yl_bf16_t*>(H1), (long)ldh, reinterpret_cast<yl_bf16_t*>(dH1),
                       (long)lddh, save_mean, save_invstd, scale, shift, relu, coef, row_ptr,
                       reinterpret_cast<const float4*>(attr_csr), (int)N, (int)nodes_wg, dUV, (long)ld_uv, work);
  else
    hipLaunchKernelGGL(k_bn_apply_edge_sums<float>, dim3(nwg), dim3(256), 0, st, reinterpret_cast<const float*>(dA1),
                       (long)ldda, reinterpret_cast<const float*>(H1), (long)ldh, reinterpret_cast<float*>(dH1), (long)lddh,
                       save_mean, save_invstd, scale, shift, relu, coef, row_ptr, reinterpret_cast<const float4*>(attr_csr),
                       (int)N, (int)nodes_wg, dUV, (long)ld_uv, work);
  YL_LAUNCH_CHECK();
  hipLaunchKernelGGL(k_attr_dw_reduce, dim3(320), dim3(256), 0, st, work, nwg, dWc4, db1);
  YL_LAUNCH_CHECK();
  return 0;
}

// dW1 [C, 2Cin+4] from the gradients of the split weights (inverse of yolat_conv_split_w1):
//   dW1[:, 0:Cin] = dWuv[0:C],  dW1[:, Cin:2Cin] = dWuv[C:2C] - dWuv[0:C],  dW1[:, 2Cin:] = dWc4
static __global__ void k_conv_merge_dw1(const float* __restrict__ dWuv, const float* __restrict__ dWc4, int Cin, int C,
                                        float* dW1, long ld, int accumulate) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < C * Cin) {
    const int c = i / Cin, k = i % Cin;
    const float a = dWuv[(long)c * Cin + k], b = dWuv[(long)(C + c) * Cin + k] - a;
    float* row = dW1 + (long)c * ld;
    row[k] = accumulate ? row[k] + a : a;
    row[Cin + k] = accumulate ? row[Cin + k] + b : b;
  }
  if (i < C * 4) {
    float* d = dW1 + (long)(i / 4) * ld + 2 * Cin + (i % 4);
    *d = accumulate ? *d + dWc4[i] : dWc4[i];
  }
}

extern "C" int yolat_conv_merge_dw1(const float* dWuv, const float* dWc4, int64_t Cin, int64_t C, float* dW1,
                                    int64_t lddw, int accumulate, yolat_stream_t stream) {
  if (!dWuv || !dWc4 || !dW1 || Cin <= 0 || C <= 0 || lddw < 2 * Cin + 4) return YOLAT_E_INVALID;
  const long n = C * (Cin > 4 ? Cin : 4);
  hipLaunchKernelGGL(k_conv_merge_dw1, dim3(yl_cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, dWuv, dWc4, (int)Cin,
                     (int)C, dW1, (long)lddw, accumulate);
  YL_LAUNCH_CHECK();
  return 0;
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
