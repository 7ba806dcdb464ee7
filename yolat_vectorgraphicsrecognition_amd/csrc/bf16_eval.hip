// bf16_eval.hip — SparseCADGCN.forward (eval) with bfloat16 STORAGE and fp32 accumulation: the precision mode
// BASELINE.json's large-graph configuration names (N = 200k / E = 1.2M / n_blocks = 4, "bf16").
//
// Same kernel sequence as forward_eval.hip (cad_recognition/architecture3cc_rpn_gp_iter2.py:44-71,106-137;
// gcn_lib/sparse/torch_vertex.py:319-337), with
//   * every [N,*] activation that crosses HBM stored as bf16: the per-node U|V products the edge kernel
//     gathers (256 -> 128 B per row), the layer outputs / concat slots, the node branch;
//   * every Linear with K >= 64 on v_mfma_f32_32x32x16_bf16 (16x the fp32 MFMA rate), weights converted to
//     bf16 once per weight version, accumulators / bias / folded BatchNorm / ReLU / mean / max in fp32;
//   * the first layer's K = Cin0 = 5 products, the e_attr term (W1c.attr), the per-proposal pooled matrix Z
//     and the classifier activations stay fp32 (small or precision-critical).
// Rounding to bf16 is round-to-nearest-even (v_cvt_pk_bf16_f32).  Parity target: <= 1e-2 of the logits'
// scale against the fp32 oracle (SURVEY.md §8c "bf16 variant"); the fp32 path keeps the 1e-4 bar.
#include "common.hpp"
#include <stdlib.h>
#include <atomic>

typedef unsigned short u16;
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

// ------------------------------------------------------------------------------------------------
// fp32 -> bf16 (weights, once per weight version)
// ------------------------------------------------------------------------------------------------
static __global__ void k_f32_to_bf16(const float* __restrict__ src, long n, u16* __restrict__ dst) {
  const long i = ((long)blockIdx.x * blockDim.x + threadIdx.x) * 2;
  if (i + 1 < n) *reinterpret_cast<unsigned*>(dst + i) = yl_pack_bf16(src[i], src[i + 1]);
  else if (i < n) dst[i] = (u16)(yl_pack_bf16(src[i], 0.f) & 0xFFFFu);
}

extern "C" int yolat_f32_to_bf16(const float* src, int64_t n, uint16_t* dst, yolat_stream_t stream) {
  if (n < 0 || (n > 0 && (!src || !dst))) return YOLAT_E_INVALID;
  if (n == 0) return 0;
  if ((((uintptr_t)dst) & 3) != 0) return YOLAT_E_UNSUPPORTED;
  hipLaunchKernelGGL(k_f32_to_bf16, dim3(yl_cdiv((n + 1) / 2, 256)), dim3(256), 0, (hipStream_t)stream, src, (long)n, dst);
  YL_LAUNCH_CHECK();
  return 0;
}

// ------------------------------------------------------------------------------------------------
// NT GEMM on bf16 MFMAs:  Y[M,N] = epi( A[M,K] . B[N,K]^T ),  K % 64 == 0.
// 256 threads = 2x2 waves, wave tile (32 TM) x (32 TN); K step 64 bf16 = 128 B per row.  LDS rows are padded
// to 144 B so that both the 16-byte staging writes (8 lanes = one row) and the 16-byte fragment reads
// (8 consecutive rows at one k offset) hit 8 distinct 16-byte bank groups.
//   MFMA operand layout: A/B lane l holds the 8 consecutive k = 16 ks + 8 (l>>5) .. +7 of row / column l&31;
//   C/D layout as for the fp32 MFMA, so wave_epilogue (common.hpp) is shared with the fp32 kernels.
// A operand: bf16 rows (HOp) or fp32 rows converted while staging (FOp); B operand: bf16 weight rows.
// Rows beyond M / N are clamped (their outputs are masked by the epilogue).
// ------------------------------------------------------------------------------------------------
struct HOp {
  const u16* p; long ld; int rows;
  static constexpr int NR = 1;
  __device__ __forceinline__ void load(int r, int k, u32x4* raw) const {
    raw[0] = *reinterpret_cast<const u32x4*>(p + (long)yl_min(r, rows - 1) * ld + k);
  }
  static __device__ __forceinline__ u32x4 pack(const u32x4* raw) { return raw[0]; }
};
struct FOp {
  const float* p; long ld; int rows;
  static constexpr int NR = 2;
  __device__ __forceinline__ void load(int r, int k, u32x4* raw) const {
    const float* q = p + (long)yl_min(r, rows - 1) * ld + k;
    raw[0] = *reinterpret_cast<const u32x4*>(q);
    raw[1] = *reinterpret_cast<const u32x4*>(q + 4);
  }
  static __device__ __forceinline__ u32x4 pack(const u32x4* raw) {
    u32x4 o;
    o.x = yl_pack_bf16(__uint_as_float(raw[0].x), __uint_as_float(raw[0].y));
    o.y = yl_pack_bf16(__uint_as_float(raw[0].z), __uint_as_float(raw[0].w));
    o.z = yl_pack_bf16(__uint_as_float(raw[1].x), __uint_as_float(raw[1].y));
    o.w = yl_pack_bf16(__uint_as_float(raw[1].z), __uint_as_float(raw[1].w));
    return o;
  }
};

// bf16 rows with the producer's BatchNorm + ReLU applied while staging (training with bf16 storage: the consumer of
// a lazily normalised activation): widen, fma with the per-column (scale, shift), floor, round back to bf16.
struct HProOp {
  const u16* p; long ld; int rows;
  const float* scale; const float* shift; float floor;
  static constexpr int NR = 1;
  __device__ __forceinline__ void load(int r, int k, u32x4* raw) const {
    const u32x4 v = *reinterpret_cast<const u32x4*>(p + (long)yl_min(r, rows - 1) * ld + k);
    const float4 s0 = *reinterpret_cast<const float4*>(scale + k), s1 = *reinterpret_cast<const float4*>(scale + k + 4);
    const float4 h0 = *reinterpret_cast<const float4*>(shift + k), h1 = *reinterpret_cast<const float4*>(shift + k + 4);
    u32x4 o;
    o.x = yl_pack_bf16(fmaxf(fmaf(yl_bf16_lo(v.x), s0.x, h0.x), floor), fmaxf(fmaf(yl_bf16_hi(v.x), s0.y, h0.y), floor));
    o.y = yl_pack_bf16(fmaxf(fmaf(yl_bf16_lo(v.y), s0.z, h0.z), floor), fmaxf(fmaf(yl_bf16_hi(v.y), s0.w, h0.w), floor));
    o.z = yl_pack_bf16(fmaxf(fmaf(yl_bf16_lo(v.z), s1.x, h1.x), floor), fmaxf(fmaf(yl_bf16_hi(v.z), s1.y, h1.y), floor));
    o.w = yl_pack_bf16(fmaxf(fmaf(yl_bf16_lo(v.w), s1.z, h1.z), floor), fmaxf(fmaf(yl_bf16_hi(v.w), s1.w, h1.w), floor));
    raw[0] = o;
  }
  static __device__ __forceinline__ u32x4 pack(const u32x4* raw) { return raw[0]; }
};

constexpr int YL_HRS = 72;   // LDS row stride in bf16 elements (144 B)
template <int TM, int TN> struct HTileSmem { static constexpr int elems = 64 * (TM + TN) * YL_HRS; };

template <int TM, int TN, class AL>
__device__ __forceinline__ void hgemm_tile(const AL& A, const HOp& B, const Epilogue& ep, int M, int N, int K, int rt_,
                                           int ct_, u16* smem) {
  constexpr int BM = 64 * TM, BN = 64 * TN, NA = 2 * TM, NB = 2 * TN;   // 16-byte slots per thread per K step
  u16* As = smem;
  u16* Bs = smem + BM * YL_HRS;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int wm = wave >> 1, wn = wave & 1, l31 = lane & 31, lhi = lane >> 5;
  const int row0 = rt_ * BM, col0 = ct_ * BN;
  const int sr = tid >> 3, sc = (tid & 7) * 8;          // staging role: row sr (+32 per slot), k chunk sc

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // PF K steps of operands in flight (round 5): with one, a long-K classifier tile kept 64 KB per CU on the way and ran at
  // the latency of its loads (1.2 us per K step for 256 cycles of MFMA per wave)
  constexpr int PF = 2;        // (3: 150 registers, no faster)
  u32x4 ra[PF][NA * AL::NR], rb[PF][NB];
#pragma unroll
  for (int q = 0; q < PF; ++q)
    if (64 * q < K) {
#pragma unroll
      for (int t = 0; t < NA; ++t) A.load(row0 + sr + 32 * t, 64 * q + sc, ra[q] + t * AL::NR);
#pragma unroll
      for (int t = 0; t < NB; ++t) B.load(col0 + sr + 32 * t, 64 * q + sc, rb[q] + t);
    }

  EpiPre pre[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
      pre[i][j] = epi_prefetch(ep, row0 + (wm * TM + i) * 32, col0 + (wn * TN + j) * 32 + l31, M, N);

  // one K step: stage the registers of step k0, refill them with step k0 + 64 PF, MFMAs
  auto kstep = [&](int k0, u32x4* qa, u32x4* qb) {
#pragma unroll
    for (int t = 0; t < NA; ++t)
      *reinterpret_cast<u32x4*>(As + (sr + 32 * t) * YL_HRS + sc) = AL::pack(qa + t * AL::NR);
#pragma unroll
    for (int t = 0; t < NB; ++t) *reinterpret_cast<u32x4*>(Bs + (sr + 32 * t) * YL_HRS + sc) = qb[t];
    __syncthreads();
    if (k0 + 64 * PF < K) {                 // PF steps ahead, in flight while the MFMAs below run
#pragma unroll
      for (int t = 0; t < NA; ++t) A.load(row0 + sr + 32 * t, k0 + 64 * PF + sc, qa + t * AL::NR);
#pragma unroll
      for (int t = 0; t < NB; ++t) B.load(col0 + sr + 32 * t, k0 + 64 * PF + sc, qb + t);
    }
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      bf16x8 a[TM], b[TN];
#pragma unroll
      for (int i = 0; i < TM; ++i)
        a[i] = *reinterpret_cast<const bf16x8*>(As + ((wm * TM + i) * 32 + l31) * YL_HRS + ks * 16 + lhi * 8);
#pragma unroll
      for (int j = 0; j < TN; ++j)
        b[j] = *reinterpret_cast<const bf16x8*>(Bs + ((wn * TN + j) * 32 + l31) * YL_HRS + ks * 16 + lhi * 8);
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
    }
    __syncthreads();
  };
  for (int k0 = 0; k0 < K; k0 += 64 * PF) {
#pragma unroll
    for (int q = 0; q < PF; ++q)
      if (k0 + 64 * q < K) kstep(k0 + 64 * q, ra[q], rb[q]);
  }
  static_assert(TM <= 2 && TN <= 2, "epilogue expansion covers up to 2x2 sub-tiles");
  // the operand tiles are dead after the loop's closing barrier: every wave stages its stores in its own 4.5 KB of them
  static_assert(HTileSmem<TM, TN>::elems * 2 >= 4 * 32 * YL_STAGE_LD * 4, "LDS too small for the store staging");
  float* stage = reinterpret_cast<float*>(smem) + wave * (32 * YL_STAGE_LD);
#define YL_EPI(i, j)                                                                                                 \
  if constexpr ((i) < TM && (j) < TN)                                                                                \
    wave_epilogue(acc[i][j], row0 + (wm * TM + (i)) * 32, col0 + (wn * TN + (j)) * 32 + l31, lhi, ep, M, N, pre[i][j], \
                  stage);
  YL_EPI(0, 0) YL_EPI(0, 1) YL_EPI(1, 0) YL_EPI(1, 1)
#undef YL_EPI
}

template <int TM, int TN, class AL>
static __global__ void __launch_bounds__(256) k_hgemm(AL A, HOp B, Epilogue ep, int M, int N, int K) {
  __shared__ __attribute__((aligned(16))) u16 smem[HTileSmem<TM, TN>::elems];
  int rt_, ct_;
  yl_xcd_tile(rt_, ct_);
  hgemm_tile<TM, TN, AL>(A, B, ep, M, N, K, rt_, ct_, smem);
}

// fusion block over the nodes (+ per-proposal max epilogue) and fusion_block_super over the per-proposal
// means in one flattened launch, like k_gemm_nt_two: the small problem's 64x64 tiles come first (padded to a
// multiple of 8 so that the big problem keeps its id % 8 = XCD alignment), then the big problem's tiles.
template <int T0>
static __global__ void __launch_bounds__(256) k_hgemm_two(HOp A0, HOp B0, Epilogue e0, int M0, int N0, int K0, int tm0,
                                                          int tn0, FOp A1, HOp B1, Epilogue e1, int M1, int N1, int K1,
                                                          int tm1, int tn1) {
  __shared__ __attribute__((aligned(16))) u16 smem[HTileSmem<T0, T0>::elems];
  const int n1 = tm1 * tn1, n1p = (n1 + 7) & ~7;
  const int id = blockIdx.x;
  if (id < n1p) {
    if (id < n1) hgemm_tile<1, 1, FOp>(A1, B1, e1, M1, N1, K1, id / tn1, id % tn1, smem);
    return;
  }
  const int n0 = tm0 * tn0, j = id - n1p;
  const int chunk = n0 >> 3, rem = n0 & 7;
  const int xcd = j & 7, slot = j >> 3;
  const int logical = xcd * chunk + (xcd < rem ? xcd : rem) + slot;
  hgemm_tile<T0, T0, HOp>(A0, B0, e0, M0, N0, K0, logical / tn0, logical % tn0, smem);
}

// ------------------------------------------------------------------------------------------------
// A-resident GEMM for a SHORT K (the fusion block: K = 128, N = 1024): with one (row, column) tile per workgroup
// every tile is two dependent global->LDS round trips in front of 16 MFMAs — at N = 200k rows the launch was
// bound by that latency (211 us; 21 us of MFMA work).  Here a workgroup keeps its (64 TM) x K row tile of A in
// LDS and walks `ng` consecutive 64-column tiles of W: the NEXT tile's weights and epilogue constants are loaded
// into registers while the current tile's MFMAs and pooling epilogue run, so only the first round trip of a
// workgroup is exposed.  K <= KD (compile-time LDS extent); columns k >= K are zero-filled.
// ------------------------------------------------------------------------------------------------
// Run structure of a lane's 16 rows, derived ONCE per row tile from their segment ids (it is the same for every
// column and every column tile): keep[r] = 1 if row r continues the run of row r-1 else 0; off[r] = 32-bit element
// offset of the row's proposal in the pooled matrix; flush_bits bit r = a run of a valid segment ends at row r;
// uflush bit r = some lane of the wave flushes at r (wave-uniform: rows where nobody flushes cost no exec traffic).
struct SegRuns { float keep[16]; unsigned off[16]; unsigned flush_bits, uflush; };
__device__ __forceinline__ void yl_seg_runs(const int sgs[16], unsigned ldpool, SegRuns& sr) {
  unsigned fb = 0, uf = 0;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    sr.keep[r] = (r > 0 && sgs[r] == sgs[r - 1]) ? 1.f : 0.f;
    sr.off[r] = (unsigned)(sgs[r] < 0 ? 0 : sgs[r]) * ldpool;
    const bool fl = sgs[r] >= 0 && (r == 15 || sgs[r] != sgs[r + 1]);
    fb |= fl ? (1u << r) : 0u;
    uf |= (__builtin_amdgcn_ballot_w64(fl) != 0ull) ? (1u << r) : 0u;
  }
  sr.flush_bits = fb; sr.uflush = uf;
}
// Pooling epilogue on that structure: per row one multiply (run reset: values are >= 0, so cur*0 restarts the max)
// and one 3-operand max (run max, activation, ReLU floor); the integer atomicMax (exact, order-independent) uses a
// 32-bit offset from a per-lane column pointer.  Same values as wave_epilogue_segmax's compare/select walk.
__device__ __forceinline__ void segmax_runs(const f32x16& acc, float* pool, unsigned col, bool col_ok, float sc,
                                            float sh, const SegRuns& sr) {
  float cur = 0.f;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    cur = fmaxf(fmaxf(cur * sr.keep[r], fmaf(acc[r], sc, sh)), 0.f);
    if ((sr.uflush >> r) & 1u) {
      if (((sr.flush_bits >> r) & 1u) && col_ok && cur > 0.f)
        atomicMax(reinterpret_cast<int*>(pool) + (sr.off[r] + col), __float_as_int(cur));   // uniform base + 32-bit offset
    }
  }
}

template <int TM, int KD> struct HRowsSmem { static constexpr int elems = 64 * (TM + 1) * (KD + 8); };

template <int TM, int KD>
__device__ __forceinline__ void hgemm_rows(const HOp& A, const HOp& B, const Epilogue& ep, int M, int N, int K, int rt_,
                                           int ct0, int ng, u16* smem) {
  constexpr int BM = 64 * TM, RS = KD + 8, CPR = KD / 8;     // row stride (elements), 16-byte chunks per row
  constexpr int NA = TM * KD / 32, NW = KD / 32;            // chunks per thread: A tile, one 64-column W tile
  u16* As = smem;
  u16* Bs = smem + BM * RS;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int wm = wave >> 1, wn = wave & 1, l31 = lane & 31, lhi = lane >> 5;
  const int row0 = rt_ * BM;
  const u32x4 zero4 = {0u, 0u, 0u, 0u};
  auto load_w = [&](int ct, u32x4* rw) {
#pragma unroll
    for (int t = 0; t < NW; ++t) {
      const int c = tid + 256 * t, r = c / CPR, kc = (c % CPR) * 8;
      const u32x4 v = *reinterpret_cast<const u32x4*>(B.p + (long)yl_min(ct * 64 + r, B.rows - 1) * B.ld + yl_min(kc, K - 8));
      rw[t] = kc < K ? v : zero4;
    }
  };
  u32x4 rw[NW];
  load_w(ct0, rw);
#pragma unroll
  for (int t = 0; t < NA; ++t) {
    const int c = tid + 256 * t, r = c / CPR, kc = (c % CPR) * 8;
    const u32x4 v = *reinterpret_cast<const u32x4*>(A.p + (long)yl_min(row0 + r, A.rows - 1) * A.ld + yl_min(kc, K - 8));
    *reinterpret_cast<u32x4*>(As + r * RS + kc) = kc < K ? v : zero4;
  }
  EpiPre pre[TM], nxt;
#pragma unroll
  for (int i = 0; i < TM; ++i) pre[i] = epi_prefetch(ep, row0 + (wm * TM + i) * 32, ct0 * 64 + wn * 32 + l31, M, N);
  nxt = pre[0];
  SegRuns runs[TM];                       // the rows' run structure is the same for every column tile
#pragma unroll
  for (int i = 0; i < TM; ++i) {
    int sgs[16];
    yl_tile_segs(pre[i].segv, lhi, sgs);
    yl_seg_runs(sgs, (unsigned)ep.ldpool, runs[i]);
  }
  for (int j = 0; j < ng; ++j) {
    const int ct = ct0 + j;
    __syncthreads();                      // the previous tile's fragment reads of Bs are done
#pragma unroll
    for (int t = 0; t < NW; ++t) {
      const int c = tid + 256 * t;
      *reinterpret_cast<u32x4*>(Bs + (c / CPR) * RS + (c % CPR) * 8) = rw[t];
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < TM; ++i) { pre[i].bias = nxt.bias; pre[i].sc = nxt.sc; pre[i].sh = nxt.sh; }
    if (j + 1 < ng) {                     // in flight while the MFMAs and the epilogue below run
      load_w(ct + 1, rw);
      nxt = epi_prefetch(ep, row0, (ct + 1) * 64 + wn * 32 + l31, M, N);
    }
    f32x16 acc[TM];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
#pragma unroll
    for (int ks = 0; ks < KD / 16; ++ks) {
      const bf16x8 b = *reinterpret_cast<const bf16x8*>(Bs + (wn * 32 + l31) * RS + ks * 16 + lhi * 8);
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        const bf16x8 a = *reinterpret_cast<const bf16x8*>(As + ((wm * TM + i) * 32 + l31) * RS + ks * 16 + lhi * 8);
        acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[i], 0, 0, 0);
      }
    }
    // pooling epilogue (this kernel exists for the fused per-proposal max): bias, folded BN + ReLU, run-length max
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      // (acc + bias)*sc + sh = acc*sc + (bias*sc + sh)
      const int colj = ct * 64 + wn * 32 + l31;
      segmax_runs(acc[i], ep.pool, (unsigned)colj, colj < N, pre[i].sc, fmaf(pre[i].bias, pre[i].sc, pre[i].sh), runs[i]);
    }
  }
}

// fusion block over the nodes (A-resident rows kernel, + per-proposal max epilogue) and fusion_block_super over the
// per-proposal means (plain 64x64 tiles, first and padded to a multiple of 8) in one flattened launch.  The big
// problem's workgroups are (row tile, column group) pairs, column group fastest, dealt to the XCDs in contiguous
// ranges so that the sharers of an A row tile hit one L2.
template <int TM, int KD>
static __global__ void __launch_bounds__(256) k_hfusion_rows(HOp A0, HOp B0, Epilogue e0, int M0, int N0, int K0, int tm0,
                                                             int groups, int ng, FOp A1, HOp B1, Epilogue e1, int M1,
                                                             int N1, int K1, int tm1, int tn1) {
  constexpr int SM = HRowsSmem<TM, KD>::elems > HTileSmem<1, 1>::elems ? HRowsSmem<TM, KD>::elems : HTileSmem<1, 1>::elems;
  __shared__ __attribute__((aligned(16))) u16 smem[SM];
  const int n1 = tm1 * tn1, n1p = (n1 + 7) & ~7;
  const int id = blockIdx.x;
  if (id < n1p) {
    if (id < n1) hgemm_tile<1, 1, FOp>(A1, B1, e1, M1, N1, K1, id / tn1, id % tn1, smem);
    return;
  }
  const int n0 = tm0 * groups, j = id - n1p;
  const int chunk = n0 >> 3, rem = n0 & 7;
  const int xcd = j & 7, slot = j >> 3;
  const int logical = xcd * chunk + (xcd < rem ? xcd : rem) + slot;
  const int rt = logical / groups, g = logical - rt * groups;
  const int tn = (N0 + 63) / 64;
  const int c0 = g * ng, cn = yl_min(ng, tn - c0);
  if (cn > 0) hgemm_rows<TM, KD>(A0, B0, e0, M0, N0, K0, rt, c0, cn, smem);
}

// node side of a factorised conv layer with Cin = 64 (bf16 in / bf16 weights):
//   y = 0,1 -> the two 64-column halves of UV = f_in.[W1a-W1b | W1b]^T   (stored bf16)
//   y = 2   -> root Linear lin_r(f_in) + br                               (stored fp32: the edge kernel adds the mean)
//   y = 3   -> node branch relu(bn(mlp_node(s_in)))                       (stored bf16)
struct NodeUvH {
  HOp af, as, wuv, wr, wn;
  Epilogue euv, er, en;
  int N, C, Cin;
};
static __global__ void __launch_bounds__(256) k_hgemm_node3(NodeUvH a) {
  __shared__ __attribute__((aligned(16))) u16 smem[HTileSmem<1, 1>::elems];
  const int x = blockIdx.x, y = blockIdx.y;
  if (y < 2) hgemm_tile<1, 1, HOp>(a.af, a.wuv, a.euv, a.N, 2 * a.C, a.Cin, x, y, smem);
  else if (y == 2) hgemm_tile<1, 1, HOp>(a.af, a.wr, a.er, a.N, a.C, a.Cin, x, 0, smem);
  else hgemm_tile<1, 1, HOp>(a.as, a.wn, a.en, a.N, a.C, a.Cin, x, 0, smem);
}
// the same as the fall-back of the one-launch conv stack (conv_local.hip): dead unless the gate word holds its value.  (A
// persistent-grid form — a few hundred workgroups looping over the row tiles, so that the dead launch dispatches fewer
// workgroups — made the compiler keep 233 registers around the loop, or spill 140 under a bound: the live fall-back then
// ran at half speed; laundering the loop variable did not change that, and the tile as a real (noinline) call spilled
// 560 bytes per lane.  The full grid costs ~4 us per dead launch instead of ~2.)
static __global__ void __launch_bounds__(256) k_hgemm_node3_gated(NodeUvH a, YlGate gate) {
  __shared__ __attribute__((aligned(16))) u16 smem[HTileSmem<1, 1>::elems];
  if (yl_gate_dead(gate)) return;
  const int x = blockIdx.x, y = blockIdx.y;
  if (y < 2) hgemm_tile<1, 1, HOp>(a.af, a.wuv, a.euv, a.N, 2 * a.C, a.Cin, x, y, smem);
  else if (y == 2) hgemm_tile<1, 1, HOp>(a.af, a.wr, a.er, a.N, a.C, a.Cin, x, 0, smem);
  else hgemm_tile<1, 1, HOp>(a.as, a.wn, a.en, a.N, a.C, a.Cin, x, 0, smem);
}

static Epilogue plain_epilogue() {
  Epilogue e;
  e.bias = nullptr; e.scale = nullptr; e.shift = nullptr; e.relu = 0;
  e.Y = nullptr; e.ldy = 0; e.accumulate = 0; e.stats = nullptr; e.seg = nullptr; e.pool = nullptr; e.ldpool = 0;
  return e;
}

// ------------------------------------------------------------------------------------------------
// Factorised edge MLP + mean aggregation, bf16 storage (see k_edge_uv_mlp2_mean, edge.hip, for the scheme).
// One workgroup = npt (<= 16) consecutive destination nodes = a contiguous CSR edge range, in passes of 64
// edges.  Per pass: gather U'[dst] + V'[src] (bf16, 128 B per row) + W1c'.attr -> ReLU (fp32) -> bf16 tile in
// LDS -> second Linear on bf16 MFMAs (this wave's W2 fragments live in registers for the whole kernel) ->
// BN+ReLU -> fp32 tile in LDS -> per-node running sums in CSR order.  Output: f_out = bf16(root + sum/deg)
// for EVERY node of the tile (nodes without in-edges get the root Linear alone, torch_vertex.py:324,337).
//
// Built to cut VALU work (PMC at cfg 5: a first version that applied layer 1's BatchNorm per edge in scalar fp32
// kept the vector ALUs ~60 % busy — the kernel is instruction-bound, not latency-bound; a higher-occupancy variant
// with the per-column constants in LDS was slower, 480 vs 435 us for the 4 layers; this one takes 383 us):
//   * layer 1's folded BatchNorm is applied where it is cheap: the node-side GEMM epilogue stores
//     U' = s1*U + (s1*b1 + t1) and V' = s1*V (uv_scale / uv_shift of yolat_model_eval_bf16), this kernel scales the
//     four W1c columns once per workgroup, so a hidden activation is relu(U' + V' + W1c'.attr): 6 flops instead of 8;
//   * those run as packed fp32 pairs (v_pk_add_f32 / v_pk_fma_f32), as does layer 2's folded BN (bias folded into
//     the shift);
//   * 32-bit element offsets for the gathers, 16-byte message rows (LDM = 68) for the per-node sums.
typedef float f32x2 __attribute__((ext_vector_type(2)));
// NG = 16-node groups per tile: 1 (<= 16 nodes) or 4 (<= 64 nodes, for graphs with ~1 edge per node).
template <int NG>
static __global__ void __launch_bounds__(256, (NG == 1 ? 4 : 3)) k_edge_uv_mlp2_mean_h(
    const u16* __restrict__ UV, unsigned ld_uv, const int* __restrict__ src, const int* __restrict__ dst,
    const float* __restrict__ attr, const int* __restrict__ row_ptr, int N, int npt, const float* __restrict__ Wc4,
    const float* __restrict__ s1, const u16* __restrict__ W2h, const float* __restrict__ b2,
    const float* __restrict__ s2, const float* __restrict__ t2, const float* __restrict__ root, unsigned ld_r,
    u16* __restrict__ f_out, unsigned ld_fo, int E, YlGate gate) {
  if (yl_gate_dead(gate)) return;      // fall-back of the one-launch conv stack (conv_local.hip): dead launch
  constexpr int LDM = 68;
  __shared__ __attribute__((aligned(16))) u16 Hs[64 * YL_HRS];   // layer-1 activations of the pass (bf16)
  __shared__ __attribute__((aligned(16))) float Ms[64 * LDM];   // layer-2 messages of the pass (fp32)
  __shared__ int rp[16 * NG + 1];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int wm = wave >> 1, wn = wave & 1, l31 = lane & 31, lhi = lane >> 5;
  const int n0 = blockIdx.x * npt;
  const int nn = yl_min(npt, N - n0);
  if (tid <= 16 * NG) rp[tid] = row_ptr[yl_min(n0 + tid, n0 + nn)];
  const int q = tid & 15, rb = tid >> 4;              // gather role: columns 4q..4q+3 of rows rb + 16t
  const int col = wn * 32 + l31;                      // MFMA role: output column of this lane
  bf16x8 w2f[4];
#pragma unroll
  for (int ks = 0; ks < 4; ++ks)
    w2f[ks] = *reinterpret_cast<const bf16x8*>(W2h + col * 64 + ks * 16 + lhi * 8);
  // W1c' = s1 * W1c for this thread's 4 columns, as (column pair) x (attr component) packed pairs
  f32x2 wp[2][4];
  {
    float4 sc = make_float4(1.f, 1.f, 1.f, 1.f);
    if (s1) sc = *reinterpret_cast<const float4*>(s1 + 4 * q);
    const float scv[4] = {sc.x, sc.y, sc.z, sc.w};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float4 w = *reinterpret_cast<const float4*>(Wc4 + (4 * q + j) * 4);
      wp[j >> 1][0][j & 1] = w.x * scv[j]; wp[j >> 1][1][j & 1] = w.y * scv[j];
      wp[j >> 1][2][j & 1] = w.z * scv[j]; wp[j >> 1][3][j & 1] = w.w * scv[j];
    }
  }
  // folded form (the product path since ABI 4): W2h already carries s2, t2 is the complete shift -> b2 = s2 = NULL
  const float sc2 = s2 ? s2[col] : 1.f;
  const float sh2 = fmaf(b2 ? b2[col] : 0.f, sc2, t2 ? t2[col] : 0.f);    // (acc + b2)*s2 + t2 = acc*s2 + sh2
  __syncthreads();
  const int e0 = rp[0], e1 = rp[nn];
  int my_b[NG], my_e[NG];                             // aggregation role: nodes rb + 16 j, columns 4q..
  float4 rootv[NG], sum[NG];
#pragma unroll
  for (int j = 0; j < NG; ++j) {
    my_b[j] = rp[yl_min(rb + 16 * j, nn)]; my_e[j] = rp[yl_min(rb + 16 * j + 1, nn)];
    rootv[j] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (rb + 16 * j < nn) rootv[j] = *reinterpret_cast<const float4*>(root + (unsigned)(n0 + rb + 16 * j) * ld_r + 4 * q);
    sum[j] = make_float4(0.f, 0.f, 0.f, 0.f);
  }
  for (int c0 = e0; c0 < e1; c0 += 64) {
    unsigned di[4], si[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const int e = yl_min(c0 + rb + 16 * t, E - 1);
      di[t] = (unsigned)dst[e]; si[t] = (unsigned)src[e];
    }
    u32x2 u[4], v[4];
    float4 a[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const unsigned e = (unsigned)yl_min(c0 + rb + 16 * t, E - 1);
      u[t] = *reinterpret_cast<const u32x2*>(UV + (di[t] * ld_uv + 4 * q));
      v[t] = *reinterpret_cast<const u32x2*>(UV + (si[t] * ld_uv + 64 + 4 * q));
      a[t] = *reinterpret_cast<const float4*>(attr + e * 4u);
    }
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      f32x2 z0, z1, uu, vv;
      uu.x = yl_bf16_lo(u[t].x); uu.y = yl_bf16_hi(u[t].x); vv.x = yl_bf16_lo(v[t].x); vv.y = yl_bf16_hi(v[t].x);
      z0 = uu + vv;
      uu.x = yl_bf16_lo(u[t].y); uu.y = yl_bf16_hi(u[t].y); vv.x = yl_bf16_lo(v[t].y); vv.y = yl_bf16_hi(v[t].y);
      z1 = uu + vv;
      const f32x2 ax = {a[t].x, a[t].x}, ay = {a[t].y, a[t].y}, az = {a[t].z, a[t].z}, aw = {a[t].w, a[t].w};
      z0 = __builtin_elementwise_fma(ax, wp[0][0], z0); z1 = __builtin_elementwise_fma(ax, wp[1][0], z1);
      z0 = __builtin_elementwise_fma(ay, wp[0][1], z0); z1 = __builtin_elementwise_fma(ay, wp[1][1], z1);
      z0 = __builtin_elementwise_fma(az, wp[0][2], z0); z1 = __builtin_elementwise_fma(az, wp[1][2], z1);
      z0 = __builtin_elementwise_fma(aw, wp[0][3], z0); z1 = __builtin_elementwise_fma(aw, wp[1][3], z1);
      u32x2 hp;
      hp.x = yl_pack_bf16(fmaxf(z0.x, 0.f), fmaxf(z0.y, 0.f));
      hp.y = yl_pack_bf16(fmaxf(z1.x, 0.f), fmaxf(z1.y, 0.f));
      *reinterpret_cast<u32x2*>(Hs + (rb + 16 * t) * YL_HRS + 4 * q) = hp;
    }
    __syncthreads();                      // Hs complete; also: every thread is past the previous pass's Ms reads
    f32x16 acc2;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc2[r] = 0.f;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      const bf16x8 av = *reinterpret_cast<const bf16x8*>(Hs + (wm * 32 + l31) * YL_HRS + ks * 16 + lhi * 8);
      acc2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, w2f[ks], acc2, 0, 0, 0);
    }
    {
      const f32x2 s2p = {sc2, sc2}, h2p = {sh2, sh2};
#pragma unroll
      for (int r = 0; r < 16; r += 2) {
        f32x2 m = {acc2[r], acc2[r + 1]};
        m = __builtin_elementwise_fma(m, s2p, h2p);
        const int row = wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhi;
        Ms[row * LDM + col] = fmaxf(m.x, 0.f);
        Ms[(row + 1) * LDM + col] = fmaxf(m.y, 0.f);
      }
    }
    __syncthreads();                      // Ms complete; every wave is done reading Hs
#pragma unroll
    for (int j = 0; j < NG; ++j) {
      if (rb + 16 * j < nn) {             // rows of node rb + 16 j inside this pass, ascending edge order
        const int lo = (my_b[j] > c0 ? my_b[j] : c0) - c0;
        const int hi = (my_e[j] < c0 + 64 ? my_e[j] : c0 + 64) - c0;
        for (int e = lo; e < hi; ++e) {
          const float4 m = *reinterpret_cast<const float4*>(Ms + e * LDM + 4 * q);
          sum[j].x += m.x; sum[j].y += m.y; sum[j].z += m.z; sum[j].w += m.w;
        }
      }
    }
  }
#pragma unroll
  for (int j = 0; j < NG; ++j) {
    if (rb + 16 * j < nn) {
      const int deg = my_e[j] - my_b[j];
      const float inv = 1.f / (float)(deg > 1 ? deg : 1);
      u32x2 o;
      o.x = yl_pack_bf16(fmaf(sum[j].x, inv, rootv[j].x), fmaf(sum[j].y, inv, rootv[j].y));
      o.y = yl_pack_bf16(fmaf(sum[j].z, inv, rootv[j].z), fmaf(sum[j].w, inv, rootv[j].w));
      *reinterpret_cast<u32x2*>(f_out + ((unsigned)(n0 + rb + 16 * j) * ld_fo + 4 * q)) = o;
    }
  }
}

// ------------------------------------------------------------------------------------------------
// Pooling prologue on bf16 node features (k_pool_prepare of segment.hip): for proposal p
//   Z[p, 0:F] = 0;  Z[p, F:F+D] = max over rows of feats;  Z[p, 2F+D:2F+2D] = mean over rows of fsup   (fp32)
// ------------------------------------------------------------------------------------------------
static __global__ void __launch_bounds__(256) k_pool_prepare_h(const u16* feats, const u16* fsup, long ld, int D, int F,
                                                               const int* seg_ptr, float* Z, long ldz,
                                                               YlGate gate, int np) {
  if (yl_gate_dead(gate)) return;      // fall-back of the one-launch conv stack (conv_local.hip): dead launch
  const int c = blockIdx.x * 256 + threadIdx.x;
  // (the gated launch covers its np proposals with a few hundred rows of workgroups: gridDim.y < np)
  for (int p = blockIdx.y; p < np; p += gridDim.y) {
    float* z = Z + (long)p * ldz;
    if (c < F) { z[c] = 0.f; continue; }
    if (c >= F + 2 * D) continue;
    const int r0 = seg_ptr[p], r1 = seg_ptr[p + 1];
    const bool is_max = c < F + D;
    const int k = is_max ? c - F : c - F - D;
    const u16* srcp = (is_max ? feats : fsup) + k;
    float best = 0.f, s = 0.f;
    bool any = false;
    int r = r0;
    for (; r + 8 <= r1; r += 8) {
      float v[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = __uint_as_float((unsigned)srcp[(long)(r + j) * ld] << 16);
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        if (!any || v[j] > best) { best = v[j]; any = true; }
        s += v[j];
      }
    }
    for (; r < r1; ++r) {
      const float v = __uint_as_float((unsigned)srcp[(long)r * ld] << 16);
      if (!any || v > best) { best = v; any = true; }
      s += v;
    }
    if (is_max) {
      z[F + k] = best;
    } else {
      const int cnt = r1 - r0;
      z[2 * F + D + k] = s / (float)(cnt > 1 ? cnt : 1);
    }
  }
}

// ------------------------------------------------------------------------------------------------
// edge stage dispatch: register-chained MFMA waves (edge_chain.hip) for large, dense-enough graphs, node tiles
// otherwise.  Both take the folded second Linear (W2f = bf16(diag(s2) W2), t2f = s2 b2 + t2).
// ------------------------------------------------------------------------------------------------
int yl_edge_chain_bf16(const uint16_t* UV, int64_t ld_uv, const int32_t* src_csr, const int32_t* dst_csr,
                       const float* attr_csr, const int32_t* row_ptr, int64_t N, int64_t E, const float* Wc4,
                       const float* s1, const uint16_t* W2f, const float* t2f, const float* root, int64_t ld_r,
                       uint16_t* f_out, int64_t ld_fo, hipStream_t st, YlGate gate);

bool yl_conv_local_model_ok(const yolat_model_eval_bf16* mh);
int yl_conv_local_bf16(const yolat_model_eval_bf16* mh, const void* pack, const float* x, int64_t ldx, const YlLocalIn& in,
                       int64_t N, int64_t E, int64_t P, uint16_t* feats, int64_t ld_feats, float* Z, int64_t ldz, int32_t* flag,
                       int32_t flag_val, hipStream_t st);
// YOLAT_CONV_LOCAL: 0 = never, 1 = batches with >= 1024 proposals (default), 2 = every batch
// (read per call: tests flip it in-process)
static __global__ void k_zero_word(int* p) { if (threadIdx.x == 0) *p = 0; }
static int yl_conv_local_mode() {
  const char* e = getenv("YOLAT_CONV_LOCAL");
  return (e && e[0] >= '0' && e[0] <= '3') ? e[0] - '0' : 1;      // (3: measurement only — forced, WITHOUT the gated fall-back)
}

int yl_hfusion_rows8(const uint16_t* A, int64_t lda, int64_t N, int64_t D, const uint16_t* Wf, const float* tf,
                     const int32_t* seg, float* pool, int64_t ld_pool, int64_t F, const float* As, int64_t lda_s, int64_t P,
                     const uint16_t* Wfs, const float* tfs, float* sup_out, int64_t ld_sup, hipStream_t st);

static int yl_edge_uv_mlp2_mean_bf16_impl(const u16* UV, long ld_uv, const int* src, const int* dst, const float* attr,
                                          const int* row_ptr, long N, long E, const float* Wc4, const float* s1,
                                          const u16* W2f, const float* t2f, const float* root, long ld_r, u16* f_out,
                                          long ld_fo, int variant, hipStream_t st, YlGate gate = YlGate{nullptr, 0}) {
  // the chained kernel walks the EDGE list: it needs enough edges to fill 2048 waves and pays per finished node
  if (variant == 0) variant = (E >= 131072 && E >= 2 * N) ? 2 : 1;
  if (variant == 2) {
    if (E < 16) return YOLAT_E_UNSUPPORTED;
    return yl_edge_chain_bf16(UV, ld_uv, src, dst, attr, row_ptr, N, E, Wc4, s1, W2f, t2f, root, ld_r, f_out, ld_fo, st, gate);
  }
  long npt = E > 0 ? (56 * N) / E : 64;
  {
    const long npt2 = E > 0 ? ((112 * N) / E < 16 ? (112 * N) / E : 16) : 16;
    if (npt2 >= 2 * npt - 2 && N / (npt2 > 0 ? npt2 : 1) >= 8192) npt = npt2;
    if (npt < 1) npt = 1;
    if (npt > 64) npt = 64;
  }
  hipLaunchKernelGGL((npt <= 16 ? k_edge_uv_mlp2_mean_h<1> : k_edge_uv_mlp2_mean_h<4>), dim3(yl_cdiv(N, npt)), dim3(256), 0,
                     st, UV, (unsigned)ld_uv, src, dst, attr, row_ptr, (int)N, (int)npt, Wc4, s1, W2f, (const float*)nullptr,
                     (const float*)nullptr, t2f, root, (unsigned)ld_r, f_out, (unsigned)ld_fo, (int)(E > 0 ? E : 1), gate);
  YL_LAUNCH_CHECK();
  return 0;
}

extern "C" int yolat_edge_uv_mlp2_mean_eval_bf16(const uint16_t* UV, int64_t ld_uv, const int32_t* src_csr,
                                                 const int32_t* dst_csr, const float* attr_csr, const int32_t* row_ptr,
                                                 int64_t N, int64_t E, const float* Wc4, const float* s1,
                                                 const uint16_t* W2f, const float* t2f, const float* root, int64_t ld_r,
                                                 uint16_t* f_out, int64_t ld_fo, int variant, yolat_stream_t stream) {
  if (!UV || !row_ptr || !Wc4 || !W2f || !t2f || !root || !f_out || N <= 0 || E < 0 || variant < 0 || variant > 2)
    return YOLAT_E_INVALID;
  if (E > 0 && (!src_csr || !dst_csr || !attr_csr)) return YOLAT_E_INVALID;
  if (N > (1LL << 23) || E > (1LL << 29)) return YOLAT_E_UNSUPPORTED;          // 32-bit element offsets in the gathers
  if (ld_uv < 128 || ld_uv % 8 != 0 || ld_r < 64 || ld_r % 4 != 0 || ld_fo < 64 || ld_fo % 4 != 0 || !yl_aligned16(UV) ||
      !yl_aligned16(root) || !yl_aligned16(attr_csr) || !yl_aligned16(Wc4) || (((uintptr_t)f_out) & 7) != 0 ||
      (((uintptr_t)W2f) & 7) != 0)
    return YOLAT_E_UNSUPPORTED;
  return yl_edge_uv_mlp2_mean_bf16_impl(UV, ld_uv, src_csr, dst_csr, attr_csr, row_ptr, N, E, Wc4, s1, W2f, t2f, root, ld_r,
                                        f_out, ld_fo, variant, (hipStream_t)stream);
}

// ------------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------------
namespace {
struct Carver {
  char* base; size_t off;
  template <class T> T* take(size_t n) {
    off = (off + 255) & ~(size_t)255;
    T* p = reinterpret_cast<T*>(base + off);
    off += n * sizeof(T);
    return p;
  }
};
struct PlanH {
  int* row_ptr; int* perm; int* src; int* dst; float* attr; int* work; int* seg_ptr; int* node_seg;
  u16* UV; float* root; u16* f_tmp[YOLAT_MAX_LAYERS]; u16* s_tmp[YOLAT_MAX_LAYERS];
  u16* feats; u16* fsup; float* Z; float* c1; float* c2;
  int* local_flag; int* eptr;
  size_t bytes;
};
PlanH carve_h(const yolat_model_eval* m, long N, long E, long P, void* ws) {
  Carver c; c.base = reinterpret_cast<char*>(ws); c.off = 0;
  PlanH p;
  const long C = m->C, F = m->F, D = C * m->n_blocks_out, Ee = E > 0 ? E : 1;
  p.row_ptr = c.take<int>(N + 1); p.perm = c.take<int>(Ee); p.src = c.take<int>(Ee); p.dst = c.take<int>(Ee);
  p.attr = c.take<float>(Ee * 4); p.work = c.take<int>(yolat_graph_work_elems(N, E));
  p.seg_ptr = c.take<int>(P + 1); p.node_seg = c.take<int>(N);
  p.UV = c.take<u16>(N * 2 * C); p.root = c.take<float>(N * C);
  const int lo = m->n_blocks - m->n_blocks_out;
  for (int l = 0; l < m->n_blocks; ++l) {
    p.f_tmp[l] = (l < lo) ? c.take<u16>(N * C) : nullptr;
    p.s_tmp[l] = (l < lo) ? c.take<u16>(N * C) : nullptr;
  }
  p.feats = c.take<u16>(N * D); p.fsup = c.take<u16>(N * D);
  p.Z = c.take<float>(P * 2 * (F + D)); p.c1 = c.take<float>(P * m->H1); p.c2 = c.take<float>(P * m->H2);
  p.local_flag = c.take<int>(64);
  p.eptr = c.take<int>(P + 1);
  p.bytes = c.off + 256;
  return p;
}

int model_ok(const yolat_model_eval_bf16* mh) {
  if (!mh || !mh->base) return YOLAT_E_INVALID;
  const yolat_model_eval* m = mh->base;
  if (m->n_blocks < 1 || m->n_blocks > YOLAT_MAX_LAYERS || m->n_blocks_out < 1 || m->n_blocks_out > m->n_blocks)
    return YOLAT_E_INVALID;
  const long D = m->C * m->n_blocks_out;
  // shapes the bf16 kernels are written for (the reference's: n_filters 64, fusion 1024, classifier 512/256)
  if (m->C != 64 || m->F % 64 != 0 || D % 64 != 0 || (2 * (m->F + D)) % 64 != 0 || m->H1 % 64 != 0 || m->H2 % 64 != 0)
    return YOLAT_E_UNSUPPORTED;
  for (int l = 0; l < m->n_blocks; ++l) {
    const yolat_conv_eval& cv = m->conv[l];
    if (!cv.Wuv || !cv.Wc4 || !mh->W2[l] || !mh->t2f[l] || !mh->uv_scale[l] || !mh->uv_shift[l]) return YOLAT_E_INVALID;
    if (l == 0 ? (cv.Cin > 16) : (cv.Cin != 64)) return YOLAT_E_UNSUPPORTED;
    if (l > 0 && (!mh->Wuv[l] || !mh->Wr[l] || !mh->Wn[l])) return YOLAT_E_INVALID;
  }
  if (!mh->Wf || !mh->Wfs || !mh->Wc1 || !mh->Wc2 || !mh->Wc3) return YOLAT_E_INVALID;
  return 0;
}

template <class AL>
int launch_hgemm(const AL& A, const HOp& B, const Epilogue& ep, long M, long N, long K, hipStream_t st) {
  if (K % 64 != 0 || M <= 0 || N <= 0) return YOLAT_E_UNSUPPORTED;
  // 64x128 tiles once they still fill the GPU (cfg 5 classifier 1: 55 -> 45 us)
  if (N % 128 == 0 && (long)yl_cdiv(M, 64) * (N / 128) >= 256)
    hipLaunchKernelGGL((k_hgemm<1, 2, AL>), dim3(yl_cdiv(M, 64), N / 128), dim3(256), 0, st, A, B, ep, (int)M, (int)N,
                       (int)K);
  else
  hipLaunchKernelGGL((k_hgemm<1, 1, AL>), dim3(yl_cdiv(M, 64), yl_cdiv(N, 64)), dim3(256), 0, st, A, B, ep, (int)M, (int)N,
                     (int)K);
  YL_LAUNCH_CHECK();
  return 0;
}
}  // namespace

#define YL_TRY(call)            \
  do {                          \
    int rc__ = (call);          \
    if (rc__ != 0) return rc__; \
  } while (0)
// one profiled stage: `body` is a statement block that may `return` an error code
#define YL_HSTAGE(name, flops, bytes, ...)                                    \
  do {                                                                        \
    const bool prof__ = yl_profile_on();                                      \
    if (prof__) yl_stage_begin(name, (double)(flops), (double)(bytes), stream); \
    __VA_ARGS__                                                               \
    if (prof__) yl_stage_end(stream);                                         \
  } while (0)

extern "C" size_t yolat_forward_eval_bf16_workspace_bytes(const yolat_model_eval_bf16* mh, int64_t N, int64_t E,
                                                          int64_t P) {
  if (!mh || !mh->base || N <= 0 || E < 0 || P <= 0) return 0;
  return carve_h(mh->base, N, E, P, nullptr).bytes;
}

static int forward_eval_bf16_impl(const yolat_model_eval_bf16* mh, const float* x, int64_t ldx, const int64_t* edge,
                                  int64_t stride_e, int64_t stride_c, const float* e_attr, const int64_t* bbox_idx,
                                  int64_t N, int64_t E, int64_t P, float* logits, int64_t ld_logits, void* workspace,
                                  size_t workspace_bytes, int32_t* status, yolat_stream_t stream, const yolat_graph_csr* g,
                                  bool primed = false, const yolat_locality* loc = nullptr);

extern "C" int yolat_forward_eval_bf16(const yolat_model_eval_bf16* mh, const float* x, int64_t ldx,
                                       const int64_t* edge, int64_t stride_e, int64_t stride_c, const float* e_attr,
                                       const int64_t* bbox_idx, int64_t N, int64_t E, int64_t P, float* logits,
                                       int64_t ld_logits, void* workspace, size_t workspace_bytes, int32_t* status,
                                       yolat_stream_t stream) {
  return forward_eval_bf16_impl(mh, x, ldx, edge, stride_e, stride_c, e_attr, bbox_idx, N, E, P, logits, ld_logits,
                                workspace, workspace_bytes, status, stream, nullptr);
}

// The same forward for a caller that vouches for the workspace: its previous use was yolat_forward_eval_bf16 /
// _bf16_primed with the same m layout, N, E and P on the same stream (every forward leaves the CSR-build counters zero,
// so the memset launch is skipped — the contract of yolat_forward_eval_primed).
extern "C" int yolat_forward_eval_bf16_primed(const yolat_model_eval_bf16* mh, const float* x, int64_t ldx,
                                              const int64_t* edge, int64_t stride_e, int64_t stride_c,
                                              const float* e_attr, const int64_t* bbox_idx, int64_t N, int64_t E,
                                              int64_t P, float* logits, int64_t ld_logits, void* workspace,
                                              size_t workspace_bytes, int32_t* status, yolat_stream_t stream) {
  return forward_eval_bf16_impl(mh, x, ldx, edge, stride_e, stride_c, e_attr, bbox_idx, N, E, P, logits, ld_logits,
                                workspace, workspace_bytes, status, stream, nullptr, true);
}

extern "C" int yolat_forward_eval_bf16_csr(const yolat_model_eval_bf16* mh, const float* x, int64_t ldx,
                                           const yolat_graph_csr* g, int64_t N, int64_t E, int64_t P, float* logits,
                                           int64_t ld_logits, void* workspace, size_t workspace_bytes,
                                           yolat_stream_t stream) {
  if (!g || !g->row_ptr || !g->seg_ptr || !g->node_seg || (E > 0 && (!g->src || !g->dst || !g->attr)))
    return YOLAT_E_INVALID;
  int32_t unused_status = 0;
  return forward_eval_bf16_impl(mh, x, ldx, nullptr, 0, 0, nullptr, reinterpret_cast<const int64_t*>(g->node_seg), N, E, P,
                                logits, ld_logits, workspace, workspace_bytes, &unused_status, stream, g);
}

extern "C" int yolat_forward_eval_bf16_loc(const yolat_model_eval_bf16* mh, const float* x, int64_t ldx,
                                           const int64_t* edge, int64_t stride_e, int64_t stride_c, const float* e_attr,
                                           const int64_t* bbox_idx, const yolat_graph_csr* g, int64_t N, int64_t E,
                                           int64_t P, float* logits, int64_t ld_logits, void* workspace,
                                           size_t workspace_bytes, int32_t* status, const yolat_locality* loc, int primed,
                                           yolat_stream_t stream) {
  if (g != nullptr) {
    if (!g->row_ptr || !g->seg_ptr || !g->node_seg || (E > 0 && (!g->src || !g->dst || !g->attr))) return YOLAT_E_INVALID;
    if (!status) return YOLAT_E_INVALID;
    return forward_eval_bf16_impl(mh, x, ldx, nullptr, 0, 0, nullptr, reinterpret_cast<const int64_t*>(g->node_seg), N, E, P,
                                  logits, ld_logits, workspace, workspace_bytes, status, stream, g, false, loc);
  }
  return forward_eval_bf16_impl(mh, x, ldx, edge, stride_e, stride_c, e_attr, bbox_idx, N, E, P, logits, ld_logits,
                                workspace, workspace_bytes, status, stream, nullptr, primed != 0, loc);
}

static int forward_eval_bf16_impl(const yolat_model_eval_bf16* mh, const float* x, int64_t ldx, const int64_t* edge,
                                  int64_t stride_e, int64_t stride_c, const float* e_attr, const int64_t* bbox_idx,
                                  int64_t N, int64_t E, int64_t P, float* logits, int64_t ld_logits, void* workspace,
                                  size_t workspace_bytes, int32_t* status, yolat_stream_t stream,
                                  const yolat_graph_csr* g, bool primed, const yolat_locality* loc) {
  if (!x || !bbox_idx || !logits || !workspace || !status || N <= 0 || E < 0 || P <= 0) return YOLAT_E_INVALID;
  YL_TRY(model_ok(mh));
  if (N > (1LL << 23) || E > (1LL << 29)) return YOLAT_E_UNSUPPORTED;     // 32-bit element offsets in the gathers
  if (P * 2 * (mh->base->F + mh->base->C * mh->base->n_blocks_out) >= (1LL << 32)) return YOLAT_E_UNSUPPORTED;
  const yolat_model_eval* m = mh->base;
  PlanH p = carve_h(m, N, E, P, workspace);
  if (p.bytes > workspace_bytes) return YOLAT_E_INVALID;
  if (g != nullptr) {                 // prepared graph (collate.hip): the caller's arrays, never written here
    p.row_ptr = const_cast<int*>(g->row_ptr); p.src = const_cast<int*>(g->src); p.dst = const_cast<int*>(g->dst);
    p.attr = const_cast<float*>(g->attr); p.seg_ptr = const_cast<int*>(g->seg_ptr);
    p.node_seg = const_cast<int*>(g->node_seg);
  }
  hipStream_t st = (hipStream_t)stream;
  const long C = m->C, F = m->F, D = C * m->n_blocks_out, ZW = 2 * (F + D);
  const int lo = m->n_blocks - m->n_blocks_out;
  auto f_slot = [&](int l) { return l - lo >= 0 ? p.feats + (l - lo) * C : p.f_tmp[l]; };
  auto s_slot = [&](int l) { return l - lo >= 0 ? p.fsup + (l - lo) * C : p.s_tmp[l]; };
  auto ld_slot = [&](int l) { return l - lo >= 0 ? D : C; };

  // ---- the one-launch conv stack (conv_local.hip) applies to models of its shapes on batches large enough to fill the
  // chip; whether THIS batch has the property (edges inside their proposal, proposals that fit a tile) is found out on the
  // device: the per-layer launches below stay enqueued, gated on the word the conv kernel raises
  const yolat_conv_eval& cv0 = m->conv[0];
  NodeUv a0;
  // the fp32 destinations are placeholders for the builder's checks; UV and the node branch go to bf16
  YL_TRY(yl_build_node_uv(&a0, x, ldx, x, ldx, N, cv0.Cin, cv0.Wuv, nullptr, cv0.Wr, cv0.br, cv0.Wn, cv0.bn, cv0.sn, cv0.tn, C,
                          p.root, 2 * C, p.root, C, p.root, C));
  a0.euv.Y = nullptr; a0.euv.Yh = p.UV; a0.euv.ldy = 2 * C;
  a0.euv.scale = mh->uv_scale[0]; a0.euv.shift = mh->uv_shift[0];
  a0.en.Y = nullptr; a0.en.Yh = s_slot(0); a0.en.ldy = ld_slot(0);
  const int local_mode = yl_conv_local_mode();
  const bool known = loc != nullptr && loc->known != 0;
  const bool fits = known && yolat_conv_local_fits(loc, P) != 0;
  // (a batch known NOT to have the property goes straight to the per-layer launches)
  const bool local = local_mode != 0 && mh->conv_local != nullptr && yl_conv_local_model_ok(mh) &&
                     yl_node3_smallk_shape_ok(a0) && D % 8 == 0 && ZW % 4 == 0 && (local_mode >= 2 || P >= 1024) &&
                     (!known || fits) && (g != nullptr || E == 0 || yl_aligned16(e_attr) || !fits);
  // vouched: the caller examined the batch (yolat_batch_locality / the host collate) — no gated fall-back launches, and
  // for a COO batch no global destination sort either: the tiles sort their edges themselves
  const bool vouched = local && fits && (g != nullptr || E > 0);      // (no edges: nothing to sort, the gated form has no cost)
  YlGate gate{nullptr, 0};
  int flag_val = 0;
  if (local) {
    // the word the conv kernel raises: a fresh value per forward instead of a reset launch.  High bit set: the workspace
    // is re-carved per shape and may hold stale int32 index data (< 2^30) at this address, which must never look raised
    static std::atomic<unsigned> epoch_counter{0};
    flag_val = (int)(0x80000000u | (++epoch_counter & 0x7FFFFFFFu));
    if (!vouched) { gate.p = p.local_flag; gate.val = flag_val; }
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing((hipStream_t)stream, &cap) == hipSuccess && cap != hipStreamCaptureStatusNone) {
      // a captured forward replays with THIS value: a replay on an unfit batch would leave the word raised for every later
      // replay (the fall-back would run for fit batches too) — reset it inside the graph (a kernel, not a memset node: §4)
      hipLaunchKernelGGL(k_zero_word, dim3(1), dim3(64), 0, (hipStream_t)stream, p.local_flag);
      YL_LAUNCH_CHECK();
    }
  }

  // ---- graph structure + node side of layer 0 (fp32 MFMA on the raw K = Cin0 features, bf16 / fp32 outputs)
  char nm[112];
  YL_HSTAGE("graph_prep[csr+attr+segments] + node_uv[layer 0, bf16 out]", 8.0 * N * m->conv[0].Cin * C,
            16.0 * E + 12.0 * E + 32.0 * E + 12.0 * N + 4.0 * N * m->conv[0].Cin + 2.0 * N * 3 * C + 4.0 * N * C, {
    const NodeUv& a = a0;
    if (vouched) {
      if (g == nullptr)
        YL_TRY(yl_local_prep(edge, stride_e, stride_c, bbox_idx, N, E, P, p.seg_ptr, p.node_seg, p.eptr, status, nullptr, true,
                             st));
    } else if (local) {
      if (g == nullptr)
        YL_TRY(yl_graph_prepare_impl(edge, stride_e, stride_c, e_attr, bbox_idx, E, N, P, p.row_ptr, p.perm, p.src, p.dst,
                                     p.attr, p.seg_ptr, p.node_seg, p.work, status, nullptr, primed, stream));
    } else if (g != nullptr && yl_node3_smallk_ok(a)) {
      YL_TRY(yl_node3_smallk(a, st));
    } else if (g != nullptr) {
      const dim3 grid(yl_cdiv(N, 64), 4);
      if (cv0.Cin <= 16) hipLaunchKernelGGL(k_gemm_nt_node3<16>, grid, dim3(256), 0, st, a);
      else hipLaunchKernelGGL(k_gemm_nt_node3<32>, grid, dim3(256), 0, st, a);
      YL_LAUNCH_CHECK();
    } else {
    YL_TRY(yl_graph_prepare_impl(edge, stride_e, stride_c, e_attr, bbox_idx, E, N, P, p.row_ptr, p.perm, p.src, p.dst,
                                 p.attr, p.seg_ptr, p.node_seg, p.work, status, &a, primed, stream));
    }
  });
  if (local) {
    double fl = 0.0;
    for (int l = 0; l < m->n_blocks; ++l) fl += 2.0 * E * (4.0 * C + C * C) + 8.0 * N * m->conv[l].Cin * C;
    // bytes: what the stack has to move — x, ids, e_attr in; feats + the pooled rows out
    YL_HSTAGE("conv_local_bf16[all conv layers + pooling prologue, one launch]", fl,
              4.0 * N * cv0.Cin + 8.0 * E + 16.0 * E + 4.0 * N + 2.0 * N * D + 4.0 * P * (F + 2 * D), {
      YlLocalIn in{};
      in.seg_ptr = p.seg_ptr;
      if (vouched && g == nullptr) {
        in.attr = e_attr; in.edge = edge; in.se = stride_e; in.sc = stride_c; in.eptr = p.eptr;
      } else {
        in.row_ptr = p.row_ptr; in.src = p.src; in.dst = p.dst; in.attr = p.attr;
      }
      in.status = vouched ? status : nullptr;
      YL_TRY(yl_conv_local_bf16(mh, mh->conv_local, x, ldx, in, N, E, P, p.feats, D, p.Z, ZW, p.local_flag, flag_val, st));
    });
    // the fall-back's layer-0 node side (dead unless the flag was raised)
    if (local_mode != 3 && !vouched) YL_TRY(yl_node3_smallk(a0, st, gate));
  }
  const bool no_layers = local && (local_mode == 3 || vouched);
  for (int l = 0; l < m->n_blocks && !no_layers; ++l) {
    const yolat_conv_eval& cv = m->conv[l];
    if (l > 0) {
      snprintf(nm, sizeof nm, "node_uv_bf16[UV | lin_r | mlp_node, N x 64 -> %ld+%ld+%ld]%s", 2 * C, C, C,
               local ? " (gated fall-back)" : "");
      YL_HSTAGE(nm, 8.0 * N * 64 * C, 2.0 * (2.0 * N * 64 + 3.0 * N * C) + 4.0 * N * C, {
      NodeUvH a;
      a.af = HOp{f_slot(l - 1), ld_slot(l - 1), (int)N};
      a.as = HOp{s_slot(l - 1), ld_slot(l - 1), (int)N};
      a.wuv = HOp{mh->Wuv[l], 64, (int)(2 * C)}; a.wr = HOp{mh->Wr[l], 64, (int)C}; a.wn = HOp{mh->Wn[l], 64, (int)C};
      a.euv = plain_epilogue(); a.euv.Yh = p.UV; a.euv.ldy = 2 * C;
      a.euv.scale = mh->uv_scale[l]; a.euv.shift = mh->uv_shift[l];
      a.er = plain_epilogue(); a.er.bias = cv.br; a.er.Y = p.root; a.er.ldy = C;
      a.en = plain_epilogue(); a.en.bias = cv.bn; a.en.scale = cv.sn; a.en.shift = cv.tn; a.en.relu = 1;
      a.en.Yh = s_slot(l); a.en.ldy = ld_slot(l);
      a.N = (int)N; a.C = (int)C; a.Cin = 64;
      const int ntile = yl_cdiv(N, 64);
      if (gate.p) hipLaunchKernelGGL(k_hgemm_node3_gated, dim3(ntile, 4), dim3(256), 0, st, a, gate);
      else hipLaunchKernelGGL(k_hgemm_node3, dim3(ntile, 4), dim3(256), 0, st, a);
      YL_LAUNCH_CHECK();
      });
    }
    snprintf(nm, sizeof nm, "edge_uv_mlp2_mean_bf16[E x (U+V+attr) -> %ld -> %ld -> mean]%s", C, C,
             local ? " (gated fall-back)" : "");
    // bytes: SURVEY.md 8(d) B_agg(l) of the UNFACTORISED layer at 2 bytes per feature element,
    // E ((2 Cin + 4) s + 2 * 4) + N C s — the credit figure; what the kernel has to move is priced in bench.py
    YL_HSTAGE(nm, 2.0 * E * (4.0 * C + C * C), E * ((2.0 * cv.Cin + 4.0) * 2.0 + 8.0) + 2.0 * N * C, {
      YL_TRY(yl_edge_uv_mlp2_mean_bf16_impl(p.UV, 2 * C, p.src, p.dst, p.attr, p.row_ptr, N, E, cv.Wc4, cv.s1, mh->W2[l],
                                            mh->t2f[l], p.root, C, f_slot(l), ld_slot(l), 0, st, gate));
    });
  }

  // ---- pooling prologue, fusion block (+ per-proposal max) | fusion_block_super, classifier
  if (!no_layers)
  YL_HSTAGE(local ? "pool_prepare_bf16[max(feats), mean(fsup), zero] (gated fall-back)"
                  : "pool_prepare_bf16[max(feats), mean(fsup), zero]", 2.0 * N * D, 4.0 * N * D + 4.0 * P * (F + 2 * D), {
  for (int64_t p0 = 0; p0 < P; p0 += 65535) {
    const int64_t np = (P - p0) < 65535 ? (P - p0) : 65535;
    hipLaunchKernelGGL(k_pool_prepare_h, dim3(yl_cdiv(F + 2 * D, 256), (unsigned)(gate.p && np > 1024 ? 1024 : np)), dim3(256), 0,
                       st, p.feats, p.fsup, D, (int)D, (int)F, p.seg_ptr + p0, p.Z + p0 * ZW, ZW, gate, (int)np);
    YL_LAUNCH_CHECK();
  }
  });
  snprintf(nm, sizeof nm, "fusion_gemm_bf16+segmax[N x %ld -> %ld -> P] | super[P x %ld -> %ld]", D, F, D, F);
  YL_HSTAGE(nm, 2.0 * (N + P) * D * F, 2.0 * (N * D + 2.0 * D * F) + 4.0 * (2.0 * P * F + N + P * D), {
    HOp a0{p.feats, D, (int)N}, b0{mh->Wf, D, (int)F}, b1{mh->Wfs, D, (int)F};
    FOp a1{p.Z + 2 * F + D, ZW, (int)P};
    Epilogue e0 = plain_epilogue(), e1 = plain_epilogue();
    e0.bias = m->bf; e0.scale = m->sf; e0.shift = m->tf; e0.relu = 1; e0.seg = p.node_seg; e0.pool = p.Z; e0.ldpool = ZW;
    e1.bias = m->bfs; e1.scale = m->sfs; e1.shift = m->tfs; e1.relu = 1; e1.Y = p.Z + F + D; e1.ldy = ZW;
    const int tm1 = yl_cdiv(P, 64), tn1 = yl_cdiv(F, 64);
    const long n1p = ((long)tm1 * tn1 + 7) & ~7L;
    if ((D == 64 || D == 128) && mh->Wf_fold && mh->Wfs_fold && mh->tf_fold && mh->tfs_fold) {
      // A-in-registers rows kernel on the BatchNorm-folded weights (fusion_h8.hip)
      YL_TRY(yl_hfusion_rows8(p.feats, D, N, D, mh->Wf_fold, mh->tf_fold, p.node_seg, p.Z, ZW, F, p.Z + 2 * F + D, ZW, P,
                              mh->Wfs_fold, mh->tfs_fold, p.Z + F + D, ZW, st));
    } else if (D <= 256) {
      // A-resident rows kernel: at least 4 column groups, and as many as it takes to put >= ~1024 workgroups on
      // the GPU (each group walks tn / groups consecutive 64-column tiles)
      const int tn = yl_cdiv(F, 64);
      // measured at cfg 5 (N = 200k): 64-row tiles x 4 column groups 160 us, x 1 group 166 us, 128-row tiles 181 us
      // (x 1) / 161 us (x 4); cfg 2 (N = 10k): 8 groups 16.1 us, 16 groups 16.8 us, 2 groups 21.2 us
      bool big = false;
      long min_wgs = 1024;
      const int tm0 = yl_cdiv(N, big ? 128 : 64);
      int groups = 4;
      while ((long)tm0 * groups < min_wgs && groups < tn) groups *= 2;
      if (groups > tn) groups = tn;
      const int ng = yl_cdiv(tn, groups);
      groups = yl_cdiv(tn, ng);
      const long total = (long)tm0 * groups + n1p;
      if (total >= (1LL << 31)) return YOLAT_E_UNSUPPORTED;
      auto go = [&](auto kern) {
        hipLaunchKernelGGL(kern, dim3((unsigned)total), dim3(256), 0, st, a0, b0, e0, (int)N, (int)F, (int)D, tm0, groups, ng,
                           a1, b1, e1, (int)P, (int)F, (int)D, tm1, tn1);
      };
      if (D <= 128) { if (big) go(k_hfusion_rows<2, 128>); else go(k_hfusion_rows<1, 128>); }
      else { if (big) go(k_hfusion_rows<2, 256>); else go(k_hfusion_rows<1, 256>); }
    } else {
    const bool big = N >= 65536;                       // 128x128 tiles once there are enough of them to fill the GPU
    const int tm0 = yl_cdiv(N, big ? 128 : 64), tn0 = yl_cdiv(F, big ? 128 : 64);
    const long total = (long)tm0 * tn0 + n1p;
    if (total >= (1LL << 31)) return YOLAT_E_UNSUPPORTED;
    if (big)
      hipLaunchKernelGGL(k_hgemm_two<2>, dim3((unsigned)total), dim3(256), 0, st, a0, b0, e0, (int)N, (int)F, (int)D, tm0,
                         tn0, a1, b1, e1, (int)P, (int)F, (int)D, tm1, tn1);
    else
      hipLaunchKernelGGL(k_hgemm_two<1>, dim3((unsigned)total), dim3(256), 0, st, a0, b0, e0, (int)N, (int)F, (int)D, tm0,
                         tn0, a1, b1, e1, (int)P, (int)F, (int)D, tm1, tn1);
    }
    YL_LAUNCH_CHECK();
  });
  {
    Epilogue e = plain_epilogue();
    // the hidden activations of the classifier cross HBM as bf16 (in the fp32-sized buffers c1 / c2): the next Linear rounds
    // its rows to bf16 while staging anyway (FOp::pack — the same round-to-nearest-even), so the logits are the same bits
    u16* const c1h = reinterpret_cast<u16*>(p.c1);
    u16* const c2h = reinterpret_cast<u16*>(p.c2);
    e.bias = m->bc1; e.scale = m->sc1; e.shift = m->tc1; e.relu = 1; e.Y = nullptr; e.Yh = c1h; e.ldy = m->H1;
    snprintf(nm, sizeof nm, "cls1_bf16[P x %ld -> %ld]", ZW, (long)m->H1);
    YL_HSTAGE(nm, 2.0 * P * ZW * m->H1, 4.0 * P * ZW + 2.0 * P * m->H1 + 2.0 * ZW * m->H1,
              { YL_TRY(launch_hgemm(FOp{p.Z, ZW, (int)P}, HOp{mh->Wc1, ZW, (int)m->H1}, e, P, m->H1, ZW, st)); });
    e.bias = m->bc2; e.scale = m->sc2; e.shift = m->tc2; e.Yh = c2h; e.ldy = m->H2;
    snprintf(nm, sizeof nm, "cls2_bf16[P x %ld -> %ld]", (long)m->H1, (long)m->H2);
    YL_HSTAGE(nm, 2.0 * P * m->H1 * m->H2, 2.0 * (P * m->H1 + P * m->H2) + 2.0 * m->H1 * m->H2,
              { YL_TRY(launch_hgemm(HOp{c1h, m->H1, (int)P}, HOp{mh->Wc2, m->H1, (int)m->H2}, e, P, m->H2, m->H1, st)); });
    e = plain_epilogue();
    e.bias = m->bc3; e.Y = logits; e.ldy = ld_logits;
    snprintf(nm, sizeof nm, "cls3_bf16[P x %ld -> %d]", (long)m->H2, (int)m->n_classes);
    YL_HSTAGE(nm, 2.0 * P * m->H2 * m->n_classes, 2.0 * P * m->H2 + 4.0 * P * m->n_classes + 2.0 * m->H2 * m->n_classes, {
      YL_TRY(launch_hgemm(HOp{c2h, m->H2, (int)P}, HOp{mh->Wc3, m->H2, (int)m->n_classes}, e, P, m->n_classes, m->H2, st));
    });
  }
  return 0;
}

// ------------------------------------------------------------------------------------------------
// Training with bf16 storage: the two [E,64] x [64,64] Linears of the edge MLP on the bf16 matrix cores.
//   yolat_linear_fwd_h     Y (bf16) = pro(A bf16) . W^T + bias, BatchNorm partial statistics from the fp32 accumulators
//   yolat_linear_fwd_wt_h  Y (bf16) = A (bf16) . Wt          (dX = dY . W)
// W / Wt are fp32 parameters: converted (and, for Wt, transposed) into `w_work` ([Nout*K] bf16) on the stream first.
// K % 64 == 0 (the edge MLP: K = 64).
// ------------------------------------------------------------------------------------------------
static __global__ void k_f32_to_bf16_t(const float* __restrict__ Wt, long ldw, int K, int Nout, u16* __restrict__ dst) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;          // dst[n][k] = Wt[k][n]
  if (i >= K * Nout) return;
  const int n = i / K, k = i % K;
  dst[i] = (u16)(yl_pack_bf16(Wt[(long)k * ldw + n], 0.f) & 0xFFFFu);
}
static __global__ void k_f32_to_bf16_rows(const float* __restrict__ W, long ldw, int K, int Nout, u16* __restrict__ dst) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= K * Nout) return;
  dst[i] = (u16)(yl_pack_bf16(W[(long)(i / K) * ldw + (i % K)], 0.f) & 0xFFFFu);
}

// ------------------------------------------------------------------------------------------------
// Many-row 64 -> 64 Linear on bf16-stored rows as a row STREAM (round 4; the bf16 twin of dense.hip's k_lin64_stream):
// the second edge Linear of a bf16-storage training conv layer was one 64 x 64 hgemm tile per workgroup (18 750 at
// E = 1.2 M: 86 us, 3.8 TB/s by counters against the 4.7 TB/s of the fp32 stream).  Persistent workgroups, the bf16 weight
// in LDS once, the next tile's rows in flight under the current tile's work, BatchNorm + ReLU prologue while staging
// (HProOp's arithmetic), statistics from the fp32 accumulators (wave_epilogue's arithmetic), output through LDS with
// 8-byte bf16 stores.  Same MFMAs on the same operands in the same order as hgemm_tile: bit-identical.
// ------------------------------------------------------------------------------------------------
static __global__ void __launch_bounds__(256) k_hlin64_stream(const u16* __restrict__ A, long lda, int M,
                                                              const float* __restrict__ a_scale,
                                                              const float* __restrict__ a_shift, float a_floor,
                                                              const u16* __restrict__ Wb, const float* __restrict__ bias,
                                                              u16* __restrict__ Y, long ldy, float2* __restrict__ stats,
                                                              int tiles_per_wg) {
  constexpr int LDH = YL_HRS, LDO = 68;                    // 72 bf16 = 144 B rows (16-byte fragment reads)
  __shared__ __attribute__((aligned(16))) u16 Ah[64 * LDH], Wh[64 * LDH];
  __shared__ __attribute__((aligned(16))) float Os[64 * LDO];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, l31 = lane & 31, lhi = lane >> 5;
  const int wm = wave >> 1, wn = wave & 1;
  const int q = tid & 15, rb = tid >> 4;                   // staging role: columns 4q.., rows rb + 16 t
  const int ntiles = (M + 63) >> 6;
  const int t0 = blockIdx.x * tiles_per_wg, t1 = yl_min(ntiles, t0 + tiles_per_wg);
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    const int r = rb + 16 * t;
    *reinterpret_cast<uint2*>(&Wh[r * LDH + 4 * q]) = *reinterpret_cast<const uint2*>(Wb + r * 64 + 4 * q);
  }
  float4 as = make_float4(1.f, 1.f, 1.f, 1.f), ah = make_float4(0.f, 0.f, 0.f, 0.f);
  if (a_scale) { as = *reinterpret_cast<const float4*>(a_scale + 4 * q); ah = *reinterpret_cast<const float4*>(a_shift + 4 * q); }
  const float bv = bias ? bias[wn * 32 + l31] : 0.f;
  uint2 ra[4];
  auto fetch = [&](int tile) {
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const long e = yl_min(tile * 64 + rb + 16 * t, M - 1);
      ra[t] = *reinterpret_cast<const uint2*>(A + e * lda + 4 * q);
    }
  };
  if (t0 < t1) fetch(t0);
  for (int tile = t0; tile < t1; ++tile) {
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      uint2 o = ra[t];
      if (a_scale) {                                       // HProOp::load's arithmetic
        o.x = yl_pack_bf16(fmaxf(fmaf(yl_bf16_lo(ra[t].x), as.x, ah.x), a_floor), fmaxf(fmaf(yl_bf16_hi(ra[t].x), as.y, ah.y), a_floor));
        o.y = yl_pack_bf16(fmaxf(fmaf(yl_bf16_lo(ra[t].y), as.z, ah.z), a_floor), fmaxf(fmaf(yl_bf16_hi(ra[t].y), as.w, ah.w), a_floor));
      }
      *reinterpret_cast<uint2*>(&Ah[(rb + 16 * t) * LDH + 4 * q]) = o;
    }
    __syncthreads();                                       // Ah complete (and the previous tile's Os reads are done)
    if (tile + 1 < t1) fetch(tile + 1);                    // in flight under the MFMAs
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      const bf16x8 a = *reinterpret_cast<const bf16x8*>(Ah + (wm * 32 + l31) * LDH + ks * 16 + lhi * 8);
      const bf16x8 b = *reinterpret_cast<const bf16x8*>(Wh + (wn * 32 + l31) * LDH + ks * 16 + lhi * 8);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0);
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] += bv;
    const int row_base = tile * 64 + wm * 32;
    if (stats != nullptr) {                                // wave_epilogue's arithmetic (common.hpp), per 32-row group
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
    __syncthreads();                                       // Os complete; every read of Ah is done
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const int r = rb + 16 * t;
      const long row = (long)tile * 64 + r;
      if (row < M) {
        const float4 v = *reinterpret_cast<const float4*>(Os + r * LDO + 4 * q);
        *reinterpret_cast<uint2*>(Y + row * ldy + 4 * q) = make_uint2(yl_pack_bf16(v.x, v.y), yl_pack_bf16(v.z, v.w));
      }
    }
  }
}

extern "C" int yolat_linear_fwd_h(const uint16_t* A, int64_t lda, int64_t M, int64_t K, const float* a_scale,
                                  const float* a_shift, int a_relu, const float* W, int64_t ldw, const float* bias,
                                  int64_t Nout, uint16_t* Y, int64_t ldy, float* stats, uint16_t* w_work,
                                  yolat_stream_t stream) {
  if (M <= 0 || K <= 0 || Nout <= 0 || !A || !Y || !W || !w_work) return YOLAT_E_INVALID;
  if (M >= (1LL << 31) || lda < K || ldw < K || ldy < Nout || (ldy & 1) || (((uintptr_t)Y) & 3)) return YOLAT_E_INVALID;
  if ((a_scale == nullptr) != (a_shift == nullptr) || (a_relu && !a_scale)) return YOLAT_E_INVALID;
  if (K % 64 != 0 || lda % 8 != 0 || (((uintptr_t)A) & 15) || (((uintptr_t)w_work) & 15) ||
      (a_scale && (!yl_aligned16(a_scale) || !yl_aligned16(a_shift))))
    return YOLAT_E_UNSUPPORTED;
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(k_f32_to_bf16_rows, dim3(yl_cdiv(K * Nout, 256)), dim3(256), 0, st, W, (long)ldw, (int)K, (int)Nout,
                     w_work);
  YL_LAUNCH_CHECK();
  if (K == 64 && Nout == 64 && M >= 65536 && ldy % 4 == 0 && (((uintptr_t)Y) & 7) == 0 && lda % 4 == 0 &&
      (!stats || (((uintptr_t)stats) & 7) == 0)) {
    const int ntiles = (int)yl_cdiv(M, 64);
    const int per = yl_cdiv(ntiles, 1024);                  // 36 KB of LDS: four workgroups per CU
    hipLaunchKernelGGL(k_hlin64_stream, dim3(yl_cdiv(ntiles, per)), dim3(256), 0, st, A, (long)lda, (int)M, a_scale, a_shift,
                       a_relu ? 0.f : -INFINITY, w_work, bias, Y, (long)ldy, reinterpret_cast<float2*>(stats), per);
    YL_LAUNCH_CHECK();
    return 0;
  }
  Epilogue ep = plain_epilogue();
  ep.bias = bias; ep.Yh = Y; ep.ldy = ldy; ep.stats = stats;
  HOp b{w_work, K, (int)Nout};
  if (a_scale != nullptr) {
    HProOp a{A, lda, (int)M, a_scale, a_shift, a_relu ? 0.f : -INFINITY};
    return launch_hgemm(a, b, ep, M, Nout, K, st);
  }
  HOp a{A, lda, (int)M};
  return launch_hgemm(a, b, ep, M, Nout, K, st);
}

extern "C" int yolat_linear_fwd_wt_h(const uint16_t* A, int64_t lda, int64_t M, int64_t K, const float* Wt,
                                     int64_t ldw, int64_t Nout, uint16_t* Y, int64_t ldy, uint16_t* w_work,
                                     yolat_stream_t stream) {
  if (M <= 0 || K <= 0 || Nout <= 0 || !A || !Y || !Wt || !w_work) return YOLAT_E_INVALID;
  if (M >= (1LL << 31) || lda < K || ldw < Nout || ldy < Nout || (ldy & 1) || (((uintptr_t)Y) & 3)) return YOLAT_E_INVALID;
  if (K % 64 != 0 || lda % 8 != 0 || (((uintptr_t)A) & 15) || (((uintptr_t)w_work) & 15)) return YOLAT_E_UNSUPPORTED;
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(k_f32_to_bf16_t, dim3(yl_cdiv(K * Nout, 256)), dim3(256), 0, st, Wt, (long)ldw, (int)K, (int)Nout,
                     w_work);
  YL_LAUNCH_CHECK();
  Epilogue ep = plain_epilogue();
  ep.Yh = Y; ep.ldy = ldy;
  HOp a{A, lda, (int)M}, b{w_work, K, (int)Nout};
  return launch_hgemm(a, b, ep, M, Nout, K, st);
}
