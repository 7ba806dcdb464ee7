// conv_chain.hip — AttrRelativeEdgeConvGlobalPool2 (eval, BatchNorm folded) as a persistent kernel in
// which every WAVE owns 32 edges end to end and the two edge-MLP GEMMs are chained through registers
// (gcn_lib/sparse/torch_vertex.py:319-337).
//
// Formulation: the MFMA runs "transposed" — matrix rows = output channels (A operand = weights from
// LDS), matrix columns = the wave's 32 edges (B operand):
//     D1[c][e] = sum_k W1[c][k] * F[e][k]            F[e] = [x[dst_e] | x[src_e]-x[dst_e] | attr_e]
//   * B operand of GEMM1: lane (e = lane&31, h = lane>>5) needs F[e][k] for k = k0+h.  The lane simply
//     loads ITS OWN edge's feature row from global memory with 16-byte loads and picks .x/.y (.z/.w) by h —
//     no LDS staging, no barrier.  The 33 float4 slots of a row stream through a small register ring.
//   * A operand: the weights are pre-packed (yolat_conv_pack_weights, once per weight version) in MFMA
//     fragment order, 4 consecutive k-steps per lane = one 16-byte LDS read (ds_read_b128, conflict-free)
//     per 4 MFMAs.  (A per-MFMA ds_read_b32 exposes ~120 cycles of LDS latency per 64-cycle MFMA when
//     there is one wave per SIMD: measured 190 cycles/MFMA.)
//   * C/D layout of v_mfma_f32_32x32x2: lane (col = e, h) holds rows (r&3) + 8*(r>>2) + 4*h.  Those are
//     exactly "k = k0 + 4*h" pairs for the next GEMM, so after BN1+ReLU the 32 accumulator registers ARE
//     the B operands of GEMM2 (D2[c2][e] = sum_c W2[c2][c] * H1[c][e]) — H1 never leaves the registers.
//   * mean aggregation: M (lane = edge) is written once to a wave-private LDS tile and reduced by a third
//     MFMA, agg[32 nodes x 64] += Sel[32 nodes x 32 edges] . M[32 edges x 64], Sel[n][e] = 1/deg(n) iff
//     edge e points to node n.  Wave-private LDS + in-order DS execution = no workgroup barrier in the
//     edge loop; the four waves of a workgroup run de-synchronised on their four SIMDs.
//   * a workgroup owns a 32-node tile (its CSR edge range, 128 edges per step = 4 waves x 32), is
//     persistent (weights are copied to LDS once per workgroup) and combines the four partial
//     accumulators in a fixed order at the end of the tile (deterministic), adding the root term
//     lin_r(x) and computing the node branch mlp_node(xn) with the otherwise idle waves.
#include "common.hpp"

struct ChainArgs {
  const float* x; long ldx;
  const float* xn; long ldxn;
  int N, E;
  const int* row_ptr; const int* src; const int* dst; const float* attr;
  const float *b1, *s1, *t1, *b2, *s2, *t2, *Wr, *br, *Wn, *bn, *sn, *tn;
  const float *W1p, *W2p;         // MFMA-fragment-packed weights (yolat_conv_pack_weights)
  float* f_out; long ldf;
  float* s_out; long lds;
  int ntiles;
};

#define CH_TN 32
#define CH_C 64
#ifndef CH_RING
#define CH_RING 8
#endif

template <int CIN>
struct ChainDims {
  static constexpr int K1 = 2 * CIN + 4;
  static constexpr int K1P = (K1 + 1) & ~1;
  static constexpr int NS1 = K1P / 2;                  // MFMA k-steps of GEMM1
  static constexpr int G1 = (NS1 + 3) / 4;             // groups of 4 steps = one float4 of A operands per lane
  static constexpr int CINP = (CIN + 1) & ~1;
  static constexpr int LDC = CINP + 1;
  static constexpr int F_W1 = 2 * G1 * 256, F_W2 = 2 * 8 * 256, F_WR = CH_C * LDC, F_TAB = 4 * CH_C;
  static constexpr int F_MS = 4 * 32 * 65, F_SEL = 4 * 32 * 33, F_X = CH_TN * LDC;
  static constexpr int TOTAL = F_W1 + F_W2 + 2 * F_WR + F_TAB + F_MS + F_SEL + 2 * F_X + 32 + 64;
};

// ---- weight packing ------------------------------------------------------------------------------------
// W1p[((cb*G1 + g)*64 + lane)*4 + j] = W1[cb*32 + (lane&31)][2*(4g+j) + (lane>>5)]        (0 beyond K1)
// W2p[((cb*8  + g)*64 + lane)*4 + j] = W2[cb*32 + (lane&31)][kc(4g+j) + 4*(lane>>5)],
//        step s = 4g+j <-> (cbi = s/16, r = s%16),  kc = cbi*32 + (r&3) + 8*(r>>2)
__global__ void k_conv_pack(const float* W1, int K1, int G1, const float* W2, float* W1p, float* W2p) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  const int n1 = 2 * G1 * 256;
  if (i < n1) {
    const int j = i & 3, lane = (i >> 2) & 63, g = (i >> 8) % G1, cb = (i >> 8) / G1;
    const int k = 2 * (4 * g + j) + (lane >> 5);
    W1p[i] = (k < K1) ? W1[(long)(cb * 32 + (lane & 31)) * K1 + k] : 0.f;
  } else if (i < n1 + 2 * 8 * 256) {
    const int t = i - n1;
    const int j = t & 3, lane = (t >> 2) & 63, g = (t >> 8) & 7, cb = t >> 11;
    const int s = 4 * g + j, cbi = s >> 4, r = s & 15;
    const int kc = cbi * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
    W2p[t] = W2[(long)(cb * 32 + (lane & 31)) * 64 + kc];
  }
}

extern "C" size_t yolat_conv_pack_elems(int64_t Cin) {
  const int K1P = (int)((2 * Cin + 4 + 1) & ~1), G1 = (K1P / 2 + 3) / 4;
  return (size_t)(2 * G1 * 256 + 2 * 8 * 256);
}

extern "C" int yolat_conv_pack_weights(const float* W1, const float* W2, int64_t Cin, float* packed,
                                       yolat_stream_t stream) {
  if (!W1 || !W2 || !packed || Cin <= 0) return YOLAT_E_INVALID;
  const int K1 = (int)(2 * Cin + 4), K1P = (K1 + 1) & ~1, G1 = (K1P / 2 + 3) / 4;
  const int total = 2 * G1 * 256 + 2 * 8 * 256;
  hipLaunchKernelGGL(k_conv_pack, dim3(yl_cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream, W1, K1, G1, W2,
                     packed, packed + 2 * G1 * 256);
  YL_LAUNCH_CHECK();
  return 0;
}

template <int CIN>
__global__ void __launch_bounds__(256) k_conv_chain(ChainArgs a) {
  using D = ChainDims<CIN>;
  constexpr int NS1 = D::NS1, G1 = D::G1, CINP = D::CINP, LDC = D::LDC;
  constexpr bool VEC = (CIN == 64);
  extern __shared__ __attribute__((aligned(16))) float sm[];
  float* W1s = sm;                     // packed [2][G1][64][4]
  float* W2s = W1s + D::F_W1;          // packed [2][8][64][4]
  float* Wrs = W2s + D::F_W2;          // [64][LDC]
  float* Wns = Wrs + D::F_WR;
  float* tab = Wns + D::F_WR;          // [4][64]: alpha1, beta1, alpha2, beta2  (y = max(acc*alpha + beta, 0))
  float* Ms = tab + D::F_TAB;          // [4 waves][32 edges][65]   (re-used as the combine buffer)
  float* Sel = Ms + D::F_MS;           // [4 waves][32 nodes][33]
  float* Xs = Sel + D::F_SEL;          // [32][LDC]
  float* XNs = Xs + D::F_X;            // [32][LDC]
  float* invd = XNs + D::F_X;          // [32]
  int* rps = reinterpret_cast<int*>(invd + 32);   // [33]

  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int l31 = lane & 31, lhi = lane >> 5;

  // ---- packed weights (straight 16-byte copies) + folded BatchNorm tables -> LDS, once per workgroup
  {
    constexpr int NQ1 = D::F_W1 / 4, NQ2 = D::F_W2 / 4;
    const float4* s1 = reinterpret_cast<const float4*>(a.W1p);
    const float4* s2 = reinterpret_cast<const float4*>(a.W2p);
    float4* d1 = reinterpret_cast<float4*>(W1s);
    float4* d2 = reinterpret_cast<float4*>(W2s);
    for (int base = tid; base < NQ1; base += 256 * 8) {
      float4 v[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = s1[yl_min(base + j * 256, NQ1 - 1)];
#pragma unroll
      for (int j = 0; j < 8; ++j) if (base + j * 256 < NQ1) d1[base + j * 256] = v[j];
    }
    {
      float4 v[NQ2 / 256];
#pragma unroll
      for (int j = 0; j < NQ2 / 256; ++j) v[j] = s2[tid + j * 256];
#pragma unroll
      for (int j = 0; j < NQ2 / 256; ++j) d2[tid + j * 256] = v[j];
    }
  }
  for (int base = tid; base < CH_C * CINP; base += 256 * 4) {
    float vr[4], vn[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int i = yl_min(base + j * 256, CH_C * CINP - 1);
      const long o = (long)(i / CINP) * CIN + yl_min(i % CINP, CIN - 1);
      vr[j] = a.Wr[o]; vn[j] = a.Wn[o];
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int i = base + j * 256;
      if (i < CH_C * CINP) {
        const bool ok = (i % CINP) < CIN;
        Wrs[(i / CINP) * LDC + (i % CINP)] = ok ? vr[j] : 0.f;
        Wns[(i / CINP) * LDC + (i % CINP)] = ok ? vn[j] : 0.f;
      }
    }
  }
  if (tid < CH_C) {
    const float s1 = a.s1[tid], s2 = a.s2[tid];
    tab[tid] = s1;            tab[64 + tid] = fmaf(a.b1[tid], s1, a.t1[tid]);
    tab[128 + tid] = s2;      tab[192 + tid] = fmaf(a.b2[tid], s2, a.t2[tid]);
  }
  const int ocol = l31;       // output column inside a 32-channel block (tile epilogues)

  // ---- per-lane edge features (B operand of GEMM1), loaded straight from global memory.
  // Cin = 64: the 33 float4 "slots" of my edge's row (16 x[dst], 16 x[src]-x[dst], 1 attr) stream through a
  // ring of RING slots; slot k+RING is requested as soon as slot k has been consumed.  The first RING slots
  // of the NEXT step are requested before GEMM2, i.e. they arrive under GEMM2 + aggregation.
  constexpr int RING = CH_RING;
  float4 ra[VEC ? RING : 1], rb[VEC ? RING : 1];        // slot value = ra - rb  (rb only loaded for the diff slots)
  float xd1[VEC ? 1 : CIN], xs1[VEC ? 1 : CIN];
  float4 at4 = make_float4(0.f, 0.f, 0.f, 0.f);
  (void)ra; (void)rb; (void)xd1; (void)xs1;
  const float4* pd = nullptr;                            // my edge's x[dst] / x[src] rows and attr (current step)
  const float4* ps = nullptr;
  const float4* pa = nullptr;
  int e_dl = -1;                                         // destination node of my edge, tile-local (-1: no edge)

#define CH_SET_EDGE(eb_, q1_, n0_)                                                        \
  do {                                                                                    \
    const int q__ = (eb_) + l31;                                                          \
    const int qc__ = yl_min(q__, a.E - 1);                                                \
    const int s__ = a.src[qc__], d__ = a.dst[qc__];                                       \
    e_dl = (q__ < (q1_)) ? d__ - (n0_) : -1;                                              \
    pd = reinterpret_cast<const float4*>(a.x + (long)d__ * a.ldx);                        \
    ps = reinterpret_cast<const float4*>(a.x + (long)s__ * a.ldx);                        \
    pa = reinterpret_cast<const float4*>(a.attr + (long)qc__ * 4);                        \
  } while (0)
// request slot K4 (compile-time) into ring position K4 % RING
#define CH_ISSUE(K4)                                                                      \
  do {                                                                                    \
    if ((K4) < 16) { ra[(K4) % RING] = pd[(K4)]; }                                        \
    else if ((K4) < 32) { ra[(K4) % RING] = ps[(K4) - 16]; rb[(K4) % RING] = pd[(K4) - 16]; } \
    else { ra[(K4) % RING] = pa[0]; }                                                     \
  } while (0)
#define CH_LOAD_SMALL()                                                                   \
  do {                                                                                    \
    at4 = pa[0];                                                                          \
    _Pragma("unroll") for (int k = 0; k < CIN; ++k) {                                     \
      xd1[k] = reinterpret_cast<const float*>(pd)[k];                                     \
      xs1[k] = reinterpret_cast<const float*>(ps)[k];                                     \
    }                                                                                     \
  } while (0)
#define CH_PREFETCH()                                                                     \
  do {                                                                                    \
    if constexpr (VEC) {                                                                  \
      _Pragma("unroll") for (int k4 = 0; k4 < RING; ++k4) CH_ISSUE(k4);                   \
    } else {                                                                              \
      CH_LOAD_SMALL();                                                                    \
    }                                                                                     \
  } while (0)

  int tile = blockIdx.x;
  int q0 = 0, q1 = 0;
  if (tile < a.ntiles) {
    q0 = a.row_ptr[tile * CH_TN];
    q1 = a.row_ptr[yl_min(tile * CH_TN + CH_TN, a.N)];
  }
  bool have = false;                                    // first slots of (this tile, first step) already requested
  if (q0 + wave * 32 < q1) { CH_SET_EDGE(q0 + wave * 32, q1, tile * CH_TN); CH_PREFETCH(); have = true; }
  __syncthreads();                                      // weights / tables visible

  const float4* w1q = reinterpret_cast<const float4*>(W1s) + lane;    // + (cb*G1 + g)*64
  const float4* w2q = reinterpret_cast<const float4*>(W2s) + lane;    // + (cb*8 + g)*64

  for (; tile < a.ntiles; tile += gridDim.x) {
    const int n0 = tile * CH_TN;
    const int tnext = tile + gridDim.x;
    int q0n = 0, q1n = 0;
    if (tnext < a.ntiles) {
      q0n = a.row_ptr[tnext * CH_TN];
      q1n = a.row_ptr[yl_min(tnext * CH_TN + CH_TN, a.N)];
    }
    if (tid <= CH_TN) rps[tid] = a.row_ptr[yl_min(n0 + tid, a.N)];
    if (tid < CH_TN) {
      const int nn = yl_min(n0 + tid, a.N - 1);
      const int dg = a.row_ptr[nn + 1] - a.row_ptr[nn];
      invd[tid] = 1.f / (float)(dg > 1 ? dg : 1);
    }
    for (int base = tid; base < CH_TN * CINP; base += 256 * 4) {
      float vx[4], vn[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int i = yl_min(base + j * 256, CH_TN * CINP - 1);
        const int n = yl_min(n0 + i / CINP, a.N - 1), kc = yl_min(i % CINP, CIN - 1);
        vx[j] = a.x[(long)n * a.ldx + kc];
        vn[j] = a.xn[(long)n * a.ldxn + kc];
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int i = base + j * 256;
        if (i < CH_TN * CINP) {
          const bool ok = (i % CINP) < CIN;
          Xs[(i / CINP) * LDC + (i % CINP)] = ok ? vx[j] : 0.f;
          XNs[(i / CINP) * LDC + (i % CINP)] = ok ? vn[j] : 0.f;
        }
      }
    }
    __syncthreads();                                    // invd / Xs / rps published

    f32x16 tacc0, tacc1;                                // [32 nodes][64 ch] partial of this wave
#pragma unroll
    for (int r = 0; r < 16; ++r) { tacc0[r] = 0.f; tacc1[r] = 0.f; }

    float* Msw = Ms + wave * 32 * 65;
    float* Selw = Sel + wave * 32 * 33;
    for (int eb = q0 + wave * 32; eb < q1; eb += 128) {
      if (!have) { CH_SET_EDGE(eb, q1, n0); CH_PREFETCH(); }
      have = false;
      const int my_dl = e_dl;
      // ---- GEMM1: D1[64 ch][32 edges], A = packed W1 (one float4 = 4 k-steps), B = my edge's features
      f32x16 c0, c1;
#pragma unroll
      for (int r = 0; r < 16; ++r) { c0[r] = 0.f; c1[r] = 0.f; }
      if constexpr (VEC) {
        float4 wa = w1q[0], wb = w1q[G1 * 64];            // group 0 of channel block 0 / 1
#pragma unroll
        for (int g = 0; g < G1; ++g) {
          float4 na = wa, nb = wb;
          if (g + 1 < G1) { na = w1q[(g + 1) * 64]; nb = w1q[(G1 + g + 1) * 64]; }   // next group in flight
          // slots 2g and 2g+1 (the last group holds only the attr slot)
          {
            float4 v = ra[(2 * g) % RING];
            if (2 * g >= 16 && 2 * g < 32) { const float4 w = rb[(2 * g) % RING]; v.x -= w.x; v.y -= w.y; v.z -= w.z; v.w -= w.w; }
            if (2 * g + RING < 33) CH_ISSUE(2 * g + RING);
            const float bA = lhi ? v.y : v.x, bB = lhi ? v.w : v.z;
            c0 = __builtin_amdgcn_mfma_f32_32x32x2f32(wa.x, bA, c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f32_32x32x2f32(wb.x, bA, c1, 0, 0, 0);
            c0 = __builtin_amdgcn_mfma_f32_32x32x2f32(wa.y, bB, c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f32_32x32x2f32(wb.y, bB, c1, 0, 0, 0);
          }
          if (2 * g + 1 < 33) {
            float4 v = ra[(2 * g + 1) % RING];
            if (2 * g + 1 >= 16 && 2 * g + 1 < 32) { const float4 w = rb[(2 * g + 1) % RING]; v.x -= w.x; v.y -= w.y; v.z -= w.z; v.w -= w.w; }
            if (2 * g + 1 + RING < 33) CH_ISSUE(2 * g + 1 + RING);
            const float bA = lhi ? v.y : v.x, bB = lhi ? v.w : v.z;
            c0 = __builtin_amdgcn_mfma_f32_32x32x2f32(wa.z, bA, c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f32_32x32x2f32(wb.z, bA, c1, 0, 0, 0);
            c0 = __builtin_amdgcn_mfma_f32_32x32x2f32(wa.w, bB, c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f32_32x32x2f32(wb.w, bB, c1, 0, 0, 0);
          }
          wa = na; wb = nb;
        }
      } else {
        float f[4 * G1 * 2];
#pragma unroll
        for (int k = 0; k < 4 * G1 * 2; ++k) f[k] = 0.f;
#pragma unroll
        for (int k = 0; k < CIN; ++k) { f[k] = xd1[k]; f[CIN + k] = xs1[k] - xd1[k]; }
        f[2 * CIN] = at4.x; f[2 * CIN + 1] = at4.y; f[2 * CIN + 2] = at4.z; f[2 * CIN + 3] = at4.w;
#pragma unroll
        for (int g = 0; g < G1; ++g) {
          const float4 wa = w1q[g * 64], wb = w1q[(G1 + g) * 64];
          const float wav[4] = {wa.x, wa.y, wa.z, wa.w}, wbv[4] = {wb.x, wb.y, wb.z, wb.w};
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            if (4 * g + j < NS1) {
              const float b = lhi ? f[2 * (4 * g + j) + 1] : f[2 * (4 * g + j)];
              c0 = __builtin_amdgcn_mfma_f32_32x32x2f32(wav[j], b, c0, 0, 0, 0);
              c1 = __builtin_amdgcn_mfma_f32_32x32x2f32(wbv[j], b, c1, 0, 0, 0);
            }
          }
        }
      }
      // feature registers are dead: request the next step's (or the next tile's first) features now
      {
        const int ebn = eb + 128;
        if (ebn < q1) { CH_SET_EDGE(ebn, q1, n0); CH_PREFETCH(); have = true; }
        else if (q0n + wave * 32 < q1n) { CH_SET_EDGE(q0n + wave * 32, q1n, tnext * CH_TN); CH_PREFETCH(); have = true; }
      }
      // ---- BN1 + ReLU in registers: h[r] is channel (r&3)+8*(r>>2)+4*lhi (+32 for c1) of my edge
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int ch = (r & 3) + 8 * (r >> 2) + 4 * lhi;
        c0[r] = fmaxf(fmaf(c0[r], tab[ch], tab[64 + ch]), 0.f);
        c1[r] = fmaxf(fmaf(c1[r], tab[32 + ch], tab[64 + 32 + ch]), 0.f);
      }
      // ---- GEMM2 chained through registers: step s = (cbi, r) contracts channels {kc, kc+4}; packed W2:
      // one float4 per lane = steps 4g..4g+3
      f32x16 d0, d1;
#pragma unroll
      for (int r = 0; r < 16; ++r) { d0[r] = 0.f; d1[r] = 0.f; }
      {
        float4 wa = w2q[0], wb = w2q[8 * 64];
#pragma unroll
        for (int g = 0; g < 8; ++g) {
          float4 na = wa, nb = wb;
          if (g + 1 < 8) { na = w2q[(g + 1) * 64]; nb = w2q[(8 + g + 1) * 64]; }
          const float wav[4] = {wa.x, wa.y, wa.z, wa.w}, wbv[4] = {wb.x, wb.y, wb.z, wb.w};
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const int s = 4 * g + j;
            const float b = (s < 16) ? c0[s & 15] : c1[s & 15];
            d0 = __builtin_amdgcn_mfma_f32_32x32x2f32(wav[j], b, d0, 0, 0, 0);
            d1 = __builtin_amdgcn_mfma_f32_32x32x2f32(wbv[j], b, d1, 0, 0, 0);
          }
          wa = na; wb = nb;
        }
      }
      // ---- BN2 + ReLU, messages of my edge -> wave-private LDS tile Msw[edge][channel]
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int ch = (r & 3) + 8 * (r >> 2) + 4 * lhi;
        Msw[l31 * 65 + ch] = fmaxf(fmaf(d0[r], tab[128 + ch], tab[192 + ch]), 0.f);
        Msw[l31 * 65 + 32 + ch] = fmaxf(fmaf(d1[r], tab[128 + 32 + ch], tab[192 + 32 + ch]), 0.f);
      }
      // ---- aggregation operator of my 32 edges: Selw[n][e] = 1/deg(n) iff edge e -> node n
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        const int n = lhi * 16 + j;
        Selw[n * 33 + l31] = (my_dl == n) ? invd[n] : 0.f;
      }
      // wave-private LDS: DS ops of one wave execute in order; only make sure the compiler keeps order
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      {
        const float* ap = Selw + l31 * 33 + lhi;                  // A[i = node][k = edge]
        const float* bp = Msw + lhi * 65 + l31;                   // B[k = edge][j = channel]
#pragma unroll
        for (int k8 = 0; k8 < 32; k8 += 8) {                      // 4 k-steps: 12 LDS reads in flight, then 8 MFMAs
          float av[4], b0v[4], b1v[4];
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            av[j] = ap[k8 + 2 * j];
            b0v[j] = bp[(k8 + 2 * j) * 65];
            b1v[j] = bp[(k8 + 2 * j) * 65 + 32];
          }
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            tacc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(av[j], b0v[j], tacc0, 0, 0, 0);
            tacc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(av[j], b1v[j], tacc1, 0, 0, 0);
          }
        }
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
    }

    // ---- tile epilogue.  wave 2 / wave 3 add the root term for channel block 0 / 1 into their partial,
    // wave 1 computes the node branch, waves 1..3 park their partials, wave 0 sums them in fixed order.
    if (wave >= 2) {
      const int cb = wave - 2;
      const float* ap = Xs + l31 * LDC + lhi;
      const float* bp = Wrs + (cb * 32 + l31) * LDC + lhi;
      if (cb == 0) {
#pragma unroll 4
        for (int kk = 0; kk < CINP; kk += 2) tacc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(ap[kk], bp[kk], tacc0, 0, 0, 0);
      } else {
#pragma unroll 4
        for (int kk = 0; kk < CINP; kk += 2) tacc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(ap[kk], bp[kk], tacc1, 0, 0, 0);
      }
    } else if (wave == 1) {
      f32x16 n0a, n1a;
#pragma unroll
      for (int r = 0; r < 16; ++r) { n0a[r] = 0.f; n1a[r] = 0.f; }
      const float* ap = XNs + l31 * LDC + lhi;
      const float* b0 = Wns + l31 * LDC + lhi;
      const float* b1 = Wns + (32 + l31) * LDC + lhi;
#pragma unroll 4
      for (int kk = 0; kk < CINP; kk += 2) {
        const float av = ap[kk];
        n0a = __builtin_amdgcn_mfma_f32_32x32x2f32(av, b0[kk], n0a, 0, 0, 0);
        n1a = __builtin_amdgcn_mfma_f32_32x32x2f32(av, b1[kk], n1a, 0, 0, 0);
      }
      const float bn0 = a.bn[ocol], sn0 = a.sn[ocol], tn0 = a.tn[ocol];
      const float bn1 = a.bn[32 + ocol], sn1 = a.sn[32 + ocol], tn1 = a.tn[32 + ocol];
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = (r & 3) + 8 * (r >> 2) + 4 * lhi;
        if (n0 + row < a.N) {
          float* o = a.s_out + (long)(n0 + row) * a.lds;
          o[ocol] = fmaxf(fmaf(n0a[r] + bn0, sn0, tn0), 0.f);
          o[32 + ocol] = fmaxf(fmaf(n1a[r] + bn1, sn1, tn1), 0.f);
        }
      }
    }
    // each wave parks its partial in ITS OWN (now idle) message tile — another wave's tile may still be
    // in use, the waves are not synchronised inside the edge loop
    float* Cmb = Ms;                                    // [4][32][65], slot 0 unused
    if (wave > 0) {
      float* c = Cmb + wave * 32 * 65;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = (r & 3) + 8 * (r >> 2) + 4 * lhi;
        c[row * 65 + ocol] = tacc0[r];
        c[row * 65 + 32 + ocol] = tacc1[r];
      }
    }
    __syncthreads();
    if (wave == 0) {
      const float br0 = a.br[ocol], br1 = a.br[32 + ocol];
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = (r & 3) + 8 * (r >> 2) + 4 * lhi;
        float v0s = tacc0[r], v1s = tacc1[r];
#pragma unroll
        for (int w = 1; w < 4; ++w) {
          v0s += Cmb[(w * 32 + row) * 65 + ocol];
          v1s += Cmb[(w * 32 + row) * 65 + 32 + ocol];
        }
        if (n0 + row < a.N) {
          float* o = a.f_out + (long)(n0 + row) * a.ldf;
          o[ocol] = v0s + br0;
          o[32 + ocol] = v1s + br1;
        }
      }
    }
    __syncthreads();                                    // Cmb / Xs / rps reused by the next tile
    q0 = q0n; q1 = q1n;
  }
}

template <int CIN>
static int launch_chain(const ChainArgs& a, hipStream_t st) {
  const size_t lds_bytes = (size_t)ChainDims<CIN>::TOTAL * sizeof(float);
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k_conv_chain<CIN>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) return (int)e;
    attr_set = true;
  }
  const int grid = a.ntiles < 256 ? a.ntiles : 256;
  hipLaunchKernelGGL(k_conv_chain<CIN>, dim3(grid), dim3(256), lds_bytes, st, a);
  YL_LAUNCH_CHECK();
  return 0;
}

// Same contract as yolat_conv_eval_fused (conv_fused.hip) plus `packed` = the buffer filled by
// yolat_conv_pack_weights(w->W1, w->W2, Cin, packed) (yolat_conv_pack_elems(Cin) floats, 16-B aligned).
// Supports Cin in {5, 6, 64}.
extern "C" int yolat_conv_eval_chain(const float* x, int64_t ldx, const float* xn, int64_t ldxn, int64_t N,
                                     int64_t Cin, const int32_t* row_ptr, const int32_t* src_csr,
                                     const int32_t* dst_csr, const float* attr_csr, int64_t E,
                                     const yolat_conv_eval* w, const float* packed, int64_t C, float* f_out,
                                     int64_t ldf, float* s_out, int64_t lds, yolat_stream_t stream) {
  if (!x || !xn || !row_ptr || !w || !packed || !f_out || !s_out || N <= 0 || E < 0 || Cin <= 0)
    return YOLAT_E_INVALID;
  if (C != CH_C || !(Cin == 64 || Cin == 5 || Cin == 6)) return YOLAT_E_UNSUPPORTED;
  if (E > 0 && (!src_csr || !dst_csr || !attr_csr)) return YOLAT_E_INVALID;
  if (ldx < Cin || ldxn < Cin || ldf < C || lds < C || N >= (1LL << 31) - 64) return YOLAT_E_INVALID;
  if ((E > 0 && !yl_aligned16(attr_csr)) || !yl_aligned16(packed)) return YOLAT_E_UNSUPPORTED;
  if (Cin == 64 && (ldx % 4 != 0 || !yl_aligned16(x))) return YOLAT_E_UNSUPPORTED;
  ChainArgs a;
  a.x = x; a.ldx = ldx; a.xn = xn; a.ldxn = ldxn; a.N = (int)N; a.E = (int)E;
  a.row_ptr = row_ptr; a.src = src_csr; a.dst = dst_csr; a.attr = attr_csr;
  a.b1 = w->b1; a.s1 = w->s1; a.t1 = w->t1; a.b2 = w->b2; a.s2 = w->s2; a.t2 = w->t2;
  a.Wr = w->Wr; a.br = w->br; a.Wn = w->Wn; a.bn = w->bn; a.sn = w->sn; a.tn = w->tn;
  const int K1P = (int)((2 * Cin + 4 + 1) & ~1), G1 = (K1P / 2 + 3) / 4;
  a.W1p = packed; a.W2p = packed + 2 * G1 * 256;
  a.f_out = f_out; a.ldf = ldf; a.s_out = s_out; a.lds = lds;
  a.ntiles = yl_cdiv(N, CH_TN);
  hipStream_t st = (hipStream_t)stream;
  if (Cin == 64) return launch_chain<64>(a, st);
  if (Cin == 5) return launch_chain<5>(a, st);
  return launch_chain<6>(a, st);
}
