// bn_csr.hip — backward of  out[n] += mean_{e in row n} relu(BN_train(Y[e]))  (the aggregation of
// AttrRelativeEdgeConvGlobalPool2 on top of nn.4 / nn.5, torch_vertex.py:308,324,333-335 + torch_nn.py:58-66) WITHOUT
// materialising either [E,C] gradient (round 2, the fp32 training path).
//
// The gradient w.r.t. the message of edge e is a broadcast:  dM[e] = d_out[dst[e]] / deg[dst[e]]; the gradient w.r.t.
// the BatchNorm input is  dY[e] = scale * (g - c1 - xhat * c2)  with  g = relu'(.) dM[e],  xhat = (Y[e] - mean) invstd,
// (c1, c2) = (sum g, sum g xhat) / E.  round 1 wrote dM, read it twice with Y for the sums and the apply pass, wrote dY
// and read it in the two consumers (dW = dY^T A, dA = dY W): 11 passes over [E,C] per layer.  Here:
//   yolat_bn_csr_bwd_stats   one pass over Y (+ the L2-resident rows of d_out): dgamma, dbeta, (c1, c2)
//   yolat_linear_bwd_w_csr   dW (+)= dY^T . pro(A), db (+)= column sums of dY    — dY formed in the TN GEMM's loader
//   yolat_linear_fwd_wt_csr  dA = dY . W                                          — dY formed in the NT GEMM's loader
// = 5 passes.  The arithmetic per element is the one of k_csr_mean_bwd_v4 + k_bn_bwd_partial_v4 + k_bn_bwd_apply_v4
// (same operations in the same order).
#include "common.hpp"

// dY rows formed on the fly (loader contract of the GEMM tile kernels, common.hpp)
template <class T>
struct BnCsrOpT {
  const float* dout; long ldo;           // [N, C]
  const int* dst; const float* inv_deg;  // [E], [N]
  const T* Y; long ldy;                  // [E, C], fp32 or bfloat16-stored
  const float *mean, *invstd, *scale, *shift, *coef;   // [C] each, coef [2C]
  int C, relu;
  int rows, cols, vec;

  __device__ __forceinline__ float one(float g, float w, float y, float mu, float is, float sc, float sh, float k1,
                                       float k2) const {
    g = g * w;
    if (relu && !(fmaf(y, sc, sh) > 0.f)) g = 0.f;
    return sc * (g - k1 - ((y - mu) * is) * k2);
  }
  template <bool FAST>
  __device__ __forceinline__ void load4(int r, int k, float v[4]) const {
    const int rr = yl_min(r, rows - 1);
    const int n = dst[rr];
    const float w = inv_deg[n];
    if (FAST) {
      const float4 g = *reinterpret_cast<const float4*>(dout + (long)n * ldo + k);
      const float4 y = yl_ld4(Y + (long)rr * ldy + k);
      const float4 mu = *reinterpret_cast<const float4*>(mean + k), is = *reinterpret_cast<const float4*>(invstd + k);
      const float4 sc = *reinterpret_cast<const float4*>(scale + k), sh = *reinterpret_cast<const float4*>(shift + k);
      const float4 k1 = *reinterpret_cast<const float4*>(coef + k), k2 = *reinterpret_cast<const float4*>(coef + C + k);
      v[0] = one(g.x, w, y.x, mu.x, is.x, sc.x, sh.x, k1.x, k2.x);
      v[1] = one(g.y, w, y.y, mu.y, is.y, sc.y, sh.y, k1.y, k2.y);
      v[2] = one(g.z, w, y.z, mu.z, is.z, sc.z, sh.z, k1.z, k2.z);
      v[3] = one(g.w, w, y.w, mu.w, is.w, sc.w, sh.w, k1.w, k2.w);
    } else {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int kc = yl_min(k + j, cols - 1);
        const float t = one(dout[(long)n * ldo + kc], w, yl_ld1(Y + (long)rr * ldy + kc), mean[kc], invstd[kc], scale[kc],
                            shift[kc], coef[kc], coef[C + kc]);
        v[j] = (k + j < cols) ? t : 0.f;
      }
    }
  }
};
typedef BnCsrOpT<float> BnCsrOp;

namespace {
// 640 rows per workgroup: E = 1.2 M is 1875 workgroups = one round of the 2048 that fit (512 rows: 2344 = a full round and
// a seventh of a second one)
#define BCS_ROWS 640
// per 512-row block and column: (sum g, sum g*xhat), g = relu'(.) * d_out[dst] / deg — k_bn_bwd_partial_v4 with the
// gradient gathered instead of read
template <class T>
__global__ void __launch_bounds__(256) k_bn_csr_partial(const float* __restrict__ dout, long ldo,
                                                        const int* __restrict__ dst, const float* __restrict__ inv_deg,
                                                        const T* __restrict__ Y, long ldy, long M, int C,
                                                        const float* __restrict__ mean, const float* __restrict__ invstd,
                                                        const float* __restrict__ scale, const float* __restrict__ shift,
                                                        int relu, float2* part) {
  __shared__ float4 red1[16][16], red2[16][16];
  const int q = threadIdx.x & 15, rg = threadIdx.x >> 4;
  const int c = blockIdx.x * 64 + 4 * q;
  const long r0 = (long)blockIdx.y * BCS_ROWS;
  long r1 = r0 + BCS_ROWS;
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
      float w[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const long rr = (r + 16 * k < r1) ? r + 16 * k : r1 - 1;
        const int n = dst[rr];
        w[k] = inv_deg[n];
        y[k] = yl_ld4(Y + rr * ldy + c);
        g[k] = *reinterpret_cast<const float4*>(dout + (long)n * ldo + c);
      }
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        if (r + 16 * k < r1) {
          acc1(y[k].x, g[k].x * w[k], mu.x, is.x, sc.x, sh.x, s1.x, s2.x);
          acc1(y[k].y, g[k].y * w[k], mu.y, is.y, sc.y, sh.y, s1.y, s2.y);
          acc1(y[k].z, g[k].z * w[k], mu.z, is.z, sc.z, sh.z, s1.z, s2.z);
          acc1(y[k].w, g[k].w * w[k], mu.w, is.w, sc.w, sh.w, s1.w, s2.w);
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

// fp64 sum over the blocks in a fixed order -> dgamma, dbeta, coef = (s1/M, s2/M).  ONE 256-thread workgroup per column:
// thread t takes blocks t, t + 256, ... (eight loads in flight; E = 1.2 M: 1875 blocks = one round), the 256 thread sums
// meet in a fixed tree (shuffles inside a wave, the four wave sums in order).  (One 1024-thread workgroup for 64 columns
// walked 117 blocks per thread in 16 partitions: 16 - 27 us per call, eight calls per cfg-5 step.)
__global__ void __launch_bounds__(256) k_bn_csr_finalize(const float2* part, long nb, long M, int C, float* dgamma,
                                                         float* dbeta, int accumulate, float* coef) {
  __shared__ double wa[4], wb[4];
  const int c = blockIdx.x, t = threadIdx.x;
  double a = 0.0, b = 0.0;
  for (long i = t; i < nb; i += 8 * 256) {
    float2 v[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) v[k] = part[(i + 256 * k < nb ? i + 256 * k : nb - 1) * C + c];
#pragma unroll
    for (int k = 0; k < 8; ++k)
      if (i + 256 * k < nb) { a += (double)v[k].x; b += (double)v[k].y; }
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) { a += __shfl_down(a, off); b += __shfl_down(b, off); }
  if ((t & 63) == 0) { wa[t >> 6] = a; wb[t >> 6] = b; }
  __syncthreads();
  if (t == 0) {
    a = ((wa[0] + wa[1]) + wa[2]) + wa[3];
    b = ((wb[0] + wb[1]) + wb[2]) + wb[3];
    float dg = (float)b, dbt = (float)a;
    if (accumulate) { dg += dgamma[c]; dbt += dbeta[c]; }
    dgamma[c] = dg; dbeta[c] = dbt;
    coef[c] = (float)(a / (double)M);
    coef[C + c] = (float)(b / (double)M);
  }
}

// out (+)= sum over the workgroup slabs in order: element i < 4096 -> dW[i / 64][i % 64], the rest -> db
// 64 elements x 16 slab lanes per workgroup: lane q sums the slabs q, q + 16, ... (ascending, 8 loads in flight), the
// sixteen lane sums are added in lane order — a fixed order, deterministic.  (One thread per element walking all S = 475
// slabs was 60 dependent load rounds: 25 us per call for 8 MB.)
__global__ void __launch_bounds__(1024) k_reduce_slabs(const float* __restrict__ partial, int S, int slab, float* dW,
                                                       long lddw, float* db, int accumulate) {
  __shared__ float part[16][64];
  const int ix = threadIdx.x & 63, q = threadIdx.x >> 6, i = blockIdx.x * 64 + ix;
  float s = 0.f;
  if (i < slab) {
    for (int w = q; w < S; w += 16 * 8) {
      float v[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) v[k] = partial[(long)yl_min(w + 16 * k, S - 1) * slab + i];
#pragma unroll
      for (int k = 0; k < 8; ++k)
        if (w + 16 * k < S) s += v[k];
    }
  }
  part[q][ix] = s;
  __syncthreads();
  if (q != 0 || i >= slab) return;
  s = 0.f;
#pragma unroll
  for (int j = 0; j < 16; ++j) s += part[j][ix];
  if (i < 64 * 64) {
    float* o = dW + (long)(i >> 6) * lddw + (i & 63);
    *o = accumulate ? *o + s : s;
  } else if (db != nullptr) {
    float* o = db + (i - 64 * 64);
    *o = accumulate ? *o + s : s;
  }
}

int check_grad(const yolat_bn_csr_grad* g, int64_t E, int64_t C) {
  if (!g || E <= 0 || C <= 0 || E >= (1LL << 31)) return YOLAT_E_INVALID;
  if (!g->d_out || !g->dst || !g->inv_deg || !g->Y || !g->mean || !g->invstd || !g->scale || !g->shift || !g->coef)
    return YOLAT_E_INVALID;
  if (g->ld_out < C || g->ldy < C) return YOLAT_E_INVALID;
  return 0;
}
template <class T>
BnCsrOpT<T> make_op_t(const yolat_bn_csr_grad* g, int64_t E, int64_t C) {
  BnCsrOpT<T> o;
  o.dout = g->d_out; o.ldo = g->ld_out; o.dst = g->dst; o.inv_deg = g->inv_deg;
  o.Y = reinterpret_cast<const T*>(g->Y); o.ldy = g->ldy;
  o.mean = g->mean; o.invstd = g->invstd; o.scale = g->scale; o.shift = g->shift; o.coef = g->coef;
  o.C = (int)C; o.relu = g->relu; o.rows = (int)E; o.cols = (int)C;
  o.vec = (C % 4 == 0) && (g->ld_out % 4 == 0) && (g->ldy % 4 == 0) && yl_aligned16(g->d_out) &&
          ((uintptr_t)g->Y % (4 * sizeof(T)) == 0) && yl_aligned16(g->mean) && yl_aligned16(g->invstd) &&
          yl_aligned16(g->scale) && yl_aligned16(g->shift) && yl_aligned16(g->coef);
  return o;
}
BnCsrOp make_op(const yolat_bn_csr_grad* g, int64_t E, int64_t C) { return make_op_t<float>(g, E, C); }
}  // namespace

// ------------------------------------------------------------------------------------------------------------------
// Both consumers of dY in ONE kernel (C = K = 64): per 64-row tile the gradient tile dY (formed from Y, the gathered
// d_out rows and the coefficients) and the prologue'd input tile A1 = relu(A * a_scale + a_shift) go to LDS once and
// feed  dA[tile] = dY . W  (32 MFMAs per wave)  and  dW += dY^T . A1  (32 MFMAs per wave, accumulated in registers over
// all the tiles of the persistent workgroup) and the column sums db.  Reads Y and A once, writes dA: 3 passes over
// [E,64] instead of the 4 of the two separate GEMMs, and the element-wise dY arithmetic runs once instead of twice (it
// made the TN GEMM 2x slower: 134 -> 273 us per layer at E = 1.2 M).  Partials [workgroup][64*64 + 64] are summed in a
// fixed order by k_reduce_splits: deterministic.
// ------------------------------------------------------------------------------------------------------------------
#define BCL_WGS 512
__device__ __forceinline__ void bcl_store(float* p, float v) { *p = v; }
__device__ __forceinline__ void bcl_store(yl_bf16_t* p, float v) { *p = (yl_bf16_t)(yl_pack_bf16(v, 0.f) & 0xffffu); }
__device__ __forceinline__ float bcl_round(float v) { return __uint_as_float((yl_pack_bf16(v, 0.f) & 0xffffu) << 16); }
template <class T>
__global__ void __launch_bounds__(256) k_bn_csr_l2_bwd(BnCsrOpT<T> y, const T* __restrict__ A, long lda,
                                                       const float* __restrict__ a_scale, const float* __restrict__ a_shift,
                                                       float a_floor, const float* __restrict__ W, long ldw,
                                                       T* __restrict__ dA, long ldda, int E, int tiles_per_wg,
                                                       float* __restrict__ partial, const float* __restrict__ bn_mean,
                                                       const float* __restrict__ bn_invstd, float2* __restrict__ part1) {
  // part1 != NULL: also the partial sums of the NEXT BatchNorm backward (the one in front of A: dA goes through
  // relu'(pro(A)) and that BatchNorm), per workgroup and column (sum g, sum g*xhat) with g = relu'(.) dA and xhat =
  // (A - bn_mean) bn_invstd — the dA tile is passed through LDS into the staging layout, where the raw A values still
  // sit in registers, and leaves for global memory from there with 16-byte stores
  constexpr int LD = 65;
  __shared__ float Ds[64 * LD], As[64 * LD];
  __shared__ __attribute__((aligned(16))) float Ws[64 * LD];
  __shared__ float dbs[4][64];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, l31 = lane & 31, lhi = lane >> 5;
  const int wm = wave >> 1, wn = wave & 1;
  const int q = tid & 15, rb = tid >> 4;                 // staging role: columns 4q.., rows rb + 16 t
  const int ntiles = (E + 63) >> 6;
  const int t0 = blockIdx.x * tiles_per_wg, t1 = yl_min(ntiles, t0 + tiles_per_wg);
  // W [64 c][64 k] -> LDS as is
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    const int r = rb + 16 * t;
    const float4 w = *reinterpret_cast<const float4*>(W + (long)r * ldw + 4 * q);
    float* d = Ws + r * LD + 4 * q;
    d[0] = w.x; d[1] = w.y; d[2] = w.z; d[3] = w.w;
  }
  const float4 mu = *reinterpret_cast<const float4*>(y.mean + 4 * q), is = *reinterpret_cast<const float4*>(y.invstd + 4 * q);
  const float4 sc = *reinterpret_cast<const float4*>(y.scale + 4 * q), sh = *reinterpret_cast<const float4*>(y.shift + 4 * q);
  const float4 k1 = *reinterpret_cast<const float4*>(y.coef + 4 * q), k2 = *reinterpret_cast<const float4*>(y.coef + 64 + 4 * q);
  float4 as = make_float4(1.f, 1.f, 1.f, 1.f), ah = make_float4(0.f, 0.f, 0.f, 0.f);
  if (a_scale) { as = *reinterpret_cast<const float4*>(a_scale + 4 * q); ah = *reinterpret_cast<const float4*>(a_shift + 4 * q); }
  f32x16 accw;
#pragma unroll
  for (int r = 0; r < 16; ++r) accw[r] = 0.f;
  float dbacc = 0.f;                                     // threads 0..63: column tid of db

  float4 ry[4], rg[4], ra[4], rh[4];
  float rw[4];
  float4 p1 = make_float4(0.f, 0.f, 0.f, 0.f), p2 = p1;   // next BatchNorm's partial sums, columns 4q..4q+3
  float4 bm = p1, bi = p1;
  if (part1 != nullptr) { bm = *reinterpret_cast<const float4*>(bn_mean + 4 * q); bi = *reinterpret_cast<const float4*>(bn_invstd + 4 * q); }
  // the destination ids run one tile AHEAD of the rows: dst[e] -> (inv_deg[n], d_out[n]) is a chain of two dependent
  // global loads (measured neutral here, 276 -> 280 us: this kernel is bound by its 921 MB of mixed read / write traffic
  // at ~3.3 TB/s beside 125 us of fp32 matrix-core time, not by the chain; two tiles of rows in flight with the constants
  // moved to LDS, 252 registers, was 285 us — DESIGN.md Appendix R)
  int nn[4];
  auto fetch_idx = [&](int tile) {
#pragma unroll
    for (int t = 0; t < 4; ++t) nn[t] = y.dst[yl_min(tile * 64 + rb + 16 * t, E - 1)];
  };
  auto fetch = [&](int tile) {
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const int e = yl_min(tile * 64 + rb + 16 * t, E - 1);
      const int n = nn[t];
      rw[t] = y.inv_deg[n];
      ry[t] = yl_ld4(y.Y + (long)e * y.ldy + 4 * q);
      rg[t] = *reinterpret_cast<const float4*>(y.dout + (long)n * y.ldo + 4 * q);
      ra[t] = yl_ld4(A + (long)e * lda + 4 * q);
    }
  };
  if (t0 < t1) {
    fetch_idx(t0);
    fetch(t0);
    if (t0 + 1 < t1) fetch_idx(t0 + 1);
  }
  for (int tile = t0; tile < t1; ++tile) {
    // ---- the two tiles into LDS (rows beyond E: zero — they are in the reduction of dW / db)
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const int r = rb + 16 * t;
      const bool ok = tile * 64 + r < E;
      float* d = Ds + r * LD + 4 * q;
      float* a = As + r * LD + 4 * q;
      d[0] = ok ? y.one(rg[t].x, rw[t], ry[t].x, mu.x, is.x, sc.x, sh.x, k1.x, k2.x) : 0.f;
      d[1] = ok ? y.one(rg[t].y, rw[t], ry[t].y, mu.y, is.y, sc.y, sh.y, k1.y, k2.y) : 0.f;
      d[2] = ok ? y.one(rg[t].z, rw[t], ry[t].z, mu.z, is.z, sc.z, sh.z, k1.z, k2.z) : 0.f;
      d[3] = ok ? y.one(rg[t].w, rw[t], ry[t].w, mu.w, is.w, sc.w, sh.w, k1.w, k2.w) : 0.f;
      a[0] = ok ? fmaxf(fmaf(ra[t].x, as.x, ah.x), a_floor) : 0.f;
      a[1] = ok ? fmaxf(fmaf(ra[t].y, as.y, ah.y), a_floor) : 0.f;
      a[2] = ok ? fmaxf(fmaf(ra[t].z, as.z, ah.z), a_floor) : 0.f;
      a[3] = ok ? fmaxf(fmaf(ra[t].w, as.w, ah.w), a_floor) : 0.f;
      rh[t] = ra[t];                                     // raw A of this tile (ra is refilled by the prefetch)
    }
    __syncthreads();
    if (tile + 1 < t1) {                                 // in flight under the MFMAs
      fetch(tile + 1);
      if (tile + 2 < t1) fetch_idx(tile + 2);
    }
    // ---- dA tile = dY . W   (rows wm*32.., columns wn*32..)
    f32x16 acca;
#pragma unroll
    for (int r = 0; r < 16; ++r) acca[r] = 0.f;
#pragma unroll 8
    for (int c = 0; c < 64; c += 2) {
      const float av = Ds[(wm * 32 + l31) * LD + c + lhi];
      const float bv = Ws[(c + lhi) * LD + wn * 32 + l31];
      acca = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acca, 0, 0, 0);
    }
    // ---- dW += dY^T . A1   (rows = dY columns wm*32.., columns = A1 columns wn*32..)
#pragma unroll 8
    for (int e = 0; e < 64; e += 2) {
      const float av = Ds[(e + lhi) * LD + wm * 32 + l31];
      const float bv = As[(e + lhi) * LD + wn * 32 + l31];
      accw = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, accw, 0, 0, 0);
    }
    // ---- db: wave w sums rows 16w..16w+15 of column `lane`, combined in wave order below
    {
      float s = 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) s += Ds[(wave * 16 + r) * LD + lane];
      dbs[wave][lane] = s;
    }
    if (part1 == nullptr) {
      // ---- store the dA tile (bfloat16 storage: round to nearest even, as yl_st4 does)
      const int col = wn * 32 + l31;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = tile * 64 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhi;
        if (row < E) bcl_store(dA + (long)row * ldda + col, acca[r]);
      }
    }
    __syncthreads();                                     // all reads of Ds / As done; dbs complete
    if (tid < 64) dbacc += ((dbs[0][tid] + dbs[1][tid]) + dbs[2][tid]) + dbs[3][tid];
    if (part1 != nullptr) {
      // ---- dA tile -> LDS (Ds is free now) -> staging layout: next BatchNorm's partial sums + 16-byte stores
      const int col = wn * 32 + l31;
#pragma unroll
      for (int r = 0; r < 16; ++r) Ds[(wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhi) * LD + col] = acca[r];
      __syncthreads();
      auto acc1 = [&](float h, float g, float m, float i, float a, float b, float& t1s, float& t2s) {
        if (a_floor == 0.f && !(fmaf(h, a, b) > 0.f)) g = 0.f;
        t1s += g;
        t2s += g * ((h - m) * i);
      };
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const int r = rb + 16 * t;
        const long row = (long)tile * 64 + r;
        if (row < E) {
          const float* d = Ds + r * LD + 4 * q;
          float4 v = make_float4(d[0], d[1], d[2], d[3]);
          T* o = dA + row * ldda + 4 * q;
          yl_st4(o, v);
          // the statistics take the STORED values (bfloat16 storage: rounded), as the apply pass will read them
          if (sizeof(T) != 4) v = make_float4(bcl_round(d[0]), bcl_round(d[1]), bcl_round(d[2]), bcl_round(d[3]));
          acc1(rh[t].x, v.x, bm.x, bi.x, as.x, ah.x, p1.x, p2.x);
          acc1(rh[t].y, v.y, bm.y, bi.y, as.y, ah.y, p1.y, p2.y);
          acc1(rh[t].z, v.z, bm.z, bi.z, as.z, ah.z, p1.z, p2.z);
          acc1(rh[t].w, v.w, bm.w, bi.w, as.w, ah.w, p1.w, p2.w);
        }
      }
      __syncthreads();                                   // before the next tile's fill overwrites Ds
    }
  }
  if (part1 != nullptr) {
    // the 16 row groups of a column quad, summed in order by row group 0
    float4* red1 = reinterpret_cast<float4*>(Ws);        // Ws (64 * 65 floats) holds 2 x 16 x 16 float4
    float4* red2 = red1 + 256;
    __syncthreads();
    red1[rb * 16 + q] = p1; red2[rb * 16 + q] = p2;
    __syncthreads();
    if (rb == 0) {
      float4 a = red1[q], b = red2[q];
      for (int t = 1; t < 16; ++t) {
        a.x += red1[t * 16 + q].x; a.y += red1[t * 16 + q].y; a.z += red1[t * 16 + q].z; a.w += red1[t * 16 + q].w;
        b.x += red2[t * 16 + q].x; b.y += red2[t * 16 + q].y; b.z += red2[t * 16 + q].z; b.w += red2[t * 16 + q].w;
      }
      float2* o = part1 + (long)blockIdx.x * 64 + 4 * q;
      o[0] = make_float2(a.x, b.x); o[1] = make_float2(a.y, b.y); o[2] = make_float2(a.z, b.z); o[3] = make_float2(a.w, b.w);
    }
  }
  float* P = partial + (long)blockIdx.x * (64 * 64 + 64);
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int c = wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhi;
    P[c * 64 + wn * 32 + l31] = accw[r];
  }
  if (tid < 64) P[64 * 64 + tid] = dbacc;
}

// ------------------------------------------------------------------------------------------------------------------
// The same kernel for bfloat16 STORAGE on the bf16 matrix cores (round 4).  The fp32-MFMA instantiation above spends
// 64 x 64 cycles of v_mfma_f32_32x32x2f32 per tile and wave — 125 us of matrix-core time at E = 1.2 M before any
// staging, which made the bf16-storage step's kernel SLOWER than the fp32 one (288 vs 276 us) although it moves half the
// bytes.  In this mode the operands are bfloat16-stored values anyway and the two [E,64] x [64,64] Linears of the
// forward / dX path already run on v_mfma_f32_32x32x16_bf16 (dense.hip hgemm_tile), so here too: the dY tile (formed in
// fp32 from Y, the gathered d_out rows and the coefficients) and the prologue'd input tile A1 are rounded to bfloat16
// (nearest even) on their way into LDS, W is rounded once per workgroup; accumulation stays fp32.
//   dA[tile] = dY . W        A operand: dY rows (two 8-byte LDS reads), B operand: W^T rows (one 16-byte read)   4 MFMAs
//   dW      += dY^T . A1     both operands are COLUMNS of the row-major tiles: eight 2-byte LDS reads each       4 MFMAs
// (column reads: 64 per tile and wave instead of the 128 4-byte reads of the fp32 kernel; row stride 68 bf16 = 34 banks,
// so the two lane halves, 8 rows apart, land 16 banks apart).  db: per-thread sums of the fp32 dY values over the
// thread's rows, combined over the 16 row groups in fixed order at the end.  Two barriers per tile instead of four.
// ------------------------------------------------------------------------------------------------------------------
typedef __bf16 bcl_bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned bcl_u32x4 __attribute__((ext_vector_type(4)));
__global__ void __launch_bounds__(256, 2) k_bn_csr_l2_bwd_h(BnCsrOpT<yl_bf16_t> y, const yl_bf16_t* __restrict__ A, long lda,
                                                         const float* __restrict__ a_scale, const float* __restrict__ a_shift,
                                                         float a_floor, const float* __restrict__ W, long ldw,
                                                         yl_bf16_t* __restrict__ dA, long ldda, int E, int tiles_per_wg,
                                                         float* __restrict__ partial, const float* __restrict__ bn_mean,
                                                         const float* __restrict__ bn_invstd, float2* __restrict__ part1) {
  constexpr int LDH = 68, LDW = 72, LDO = 68;
  __shared__ __attribute__((aligned(16))) unsigned short Dh[64 * LDH], Ah[64 * LDH], Wt[64 * LDW];
  __shared__ __attribute__((aligned(16))) float Os[64 * LDO];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, l31 = lane & 31, lhi = lane >> 5;
  const int wm = wave >> 1, wn = wave & 1;
  const int q = tid & 15, rb = tid >> 4;                 // staging role: columns 4q.., rows rb + 16 t
  const int ntiles = (E + 63) >> 6;
  const int t0 = blockIdx.x * tiles_per_wg, t1 = yl_min(ntiles, t0 + tiles_per_wg);
  // W [64 c][64 k] -> Wt[k][c] (bf16): the B operand of dA = dY . W wants eight consecutive c per lane
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    const int r = rb + 16 * t;
    const float4 w = *reinterpret_cast<const float4*>(W + (long)r * ldw + 4 * q);
    Wt[(4 * q + 0) * LDW + r] = (unsigned short)(yl_pack_bf16(w.x, 0.f) & 0xffffu);
    Wt[(4 * q + 1) * LDW + r] = (unsigned short)(yl_pack_bf16(w.y, 0.f) & 0xffffu);
    Wt[(4 * q + 2) * LDW + r] = (unsigned short)(yl_pack_bf16(w.z, 0.f) & 0xffffu);
    Wt[(4 * q + 3) * LDW + r] = (unsigned short)(yl_pack_bf16(w.w, 0.f) & 0xffffu);
  }
  __shared__ __attribute__((aligned(16))) float cst[10][64];   // per-column constants, as in k_bn_csr_l2_bwd
  if (tid < 64) {
    cst[0][tid] = y.mean[tid]; cst[1][tid] = y.invstd[tid]; cst[2][tid] = y.scale[tid]; cst[3][tid] = y.shift[tid];
    cst[4][tid] = y.coef[tid]; cst[5][tid] = y.coef[64 + tid];
    cst[6][tid] = a_scale ? a_scale[tid] : 1.f; cst[7][tid] = a_scale ? a_shift[tid] : 0.f;
    cst[8][tid] = part1 ? bn_mean[tid] : 0.f; cst[9][tid] = part1 ? bn_invstd[tid] : 0.f;
  }
  auto cq = [&](int i) { return *reinterpret_cast<const float4*>(&cst[i][4 * q]); };
  auto widen = [](const uint2& u) { return make_float4(yl_bf16_lo(u.x), yl_bf16_hi(u.x), yl_bf16_lo(u.y), yl_bf16_hi(u.y)); };
  f32x16 accw;
#pragma unroll
  for (int r = 0; r < 16; ++r) accw[r] = 0.f;
  float4 dbp = make_float4(0.f, 0.f, 0.f, 0.f);          // column sums of dY over this thread's rows, columns 4q..4q+3
  // two tiles of rows in flight, the bfloat16 ones kept packed (36 registers per set); destination ids a tile further
  struct Rows { uint2 ry[4], ra[4]; float4 rg[4]; float rw[4]; };
  Rows R0, R1;
  uint2 rh[4];
  float4 p1 = make_float4(0.f, 0.f, 0.f, 0.f), p2 = p1;   // next BatchNorm's partial sums, columns 4q..4q+3
  int nn[4];
  auto fetch_idx = [&](int tile) {
#pragma unroll
    for (int t = 0; t < 4; ++t) nn[t] = y.dst[yl_min(tile * 64 + rb + 16 * t, E - 1)];
  };
  auto fetch = [&](int tile, Rows& R) {
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const int e = yl_min(tile * 64 + rb + 16 * t, E - 1);
      const int n = nn[t];
      R.rw[t] = y.inv_deg[n];
      R.ry[t] = *reinterpret_cast<const uint2*>(y.Y + (long)e * y.ldy + 4 * q);
      R.rg[t] = *reinterpret_cast<const float4*>(y.dout + (long)n * y.ldo + 4 * q);
      R.ra[t] = *reinterpret_cast<const uint2*>(A + (long)e * lda + 4 * q);
    }
  };
  auto col8 = [&](const unsigned short* base) {           // eight consecutive rows of one column -> an MFMA operand
    bcl_u32x4 v;
    v[0] = (unsigned)base[0 * LDH] | ((unsigned)base[1 * LDH] << 16);
    v[1] = (unsigned)base[2 * LDH] | ((unsigned)base[3 * LDH] << 16);
    v[2] = (unsigned)base[4 * LDH] | ((unsigned)base[5 * LDH] << 16);
    v[3] = (unsigned)base[6 * LDH] | ((unsigned)base[7 * LDH] << 16);
    return __builtin_bit_cast(bcl_bf16x8, v);
  };
  if (t0 < t1) {
    fetch_idx(t0);
    fetch(t0, R0);
    if (t0 + 1 < t1) {
      fetch_idx(t0 + 1);
      fetch(t0 + 1, R1);
      if (t0 + 2 < t1) fetch_idx(t0 + 2);
    }
  }
  __syncthreads();                                       // cst, Wt
  auto step = [&](int tile, Rows& R) {                   // R: this tile's rows; refilled with tile + 2's
    // ---- the two tiles into LDS as bfloat16 (rows beyond E: zero — they are in the reduction of dW / db)
    {
      const float4 mu = cq(0), is = cq(1), sc = cq(2), sh = cq(3), k1 = cq(4), k2 = cq(5), as = cq(6), ah = cq(7);
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const int r = rb + 16 * t;
        const bool ok = tile * 64 + r < E;
        const float4 yv = widen(R.ry[t]), av = widen(R.ra[t]);
        const float d0 = ok ? y.one(R.rg[t].x, R.rw[t], yv.x, mu.x, is.x, sc.x, sh.x, k1.x, k2.x) : 0.f;
        const float d1 = ok ? y.one(R.rg[t].y, R.rw[t], yv.y, mu.y, is.y, sc.y, sh.y, k1.y, k2.y) : 0.f;
        const float d2 = ok ? y.one(R.rg[t].z, R.rw[t], yv.z, mu.z, is.z, sc.z, sh.z, k1.z, k2.z) : 0.f;
        const float d3 = ok ? y.one(R.rg[t].w, R.rw[t], yv.w, mu.w, is.w, sc.w, sh.w, k1.w, k2.w) : 0.f;
        dbp.x += d0; dbp.y += d1; dbp.z += d2; dbp.w += d3;
        const float a0 = ok ? fmaxf(fmaf(av.x, as.x, ah.x), a_floor) : 0.f;
        const float a1 = ok ? fmaxf(fmaf(av.y, as.y, ah.y), a_floor) : 0.f;
        const float a2 = ok ? fmaxf(fmaf(av.z, as.z, ah.z), a_floor) : 0.f;
        const float a3 = ok ? fmaxf(fmaf(av.w, as.w, ah.w), a_floor) : 0.f;
        *reinterpret_cast<uint2*>(&Dh[r * LDH + 4 * q]) = make_uint2(yl_pack_bf16(d0, d1), yl_pack_bf16(d2, d3));
        *reinterpret_cast<uint2*>(&Ah[r * LDH + 4 * q]) = make_uint2(yl_pack_bf16(a0, a1), yl_pack_bf16(a2, a3));
        rh[t] = R.ra[t];                                 // raw A of this tile (R is refilled by the prefetch)
      }
    }
    __syncthreads();                                     // tiles complete; the previous tile's Os reads are done
    if (tile + 2 < t1) {                                 // two tiles ahead, in flight under the MFMAs
      fetch(tile + 2, R);
      if (tile + 3 < t1) fetch_idx(tile + 3);
    }
    f32x16 acca;
#pragma unroll
    for (int r = 0; r < 16; ++r) acca[r] = 0.f;
    // ---- dA tile = dY . W   (rows wm*32.., columns wn*32..)
    {
      const unsigned short* dr = &Dh[(wm * 32 + l31) * LDH + 8 * lhi];
      const unsigned short* wr = &Wt[(wn * 32 + l31) * LDW + 8 * lhi];
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        const uint2 lo = *reinterpret_cast<const uint2*>(dr + 16 * ks), hi = *reinterpret_cast<const uint2*>(dr + 16 * ks + 4);
        const bcl_u32x4 av = {lo.x, lo.y, hi.x, hi.y};
        const bcl_bf16x8 bv = *reinterpret_cast<const bcl_bf16x8*>(wr + 16 * ks);
        acca = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bcl_bf16x8, av), bv, acca, 0, 0, 0);
      }
    }
    // ---- dW += dY^T . A1   (rows = dY columns wm*32.., columns = A1 columns wn*32..; k = the tile's 64 rows)
    // (not unrolled: with all 64 column reads of the four steps hoisted the kernel needed 293 registers)
#pragma unroll 1
    for (int ks = 0; ks < 4; ++ks) {
      const bcl_bf16x8 av = col8(&Dh[(16 * ks + 8 * lhi) * LDH + wm * 32 + l31]);
      const bcl_bf16x8 bv = col8(&Ah[(16 * ks + 8 * lhi) * LDH + wn * 32 + l31]);
      accw = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, bv, accw, 0, 0, 0);
    }
    // ---- dA tile -> LDS in the staging layout
    {
      const int col = wn * 32 + l31;
#pragma unroll
      for (int r = 0; r < 16; ++r) Os[(wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhi) * LDO + col] = acca[r];
    }
    __syncthreads();                                     // Os complete; every read of Dh / Ah is done
    const float4 as = cq(6), ah = cq(7), bm = cq(8), bi = cq(9);
    auto acc1 = [&](float h, float g, float m, float i, float a, float b, float& t1s, float& t2s) {
      if (a_floor == 0.f && !(fmaf(h, a, b) > 0.f)) g = 0.f;
      t1s += g;
      t2s += g * ((h - m) * i);
    };
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const int r = rb + 16 * t;
      const long row = (long)tile * 64 + r;
      if (row < E) {
        float4 v = *reinterpret_cast<const float4*>(Os + r * LDO + 4 * q);
        yl_st4(dA + row * ldda + 4 * q, v);
        if (part1 != nullptr) {
          // the statistics take the STORED (rounded) values, as the apply pass will read them
          v = make_float4(bcl_round(v.x), bcl_round(v.y), bcl_round(v.z), bcl_round(v.w));
          const float4 hv = widen(rh[t]);
          acc1(hv.x, v.x, bm.x, bi.x, as.x, ah.x, p1.x, p2.x);
          acc1(hv.y, v.y, bm.y, bi.y, as.y, ah.y, p1.y, p2.y);
          acc1(hv.z, v.z, bm.z, bi.z, as.z, ah.z, p1.z, p2.z);
          acc1(hv.w, v.w, bm.w, bi.w, as.w, ah.w, p1.w, p2.w);
        }
      }
    }
  };
  for (int tile = t0; tile < t1; tile += 2) {
    step(tile, R0);
    if (tile + 1 < t1) step(tile + 1, R1);
  }
  // ---- the 16 row groups of a column quad, summed in order by row group 0: next BatchNorm's partials, then db
  float4* red1 = reinterpret_cast<float4*>(Os);          // 64 * 68 floats hold 3 x 16 x 16 float4
  float4* red2 = red1 + 256;
  float4* red3 = red2 + 256;
  __syncthreads();
  red1[rb * 16 + q] = p1; red2[rb * 16 + q] = p2; red3[rb * 16 + q] = dbp;
  __syncthreads();
  float* P = partial + (long)blockIdx.x * (64 * 64 + 64);
  if (rb == 0) {
    float4 a = red1[q], b = red2[q], d = red3[q];
    for (int t = 1; t < 16; ++t) {
      a.x += red1[t * 16 + q].x; a.y += red1[t * 16 + q].y; a.z += red1[t * 16 + q].z; a.w += red1[t * 16 + q].w;
      b.x += red2[t * 16 + q].x; b.y += red2[t * 16 + q].y; b.z += red2[t * 16 + q].z; b.w += red2[t * 16 + q].w;
      d.x += red3[t * 16 + q].x; d.y += red3[t * 16 + q].y; d.z += red3[t * 16 + q].z; d.w += red3[t * 16 + q].w;
    }
    if (part1 != nullptr) {
      float2* o = part1 + (long)blockIdx.x * 64 + 4 * q;
      o[0] = make_float2(a.x, b.x); o[1] = make_float2(a.y, b.y); o[2] = make_float2(a.z, b.z); o[3] = make_float2(a.w, b.w);
    }
    *reinterpret_cast<float4*>(P + 64 * 64 + 4 * q) = d;
  }
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int c = wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhi;
    P[c * 64 + wn * 32 + l31] = accw[r];
  }
}

extern "C" size_t yolat_bn_csr_l2_bwd_work_elems(void) { return (size_t)BCL_WGS * (64 * 64 + 64 + 128) + 8; }

// C = K = Nout = 64 only.  dW [64, 64] (+= when accumulate) = dY^T . pro(A), db [64] (+=) = column sums of dY (nullable),
// dA [E, 64] = dY . W (W = the Linear's weight [64, 64], row-major).  work: yolat_bn_csr_l2_bwd_work_elems() floats.
// next_mean != NULL (then a_scale / a_shift / a_relu describe the BatchNorm + ReLU in front of A): also the statistics
// of THAT BatchNorm's backward on dA — next_dgamma / next_dbeta [64] and next_coef [128] = (c1 | c2), ready for
// yolat_bn_relu_bwd_apply — so that no separate pass over dA and A is needed for them.
extern "C" int yolat_bn_csr_l2_bwd(const yolat_bn_csr_grad* g, int64_t E, const void* A, int64_t lda, const float* a_scale,
                                   const float* a_shift, int a_relu, const float* W, int64_t ldw, float* dW, int64_t lddw,
                                   float* db, int accumulate, void* dA, int64_t ldda, float* work,
                                   const float* next_mean, const float* next_invstd, float* next_dgamma,
                                   float* next_dbeta, float* next_coef, yolat_stream_t stream) {
  const int rc = check_grad(g, E, 64);
  if (rc) return rc;
  if (!A || !W || !dW || !dA || !work || lda < 64 || ldw < 64 || lddw < 64 || ldda < 64) return YOLAT_E_INVALID;
  const bool next = next_mean != nullptr;
  if (next && (!next_invstd || !next_dgamma || !next_dbeta || !next_coef || !a_scale || ldda % 4 != 0 ||
               !yl_aligned16(next_mean) || !yl_aligned16(next_invstd) || ((uintptr_t)dA % (g->half ? 8 : 16)) != 0))
    return YOLAT_E_INVALID;
  if ((a_scale == nullptr) != (a_shift == nullptr) || (a_relu && !a_scale)) return YOLAT_E_INVALID;
  const size_t al = g->half ? 8 : 16;
  if (lda % 4 != 0 || ldw % 4 != 0 || ((uintptr_t)A % al) != 0 || !yl_aligned16(W) ||
      (a_scale && (!yl_aligned16(a_scale) || !yl_aligned16(a_shift))))
    return YOLAT_E_UNSUPPORTED;
  hipStream_t st = (hipStream_t)stream;
  const int ntiles = (int)yl_cdiv(E, 64);
  int wgs = ntiles < BCL_WGS ? ntiles : BCL_WGS;
  const int per = yl_cdiv(ntiles, wgs);
  wgs = yl_cdiv(ntiles, per);
  const float floor = a_relu ? 0.f : -INFINITY;
  float2* part1 = next ? reinterpret_cast<float2*>(work + (size_t)BCL_WGS * (64 * 64 + 64)) : nullptr;
  if (g->half) {
    BnCsrOpT<yl_bf16_t> y = make_op_t<yl_bf16_t>(g, E, 64);
    if (!y.vec) return YOLAT_E_UNSUPPORTED;
    hipLaunchKernelGGL(k_bn_csr_l2_bwd_h, dim3(wgs), dim3(256), 0, st, y, reinterpret_cast<const yl_bf16_t*>(A),
                       (long)lda, a_scale, a_shift, floor, W, (long)ldw, reinterpret_cast<yl_bf16_t*>(dA), (long)ldda, (int)E,
                       per, work, next_mean, next_invstd, part1);
  } else {
    BnCsrOp y = make_op(g, E, 64);
    if (!y.vec) return YOLAT_E_UNSUPPORTED;
    hipLaunchKernelGGL(k_bn_csr_l2_bwd<float>, dim3(wgs), dim3(256), 0, st, y, reinterpret_cast<const float*>(A), (long)lda,
                       a_scale, a_shift, floor, W, (long)ldw, reinterpret_cast<float*>(dA), (long)ldda, (int)E, per, work,
                       next_mean, next_invstd, part1);
  }
  YL_LAUNCH_CHECK();
  // partial layout [wg][64*64 | 64]: reduce the two pieces with the element stride of the slab
  hipLaunchKernelGGL(k_reduce_slabs, dim3(yl_cdiv(64 * 64 + 64, 64)), dim3(1024), 0, st, work, wgs, 64 * 64 + 64, dW,
                     (long)lddw, db, accumulate);
  YL_LAUNCH_CHECK();
  if (next) {
    hipLaunchKernelGGL(k_bn_csr_finalize, dim3(64), dim3(256), 0, st, part1, (long)wgs, (long)E, 64, next_dgamma, next_dbeta,
                       0, next_coef);
    YL_LAUNCH_CHECK();
  }
  return 0;
}

namespace {
__global__ void k_inv_degree(const int* __restrict__ row_ptr, int N, float* __restrict__ inv_deg) {
  const int n = blockIdx.x * blockDim.x + threadIdx.x;
  if (n >= N) return;
  const int deg = row_ptr[n + 1] - row_ptr[n];
  inv_deg[n] = 1.f / (float)(deg > 1 ? deg : 1);       // the factor k_csr_mean_bwd applies
}
}  // namespace

// inv_deg [N] = 1 / max(in-degree, 1) from the CSR row pointers (yolat_bn_csr_grad.inv_deg)
extern "C" int yolat_inv_degree(const int32_t* row_ptr, int64_t N, float* inv_deg, yolat_stream_t stream) {
  if (N <= 0 || N >= (1LL << 31) || !row_ptr || !inv_deg) return YOLAT_E_INVALID;
  hipLaunchKernelGGL(k_inv_degree, dim3(yl_cdiv(N, 256)), dim3(256), 0, (hipStream_t)stream, row_ptr, (int)N, inv_deg);
  YL_LAUNCH_CHECK();
  return 0;
}

extern "C" size_t yolat_bn_csr_work_elems(int64_t E, int64_t C) { return (size_t)(2 * yl_cdiv(E, BCS_ROWS) * C + 4); }

// dgamma / dbeta (+= when accumulate) of the BatchNorm and g->coef = (c1 | c2) [2C] for the two consumers below.
// C % 4 == 0 and 16-byte aligned rows / vectors required.  work: yolat_bn_csr_work_elems(E, C) floats.
extern "C" int yolat_bn_csr_bwd_stats(const yolat_bn_csr_grad* g, int64_t E, int64_t C, float* dgamma, float* dbeta,
                                      int accumulate, float* coef_out, float* work, yolat_stream_t stream) {
  if (!g || E <= 0 || C <= 0 || E >= (1LL << 31) || !dgamma || !dbeta || !coef_out || !work) return YOLAT_E_INVALID;
  if (!g->d_out || !g->dst || !g->inv_deg || !g->Y || !g->mean || !g->invstd || !g->scale || !g->shift)
    return YOLAT_E_INVALID;
  if (g->ld_out < C || g->ldy < C) return YOLAT_E_INVALID;
  if (C % 4 != 0 || g->ld_out % 4 != 0 || g->ldy % 4 != 0 || !yl_aligned16(g->d_out) ||
      ((uintptr_t)g->Y % (g->half ? 8 : 16)) != 0 || !yl_aligned16(g->mean) || !yl_aligned16(g->invstd) ||
      !yl_aligned16(g->scale) || !yl_aligned16(g->shift))
    return YOLAT_E_UNSUPPORTED;
  hipStream_t st = (hipStream_t)stream;
  const long nb = yl_cdiv(E, BCS_ROWS);
  float2* part = reinterpret_cast<float2*>(work);
  if (g->half)
    hipLaunchKernelGGL(k_bn_csr_partial<yl_bf16_t>, dim3(yl_cdiv(C, 64), (unsigned)nb), dim3(256), 0, st, g->d_out,
                       (long)g->ld_out, g->dst, g->inv_deg, reinterpret_cast<const yl_bf16_t*>(g->Y), (long)g->ldy, (long)E,
                       (int)C, g->mean, g->invstd, g->scale, g->shift, g->relu, part);
  else
  hipLaunchKernelGGL(k_bn_csr_partial<float>, dim3(yl_cdiv(C, 64), (unsigned)nb), dim3(256), 0, st, g->d_out, (long)g->ld_out,
                     g->dst, g->inv_deg, reinterpret_cast<const float*>(g->Y), (long)g->ldy, (long)E, (int)C, g->mean, g->invstd,
                     g->scale, g->shift, g->relu, part);
  YL_LAUNCH_CHECK();
  hipLaunchKernelGGL(k_bn_csr_finalize, dim3((unsigned)C), dim3(256), 0, st, part, nb, (long)E, (int)C, dgamma, dbeta,
                     accumulate, coef_out);
  YL_LAUNCH_CHECK();
  return 0;
}

// dW [C, K] (+= when accumulate) = dY^T . pro(A),  db [C] = column sums of dY;  A [E, K] fp32 with the optional
// BatchNorm+ReLU prologue.  partial: yolat_linear_bwd_w_work_elems(E, C, K) floats.
extern "C" int yolat_linear_bwd_w_csr(const yolat_bn_csr_grad* g, int64_t E, int64_t C, const float* A, int64_t lda,
                                      int64_t K, const float* a_scale, const float* a_shift, int a_relu, float* dW,
                                      int64_t lddw, float* db, int accumulate, float* partial, yolat_stream_t stream) {
  const int rc = check_grad(g, E, C);
  if (rc) return rc;
  if (g->half) return YOLAT_E_UNSUPPORTED;             // bfloat16 storage: yolat_bn_csr_l2_bwd only
  if (K <= 0 || !A || !dW || !partial || lda < K || lddw < K) return YOLAT_E_INVALID;
  if ((a_scale == nullptr) != (a_shift == nullptr) || (a_relu && !a_scale)) return YOLAT_E_INVALID;
  hipStream_t st = (hipStream_t)stream;
  TnPlan p = yl_tn_plan(E, C, K);
  BnCsrOp y = make_op(g, E, C);
  float* dbpart = db ? partial + (size_t)p.S * C * K : nullptr;
  dim3 grid(yl_cdiv(C, 64), yl_cdiv(K, 64), p.S);
  if (a_scale != nullptr) {
    DenseProOp a = yl_dense_pro(A, lda, E, K, a_scale, a_shift, a_relu);
    hipLaunchKernelGGL((k_gemm_tn<BnCsrOp, DenseProOp>), grid, dim3(256), 0, st, y, a, partial, dbpart, (int)E, (int)C,
                       (int)K, p.rows_per_split);
  } else {
    DenseOp a = yl_dense(A, lda, E, K);
    hipLaunchKernelGGL((k_gemm_tn<BnCsrOp, DenseOp>), grid, dim3(256), 0, st, y, a, partial, dbpart, (int)E, (int)C, (int)K,
                       p.rows_per_split);
  }
  YL_LAUNCH_CHECK();
  const long elems = C * K;
  yl_reduce_dw_db(st, partial, elems, p.S, dW, (long)lddw, (int)K, dbpart, db, (long)C, accumulate);
  YL_LAUNCH_CHECK();
  return 0;
}

// dA [E, Nout] = dY [E, C] . W,  W given as the Linear's weight [C, Nout] (row-major, ldw)
extern "C" int yolat_linear_fwd_wt_csr(const yolat_bn_csr_grad* g, int64_t E, int64_t C, const float* W, int64_t ldw,
                                       int64_t Nout, float* dA, int64_t ldda, yolat_stream_t stream) {
  const int rc = check_grad(g, E, C);
  if (rc) return rc;
  if (g->half) return YOLAT_E_UNSUPPORTED;
  if (Nout <= 0 || !W || !dA || ldw < Nout || ldda < Nout) return YOLAT_E_INVALID;
  BnCsrOp a = make_op(g, E, C);
  TransOp b;
  b.p = W; b.ld = ldw; b.rows = (int)Nout; b.cols = (int)C; b.vec = 1;
  Epilogue ep;
  ep.bias = nullptr; ep.scale = nullptr; ep.shift = nullptr; ep.relu = 0;
  ep.Y = dA; ep.ldy = ldda; ep.accumulate = 0; ep.stats = nullptr; ep.seg = nullptr; ep.pool = nullptr; ep.ldpool = 0;
  dim3 grid(yl_cdiv(E, 64), yl_cdiv(Nout, 64));
  hipLaunchKernelGGL((k_gemm_nt<64, 64, 32, BnCsrOp, TransOp, true>), grid, dim3(256), 0, (hipStream_t)stream, a, b, ep,
                     (int)E, (int)Nout, (int)C);
  YL_LAUNCH_CHECK();
  return 0;
}
