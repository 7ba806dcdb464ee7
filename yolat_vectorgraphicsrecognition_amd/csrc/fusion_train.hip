// fusion_train.hip — training-mode fusion block + per-proposal max pooling WITHOUT the [N, F] activation.
//
// Reference ops (cad_recognition/architecture3cc_rpn_gp_iter2.py:61-63,122):
//     z      = fusion_block.Linear(feats)            [N, F=1024]   (feats [N, K=128])
//     y      = relu(BatchNorm1d(z))                  batch statistics over the N rows
//     pooled = scatter(y, bbox_idx, reduce='max')    [P, F]
// and their autograd.  The straightforward schedule materialises z (715 MB at N = 175k), reads it back for
// the statistics, the pooling, and three times in the backward, and runs two N x F x K GEMMs in the backward.
//
// Restructuring (exact algebra, different rounding):
//  * Linear is affine, so the batch statistics of z follow from those of feats:
//        mean_z = W.mean_A + b,    var_z[c] = w_c G w_c^T / N,    G = (A - mean_A)^T (A - mean_A)   [K, K]
//  * BatchNorm+ReLU is monotone per column (increasing if gamma*rstd >= 0, else decreasing), so
//        max_r relu(s*z_r + t) = relu(s * ext_r z_r + t),   ext = max if s >= 0 else min:
//    the GEMM epilogue keeps only the per-(proposal, column) extreme of z and its row (64-bit atomicMax on
//    an order-preserving key), nothing of shape [N, F] is written.
//  * Backward: the gradient w.r.t. y is non-zero only at the P*F arg rows.  With g = dL/dpooled masked by the
//    ReLU, xhat = (z - mean_z)*rstd = rstd * (A - mean_A).w_c :
//        dbeta = sum_p g,   dgamma = sum_p g * xhat[arg]
//        dz[r,c] = s_c * (dy[r,c] - dbeta_c/N - xhat[r,c]*dgamma_c/N)
//        dW[c,:] = s_c * sum_p g[p,c] * A[arg[p,c],:]  -  (s_c dbeta_c/N) * sum_r A[r,:]  -  (s_c rstd_c dgamma_c/N) * (w_c G)
//        dA[r,:] = sum_{c: arg[p(r),c]=r} s_c g[p,c] w_c  -  sum_c (s_c dbeta_c/N) w_c  -  (A[r]-mean_A) . (W^T diag(s rstd dgamma/N) W)
//        db = 0 (BatchNorm removes the Linear bias from the loss)
//    i.e. two sparse P*F-term kernels plus K x K / F x K sized dense algebra.
#include "common.hpp"
#include <stdlib.h>

#define YL_TRY(call)            \
  do {                          \
    int rc__ = (call);          \
    if (rc__ != 0) return rc__; \
  } while (0)

constexpr int CS_ROWS = 512;    // rows per column-sum workgroup
constexpr int FB_GRAM_S = 512;  // row slabs (= partial products) of the centered Gram matrix
constexpr int FB_PROWS = 32;    // proposals per column-partial workgroup (128: one workgroup per CU at P = 8000, 12 KB of loads
                                // in flight per CU — 64 us for 131 MB)
// partial area of the Gram stage: the TN GEMM's plan or FB_GRAM_S slabs of K x K
static inline size_t fus_gram_elems(int64_t N, int64_t K) {
  const size_t a = yolat_linear_bwd_w_work_elems(N, K, K), b = (size_t)FB_GRAM_S * K * K;
  return a > b ? a : b;
}
constexpr int FB_NG = 64;       // row-block groups (= partial slabs) of the sparse weight gradient

// partial[rb][k] = sum of A[r][k] over the rb-th block of CS_ROWS rows; 64 columns x 4 row lanes per WG
static __global__ void __launch_bounds__(256) k_colsum_partial(const float* __restrict__ A, long lda, long N, int K,
                                                        float* partial) {
  __shared__ float red[4][64];
  const int cl = threadIdx.x & 63, rl = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + cl;
  const long r0 = (long)blockIdx.y * CS_ROWS;
  const long r1 = (r0 + CS_ROWS < N) ? r0 + CS_ROWS : N;
  float s = 0.f;
  if (c < K) {
    long r = r0 + rl;
    for (; r + 12 < r1; r += 16) {     // 4 independent loads in flight, summed in row order
      const float v0 = A[r * lda + c], v1 = A[(r + 4) * lda + c], v2 = A[(r + 8) * lda + c],
                  v3 = A[(r + 12) * lda + c];
      s += v0; s += v1; s += v2; s += v3;
    }
    for (; r < r1; r += 4) s += A[r * lda + c];
  }
  red[rl][cl] = s;
  __syncthreads();
  if (rl == 0 && c < K) partial[(long)blockIdx.y * K + c] = (red[0][cl] + red[1][cl]) + (red[2][cl] + red[3][cl]);
}

// sumA -> (ones, -mean) prologue vectors for the centered loaders
static __global__ void k_center_vecs(const float* sumA, int K, float invN, float* ones, float* negmean) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= K) return;
  ones[k] = 1.f;
  negmean[k] = -(sumA[k] * invN);
}

// BatchNorm batch statistics of z = A W^T + b from the moments of A; same outputs as k_bn_finalize_l2.
static __global__ void k_fusion_bn_stats(const float* __restrict__ W, const float* __restrict__ T, const float* bias,
                                  const float* sumA, int K, int F, double N, const float* gamma, const float* beta,
                                  float* running_mean, float* running_var, float momentum, float eps,
                                  float* scale, float* shift, float* save_mean, float* save_invstd) {
  // 16 lanes per column: lane j takes k = j, j + 16, ... (coalesced 64-byte pieces of the column's rows of W and T), the
  // sixteen fp64 partial sums meet in a fixed butterfly.  (One thread per column walked 128 strided element pairs: 19.5 us
  // of latency for 1 MB.)
  const int gt = blockIdx.x * blockDim.x + threadIdx.x, c = gt >> 4, j = gt & 15;
  const int cc = c < F ? c : F - 1;
  const float* w = W + (long)cc * K;
  const float* t = T + (long)cc * K;
  double m = 0.0, q = 0.0;
  for (int k = j; k < K; k += 16) {
    m += (double)w[k] * (double)sumA[k];
    q += (double)w[k] * (double)t[k];
  }
#pragma unroll
  for (int off = 8; off >= 1; off >>= 1) {
    m += __shfl_xor(m, off);
    q += __shfl_xor(q, off);
  }
  if (c >= F || j != 0) return;
  const double mean = m / N + (bias ? (double)bias[c] : 0.0);
  double var_b = q / N;
  if (var_b < 0.0) var_b = 0.0;
  const double var_u = N > 1.0 ? var_b * N / (N - 1.0) : var_b;
  const float invstd = (float)(1.0 / sqrt(var_b + (double)eps));
  const float meanf = (float)mean;
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

// key -> (z*, arg); pooled = relu(fma(z*, scale, shift)) exactly like the eval epilogue
static __global__ void __launch_bounds__(256) k_pool_finish(const unsigned long long* __restrict__ key, long P, int F,
                                                     const float* scale, const float* shift, int N, float* Z,
                                                     long ldz, float* zstar, int* arg) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  const long p = blockIdx.y;
  if (c >= F) return;
  const unsigned long long k = key[p * F + c];
  float z = 0.f, v = 0.f;
  int a = N;                                   // empty proposal: 0, arg = N (torch_scatter semantics)
  if (k != 0ull) {
    const unsigned int u = (unsigned int)(k >> 32);
    const unsigned int b = (u & 0x80000000u) ? (u & 0x7FFFFFFFu) : ~u;
    const float sc = scale[c];
    z = __uint_as_float(b);
    if (sc < 0.f) z = -z;
    a = (int)(0xFFFFFFFFu - (unsigned int)(k & 0xFFFFFFFFull));
    v = fmaxf(fmaf(z, sc, shift[c]), 0.f);
  }
  Z[p * ldz + c] = v;
  zstar[p * F + c] = z;
  arg[p * F + c] = a;
}


// partial[s] [128][128] = sum over the rows of slab s of (a_r + negmean)(a_r + negmean)^T,  A [N, 128] fp32.
// 256 threads = 2 x 2 waves, each wave a 64 x 64 quadrant (4 accumulators); per 32-row stage the centered tile goes to
// LDS once ([32][132] floats) and every wave issues 64 v_mfma_f32_32x32x2_f32 on it; next stage's loads in flight
// meanwhile.  Fixed slab partition and summation order: deterministic.
static __global__ void __launch_bounds__(256) k_gram128(const float* __restrict__ A, long lda, long N,
                                                        const float* __restrict__ negmean, int rows_per,
                                                        float* __restrict__ partial) {
  constexpr int LD = 132;
  __shared__ __attribute__((aligned(16))) float As[2][32 * LD];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, l31 = lane & 31, lhi = lane >> 5;
  const int wm = wave >> 1, wn = wave & 1;
  const long r_begin = (long)blockIdx.x * rows_per;
  long r_end = r_begin + rows_per;
  if (r_end > N) r_end = N;
  // staging role: 4 float4 per thread and stage: rows (tid >> 5) + 8 t, columns 4 (tid & 31)
  const int sc = 4 * (tid & 31), sr = tid >> 5;
  const float4 nm = *reinterpret_cast<const float4*>(negmean + sc);
  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  float4 rg[4];
  auto fetch = [&](long r0) {
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const long r = r0 + sr + 8 * t;
      rg[t] = *reinterpret_cast<const float4*>(A + (r < r_end ? r : r_end - 1) * lda + sc);
    }
  };
  auto stage = [&](long r0, int buf) {
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const bool ok = r0 + sr + 8 * t < r_end;           // rows are the reduction dimension: mask them
      const float4 v = ok ? make_float4(rg[t].x + nm.x, rg[t].y + nm.y, rg[t].z + nm.z, rg[t].w + nm.w)
                          : make_float4(0.f, 0.f, 0.f, 0.f);
      *reinterpret_cast<float4*>(&As[buf][(sr + 8 * t) * LD + sc]) = v;
    }
  };
  if (r_begin < r_end) { fetch(r_begin); stage(r_begin, 0); }
  __syncthreads();
  int buf = 0;
  for (long r0 = r_begin; r0 < r_end; r0 += 32) {
    if (r0 + 32 < r_end) fetch(r0 + 32);
    const float* t = As[buf];
#pragma unroll 4
    for (int rr = 0; rr < 32; rr += 2) {
      const float* row = t + (rr + lhi) * LD + l31;
      const float a0 = row[wm * 64], a1 = row[wm * 64 + 32], b0 = row[wn * 64], b1 = row[wn * 64 + 32];
      acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
      acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
      acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
      acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
    }
    if (r0 + 32 < r_end) stage(r0 + 32, buf ^ 1);       // the other buffer: its readers passed the last barrier
    __syncthreads();
    buf ^= 1;
  }
  float* P = partial + (long)blockIdx.x * 128 * 128;
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int n = wm * 64 + 32 * i + (r & 3) + 8 * (r >> 2) + 4 * lhi;
        P[(long)n * 128 + wn * 64 + 32 * j + l31] = acc[i][j][r];
      }
}

// ---- saved-state layout (fp32 elements): sumA[K] | ones[K] | negmean[K] | G[K*K] | T[F*K] | zstar[P*F] | arg[P*F]
struct FusSaved { float *sumA, *ones, *negmean, *G, *T, *zstar; int* arg; size_t elems; };
static FusSaved fus_saved(float* base, long K, long F, long P) {
  FusSaved s;
  size_t o = 0;
  auto take = [&](size_t n) { float* q = base ? base + o : nullptr; o += (n + 3) / 4 * 4; return q; };
  s.sumA = take(K); s.ones = take(K); s.negmean = take(K); s.G = take(K * K); s.T = take(F * K);
  s.zstar = take(P * F); s.arg = reinterpret_cast<int*>(take(P * F));
  s.elems = o;
  return s;
}

extern "C" size_t yolat_fusion_pool_train_saved_elems(int64_t K, int64_t F, int64_t P) {
  return fus_saved(nullptr, K, F, P).elems;
}

extern "C" size_t yolat_fusion_pool_train_work_elems(int64_t N, int64_t K, int64_t F, int64_t P) {
  const size_t colsum = (size_t)yl_cdiv(N, CS_ROWS) * K;
  const size_t gram = fus_gram_elems(N, K);
  const size_t keys = 2 * (size_t)P * F + 4;                                  // 64-bit keys
  const size_t wsplit = (3 * (size_t)F * K + 1) / 2 + 8;                      // bf16 split of W (bf16x6 forward GEMM)
  const size_t fwd = colsum + gram + keys + wsplit;
  // backward: GM[P*F] | column partials | sparse-dW partials | small vectors / matrices
  const size_t bwd = (size_t)P * F + 2 * (size_t)yl_cdiv(P, FB_PROWS) * F + (size_t)FB_NG * F * K + 4 * F + 2 * K * K +
                     (size_t)F * K + yolat_linear_bwd_w_work_elems(F, K, K) + 64 + (size_t)F * K + 64;
  return (fwd > bwd ? fwd : bwd) + 64;
}

extern "C" int yolat_fusion_pool_train_fwd(const float* A, int64_t lda, int64_t N, int64_t K, const float* W,
                                           const float* bias, int64_t F, const float* gamma, const float* beta,
                                           float* running_mean, float* running_var, float momentum, float eps,
                                           const int32_t* node_seg, int64_t P, float* Z, int64_t ldz, float* coef,
                                           float* saved, float* work, yolat_stream_t stream) {
  if (N <= 0 || K <= 0 || F <= 0 || P <= 0 || !A || !W || !gamma || !beta || !node_seg || !Z || !coef || !saved ||
      !work)
    return YOLAT_E_INVALID;
  if (lda < K || ldz < F || N >= (1LL << 31) || (running_mean == nullptr) != (running_var == nullptr))
    return YOLAT_E_INVALID;
  if (K % 4 != 0 || lda % 4 != 0) return YOLAT_E_UNSUPPORTED;
  hipStream_t st = (hipStream_t)stream;
  FusSaved sv = fus_saved(saved, K, F, P);
  float* colpart = work;
  float* grampart = colpart + (size_t)yl_cdiv(N, CS_ROWS) * K;
  unsigned long long* keys =
      reinterpret_cast<unsigned long long*>(grampart + (fus_gram_elems(N, K) + 1) / 2 * 2);
  if (((uintptr_t)keys & 7) != 0) keys = reinterpret_cast<unsigned long long*>((char*)keys + 4);

  // 1. column sums of A (fixed-order two-level reduction)
  const int nrb = yl_cdiv(N, CS_ROWS);
  hipLaunchKernelGGL(k_colsum_partial, dim3(yl_cdiv(K, 64), nrb), dim3(256), 0, st, A, (long)lda, (long)N, (int)K,
                     colpart);
  YL_LAUNCH_CHECK();
  hipLaunchKernelGGL(k_reduce_splits, dim3(yl_cdiv(K, 32)), dim3(256), 0, st, colpart, (long)K, nrb, sv.sumA, (long)K,
                     (int)K, 0);
  YL_LAUNCH_CHECK();
  hipLaunchKernelGGL(k_center_vecs, dim3(yl_cdiv(K, 256)), dim3(256), 0, st, sv.sumA, (int)K, 1.f / (float)N, sv.ones,
                     sv.negmean);
  YL_LAUNCH_CHECK();
  // 2. centered Gram matrix G = (A - mean)^T (A - mean)
  if (K == 128 && yl_aligned16(A)) {
    // dedicated kernel: one workgroup owns a slab of rows and the WHOLE 128 x 128 product (the tile is loaded once per
    // 32 rows and feeds 64 MFMAs per wave between barriers; the generic TN GEMM — 64 x 64 output tiles, 16 MFMAs per
    // wave and stage — was latency bound: 234 us at N = 174 k)
    int S = (int)yl_cdiv(N, 32 * 8);                    // >= 256 rows per workgroup
    if (S > FB_GRAM_S) S = FB_GRAM_S;                   // two workgroups per CU: one's loads under the other's MFMAs (66 -> 4x us)
    const int rows_per = (int)yl_cdiv(yl_cdiv(N, S), 32) * 32;
    S = (int)yl_cdiv(N, rows_per);
    if ((size_t)S * K * K > fus_gram_elems(N, K)) return YOLAT_E_INVALID;
    hipLaunchKernelGGL(k_gram128, dim3(S), dim3(256), 0, st, A, (long)lda, (long)N, sv.negmean, rows_per, grampart);
    YL_LAUNCH_CHECK();
    hipLaunchKernelGGL(k_reduce_splits, dim3(yl_cdiv(K * K, 32)), dim3(256), 0, st, grampart, (long)(K * K), S, sv.G,
                       (long)K, (int)K, 0);
    YL_LAUNCH_CHECK();
  } else {
    // the weight-gradient TN GEMM with both operands centered
    TnPlan p = yl_tn_plan(N, K, K);
    DenseProOp y = yl_dense_pro(A, lda, N, K, sv.ones, sv.negmean, 0);
    dim3 grid(yl_cdiv(K, 64), yl_cdiv(K, 64), p.S);
    hipLaunchKernelGGL((k_gemm_tn<DenseProOp, DenseProOp>), grid, dim3(256), 0, st, y, y, grampart, (float*)nullptr,
                       (int)N, (int)K, (int)K, p.rows_per_split);
    YL_LAUNCH_CHECK();
    hipLaunchKernelGGL(k_reduce_splits, dim3(yl_cdiv(K * K, 32)), dim3(256), 0, st, grampart, (long)(K * K), p.S, sv.G,
                       (long)K, (int)K, 0);
    YL_LAUNCH_CHECK();
  }
  // 3. T = W G  (G symmetric), then the BatchNorm statistics of z
  YL_TRY(yolat_linear_fwd(W, K, F, K, nullptr, nullptr, 0, sv.G, K, nullptr, K, nullptr, nullptr, 0, sv.T, K, 0, nullptr,
                          stream));
  hipLaunchKernelGGL(k_fusion_bn_stats, dim3(yl_cdiv(F * 16, 256)), dim3(256), 0, st, W, sv.T, bias, sv.sumA, (int)K, (int)F,
                     (double)N, gamma, beta, running_mean, running_var, momentum, eps, coef, coef + F, coef + 2 * F,
                     coef + 3 * F);
  YL_LAUNCH_CHECK();
  // 4. GEMM with the per-(proposal, column) extreme-of-z epilogue; nothing of shape [N, F] is written
  if (hipMemsetAsync(keys, 0, sizeof(unsigned long long) * (size_t)P * F, st) != hipSuccess) return YOLAT_E_INVALID;
  // K in {64, 128}: as an fp32 GEMM emulated with six bf16 MFMA products on the rows kernel (fusion_x6.hip; the weight
  // split lives behind the keys in `work`)
  const int use_x6 = yl_strict_fp32() ? 0 : 1;
  int x6rc = YOLAT_E_UNSUPPORTED;
  if (use_x6 && bias != nullptr) {
    uint16_t* wsplit = reinterpret_cast<uint16_t*>(((uintptr_t)(keys + (size_t)P * F) + 15) & ~(uintptr_t)15);
    x6rc = yl_fusion_rows_x6_key64(A, lda, N, K, W, bias, F, coef, node_seg, keys, wsplit, stream);
    if (x6rc != 0 && x6rc != YOLAT_E_UNSUPPORTED) return x6rc;
  }
  if (x6rc != 0) {
    DenseOp a = yl_dense(A, lda, N, K), b = yl_dense(W, K, F, K);
    Epilogue ep;
    ep.bias = bias; ep.scale = coef; ep.shift = coef + F; ep.relu = 0;
    ep.Y = nullptr; ep.ldy = 0; ep.accumulate = 0; ep.stats = nullptr;
    ep.seg = node_seg; ep.pool = nullptr; ep.ldpool = F; ep.key64 = keys;
    dim3 grid(yl_cdiv(N, 64), yl_cdiv(F, 64));
    hipLaunchKernelGGL((k_gemm_nt<64, 64, 32, DenseOp, DenseOp, false>), grid, dim3(256), 0, st, a, b, ep, (int)N,
                       (int)F, (int)K);
    YL_LAUNCH_CHECK();
  }
  // 5. decode: pooled activations, z at the arg rows, arg rows
  for (int64_t p0 = 0; p0 < P; p0 += 65535) {
    const int64_t np = (P - p0) < 65535 ? (P - p0) : 65535;
    hipLaunchKernelGGL(k_pool_finish, dim3(yl_cdiv(F, 256), (unsigned)np), dim3(256), 0, st, keys + p0 * F, (long)np,
                       (int)F, coef, coef + F, (int)N, Z + p0 * ldz, (long)ldz, sv.zstar + p0 * F, sv.arg + p0 * F);
    YL_LAUNCH_CHECK();
  }
  return 0;
}

// =================================================================================================
// backward
// =================================================================================================

// Per column: partial sums over a chunk of proposals of the masked gradient g and g*xhat; GM = scale*g.
static __global__ void __launch_bounds__(256) k_fus_cols_partial(const float* __restrict__ gZ, long ldg,
                                                          const float* __restrict__ zstar,
                                                          const int* __restrict__ arg, long P, int F, int N,
                                                          const float* coef, float* GM, float* partial) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= F) return;
  const float sc = coef[c], sh = coef[F + c], mean = coef[2 * F + c], rstd = coef[3 * F + c];
  const long p0 = (long)blockIdx.y * FB_PROWS;
  const long p1 = (p0 + FB_PROWS < P) ? p0 + FB_PROWS : P;
  float sg = 0.f, sx = 0.f;
  for (long p = p0; p < p1; p += 4) {       // 12 independent loads in flight; accumulation in proposal order
    float z[4], gz[4];
    int a[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const long pj = (p + j < p1) ? p + j : p1 - 1;
      z[j] = zstar[pj * F + c];
      gz[j] = gZ[pj * ldg + c];
      a[j] = arg[pj * F + c];
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      if (p + j < p1) {
        const float gm = (a[j] < N && fmaf(z[j], sc, sh) > 0.f) ? gz[j] : 0.f;
        sg += gm;
        sx += gm * ((z[j] - mean) * rstd);
        GM[(p + j) * F + c] = sc * gm;
      }
    }
  }
  partial[((long)blockIdx.y * 2 + 0) * F + c] = sg;
  partial[((long)blockIdx.y * 2 + 1) * F + c] = sx;
}

// 64 columns x 16 chunk lanes per workgroup: lane q sums the chunks q, q + 16, ... (ascending), the sixteen lane sums are
// added in lane order — fixed order, deterministic
static __global__ void __launch_bounds__(1024) k_fus_cols_final(const float* partial, int PB, int F, float invN,
                                                                const float* coef, float* dgamma, float* dbeta, float* dbias,
                                                                float* q1, float* nq2) {
  __shared__ float part[2][16][64];
  const int cx = threadIdx.x & 63, q = threadIdx.x >> 6, c = blockIdx.x * 64 + cx;
  float sg = 0.f, sx = 0.f;
  if (c < F)
    for (int b = q; b < PB; b += 16) {
      sg += partial[((long)b * 2 + 0) * F + c];
      sx += partial[((long)b * 2 + 1) * F + c];
    }
  part[0][q][cx] = sg;
  part[1][q][cx] = sx;
  __syncthreads();
  if (q != 0 || c >= F) return;
  sg = 0.f; sx = 0.f;
#pragma unroll
  for (int j = 0; j < 16; ++j) { sg += part[0][j][cx]; sx += part[1][j][cx]; }
  dbeta[c] = sg;
  dgamma[c] = sx;
  if (dbias) dbias[c] = 0.f;
  const float sc = coef[c], rstd = coef[3 * F + c];
  q1[c] = sc * sg * invN;
  nq2[c] = -(sc * rstd * sx * invN);
}

// Sparse weight gradient: partial[g][c][k] = sum over the row blocks of group g of GM[p,c] * A[arg[p,c], k].
// One workgroup = 128 columns x one group of 64-row blocks; the block's A rows are staged in LDS, thread
// (column, k-half) keeps 64 accumulators.  Fixed iteration order -> deterministic.
static __global__ void __launch_bounds__(256, 3) k_fus_dw_sparse(const float* __restrict__ A, long lda, int N, int K,
                                                       const int* __restrict__ node_seg,
                                                       const float* __restrict__ GM, const int* __restrict__ arg,
                                                       int F, int blocks_per_group, float* partial) {
  constexpr int LDA = 136;       // the two lanes of a column read 32 adjacent bytes; rows 8 banks apart
  __shared__ __attribute__((aligned(16))) float As[64 * LDA];
  const int tid = threadIdx.x;
  const int cl = tid >> 1, kh = tid & 1;
  const int c = blockIdx.x * 128 + cl;
  const bool c_ok = c < F;
  const int cc = c_ok ? c : F - 1;
  float acc[64];
#pragma unroll
  for (int j = 0; j < 64; ++j) acc[j] = 0.f;
  const int nb = (N + 63) / 64;
  const int b0 = blockIdx.y * blocks_per_group;
  const int b1 = (b0 + blocks_per_group < nb) ? b0 + blocks_per_group : nb;
  // software pipeline: the A rows of block b+1 and the (arg, GM) entries of its first four proposals are in flight
  // while block b is walked (a workgroup is otherwise a chain of exposed global-load latencies: 43 blocks x 3.7 us)
  float4 nxt[8];
  int pf_n = 0, pl_n = -1, a_n[4];
  float g_n[4];
#pragma unroll
  for (int t = 0; t < 8; ++t) nxt[t] = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
  for (int j = 0; j < 4; ++j) { a_n[j] = 0; g_n[j] = 0.f; }
  for (int b = b0 - 1; b < b1; ++b) {      // iteration b0 - 1 only fetches
    const int R0 = b * 64, R1 = (R0 + 64 < N) ? R0 + 64 : N;
    const bool live = b >= b0;
    const int pf = pf_n, pl = pl_n;
    int a[4];
    float g[4];
    if (live) {
      __syncthreads();
#pragma unroll
      for (int t = 0; t < 8; ++t) {
        const int i = tid + t * 256;
        *reinterpret_cast<float4*>(As + (i >> 5) * LDA + 4 * (i & 31)) = nxt[t];
      }
      __syncthreads();
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) { a[j] = a_n[j]; g[j] = g_n[j]; }
    {                                      // block b + 1 (past the group's end: a clamped, unused reload)
      const int Rn0 = yl_min(b + 1, nb - 1) * 64, Rn1 = (Rn0 + 64 < N) ? Rn0 + 64 : N;
#pragma unroll
      for (int t = 0; t < 8; ++t) {        // 64 rows x 32 float4
        const int i = tid + t * 256;
        const int r = yl_min(Rn0 + (i >> 5), N - 1);
        nxt[t] = *reinterpret_cast<const float4*>(A + (long)r * lda + 4 * (i & 31));
      }
      pf_n = node_seg[Rn0];
      pl_n = node_seg[Rn1 - 1];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int pj = yl_min(pf_n + j, pl_n);
        a_n[j] = arg[(long)pj * F + cc];
        g_n[j] = GM[(long)pj * F + cc];
      }
    }
    if (!live) continue;
    for (int p = pf; p <= pl; p += 4) {  // 4 proposals' (arg, GM) together; the first four came with the prefetch
      if (p != pf) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int pj = yl_min(p + j, pl);
          a[j] = arg[(long)pj * F + cc];
          g[j] = GM[(long)pj * F + cc];
        }
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        if (p + j <= pl && a[j] >= R0 && a[j] < R1 && g[j] != 0.f) {
          const float* row = As + (a[j] - R0) * LDA + 4 * kh;      // k = 8 q + 4 kh + (0..3)
#pragma unroll
          for (int q = 0; q < 16; ++q) {
            const float4 v = *reinterpret_cast<const float4*>(row + 8 * q);
            acc[4 * q + 0] = fmaf(g[j], v.x, acc[4 * q + 0]);
            acc[4 * q + 1] = fmaf(g[j], v.y, acc[4 * q + 1]);
            acc[4 * q + 2] = fmaf(g[j], v.z, acc[4 * q + 2]);
            acc[4 * q + 3] = fmaf(g[j], v.w, acc[4 * q + 3]);
          }
        }
      }
    }
  }
  if (c_ok) {
    float* o = partial + ((long)blockIdx.y * F + c) * K + 4 * kh;
#pragma unroll
    for (int q = 0; q < 16; ++q)
      *reinterpret_cast<float4*>(o + 8 * q) = make_float4(acc[4 * q], acc[4 * q + 1], acc[4 * q + 2], acc[4 * q + 3]);
  }
}

// dW[c,k] = sum_g partial[g][c][k]  -  q1[c]*sumA[k]  +  nq2[c]*T[c,k]
static __global__ void __launch_bounds__(256) k_fus_dw_finish(const float* __restrict__ partial, int NG, long elems, int K,
                                                       const float* q1, const float* nq2, const float* sumA,
                                                       const float* __restrict__ T, float* dW) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= elems) return;
  float s = 0.f;
  int g = 0;
  for (; g + 8 <= NG; g += 8) {
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = partial[(long)(g + j) * elems + i];
#pragma unroll
    for (int j = 0; j < 8; ++j) s += v[j];
  }
  for (; g < NG; ++g) s += partial[(long)g * elems + i];
  const int c = (int)(i / K), k = (int)(i % K);
  dW[i] = fmaf(nq2[c], T[i], fmaf(-q1[c], sumA[k], s));
}

// Sparse input gradient: dA[r, :] += sum over the columns whose arg row is r of GM[p(r), c] * W[c, :].
// One workgroup = 128 rows; thread = (row, k-half) with 64 accumulators.  W streams through LDS in 32-column
// chunks together with the arg / GM entries of the (few) proposals the 128 rows belong to.  Per chunk every
// lane builds the 32-bit match mask of ITS row and walks only its set bits (ascending column ->
// deterministic), so the number of loop trips per wave is the largest match count among its 32 rows (~5),
// not the number of columns.  Measured at N = 175k / P = 8000: v1 (test every column in every lane) 1078 us,
// v2 (wave-uniform scalar walk, one match per trip) 551 us, this version: see profiles/.
// TPR threads per row (128 / TPR accumulators each), THREADS / TPR rows per workgroup.  Every workgroup streams ALL of W
// (F x 128 floats = 512 KB) through its LDS, so rows per workgroup set the L2 -> LDS traffic: 32 rows (256 threads, TPR 8)
// = 2.8 GB at N = 175 k; 128 rows (1024 threads) = 0.7 GB.
template <int TPR, int THREADS>
static __global__ void __launch_bounds__(THREADS) k_fus_da_sparse(const float* __restrict__ W, int K, int F,
                                                              const int* __restrict__ node_seg,
                                                              const float* __restrict__ GM,
                                                              const int* __restrict__ arg, int N, float* dA,
                                                              long ldda) {
  constexpr int CH = 32, LDI = CH + 1, LDW = 132, ROWS = THREADS / TPR, NK = 128 / TPR;
  __shared__ __attribute__((aligned(16))) float Ws[CH * LDW];
  __shared__ int argP[ROWS * LDI];
  __shared__ float gmP[ROWS * LDI];
  const int tid = threadIdx.x;
  const int rl = tid / TPR, kh = tid % TPR;
  const int R0 = blockIdx.x * ROWS;
  const int row = R0 + rl;
  const int p_first = node_seg[R0];
  const int p_last = node_seg[yl_min(R0 + ROWS - 1, N - 1)];
  const int np = p_last - p_first + 1;                       // proposals touched by this row block (<= ROWS)
  const int pl = node_seg[yl_min(row, N - 1)] - p_first;     // this row's local proposal
  float acc[NK];
#pragma unroll
  for (int j = 0; j < NK; ++j) acc[j] = 0.f;
  for (int c0 = 0; c0 < F; c0 += CH) {
    __syncthreads();
#pragma unroll
    for (int t = 0; t < 1024 / THREADS; ++t) {        // W chunk: 32 columns x 32 float4
      const int i = tid + t * THREADS;
      *reinterpret_cast<float4*>(Ws + (i >> 5) * LDW + 4 * (i & 31)) =
          *reinterpret_cast<const float4*>(W + (long)(c0 + (i >> 5)) * K + 4 * (i & 31));
    }
    for (int i = tid; i < np * 8; i += THREADS) {   // arg / GM: np proposals x 8 x (4 columns)
      const int pp = i >> 3, q = i & 7;
      const long off = (long)(p_first + pp) * F + c0 + 4 * q;
      const int4 av = *reinterpret_cast<const int4*>(arg + off);
      const float4 gv = *reinterpret_cast<const float4*>(GM + off);
      int* ad = argP + pp * LDI + 4 * q;
      float* gd = gmP + pp * LDI + 4 * q;
      ad[0] = av.x; ad[1] = av.y; ad[2] = av.z; ad[3] = av.w;
      gd[0] = gv.x; gd[1] = gv.y; gd[2] = gv.z; gd[3] = gv.w;
    }
    __syncthreads();
    unsigned int m = 0u;
#pragma unroll
    for (int cc = 0; cc < CH; ++cc) m |= (argP[pl * LDI + cc] == row ? 1u : 0u) << cc;
    while (m) {
      const int cc = __builtin_ctz(m);
      m &= m - 1;
      const float g = gmP[pl * LDI + cc];
      const float* w = Ws + cc * LDW + kh * NK;
#pragma unroll
      for (int q = 0; q < NK / 4; ++q) {
        const float4 v = *reinterpret_cast<const float4*>(w + 4 * q);
        acc[4 * q + 0] = fmaf(g, v.x, acc[4 * q + 0]);
        acc[4 * q + 1] = fmaf(g, v.y, acc[4 * q + 1]);
        acc[4 * q + 2] = fmaf(g, v.z, acc[4 * q + 2]);
        acc[4 * q + 3] = fmaf(g, v.w, acc[4 * q + 3]);
      }
    }
  }
  if (row < N) {
    float* o = dA + (long)row * ldda + kh * NK;
#pragma unroll
    for (int q = 0; q < NK / 4; ++q) {
      float4 d = *reinterpret_cast<float4*>(o + 4 * q);
      d.x += acc[4 * q]; d.y += acc[4 * q + 1]; d.z += acc[4 * q + 2]; d.w += acc[4 * q + 3];
      *reinterpret_cast<float4*>(o + 4 * q) = d;
    }
  }
}

// ------------------------------------------------------------------------------------------------
// The same sparse input gradient on the bf16 matrix cores (round 3).  Why: k_fus_da_sparse streams ALL of W through the
// LDS of every 32-row workgroup (2.8 GB of L2 -> LDS traffic at N = 175 k for a 512 KB weight) and walks a match mask
// per row and 32-column chunk: 358 us at cfg 3, the largest kernel of the training step.
//     dA[rows, 0:128] += M . W,   M[r, c] = GM[p(r), c] if arg[p(r), c] == r else 0      ([rows, F], one entry per column
// and proposal) is a GEMM whose A operand is BUILT in registers from arg / GM (8 consecutive columns per lane: one
// compare + select per element) and split into two bf16 terms, against W^T pre-split once per step into two bf16 terms
// (k_wt_split2): three products a_h b_h + a_h b_m + a_m b_h per 16 columns reproduce the fp32 product to 2^-16 — the
// gradient tolerance is 2e-4 — with fp32 accumulation in a fixed order (deterministic).  A 512-thread workgroup owns 256
// rows; W^T's two planes and the arg / GM entries of the workgroup's (~10) proposals stream through double-buffered LDS
// tiles of CH columns (W once per 256 rows: 0.35 GB).  Measured at cfg 3 (N = 175 k, P = 8000): 358 -> 174 us (CH = 32,
// two workgroups per CU); the matrix-core time of the 4.2 M MFMAs is 62 us.
// ------------------------------------------------------------------------------------------------
typedef __bf16 ft_bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned ft_u32x4 __attribute__((ext_vector_type(4)));

// WT_h / WT_m [K][F] bf16:  W[c][k] = h + m + O(2^-16)   (h = top 8 significand bits, m = the next 8; truncation)
static __global__ void k_wt_split2(const float* __restrict__ W, int F, int K, unsigned short* __restrict__ WTh,
                                   unsigned short* __restrict__ WTm) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;          // (k, c), c fastest
  if (i >= F * K) return;
  const int k = i / F, c = i - k * F;
  const float x = W[(long)c * K + k];
  const unsigned hb = __float_as_uint(x) & 0xFFFF0000u;
  const float r = x - __uint_as_float(hb);
  WTh[i] = (unsigned short)(hb >> 16);
  WTm[i] = (unsigned short)(__float_as_uint(r) >> 16);
}

template <int CH>
static __global__ void __launch_bounds__(512, (CH == 32 ? 4 : 2)) k_fus_da_mfma(const unsigned short* __restrict__ WTh,
                                                          const unsigned short* __restrict__ WTm, int F,
                                                          const int* __restrict__ node_seg, int P,
                                                          const float* __restrict__ GM, const int* __restrict__ arg,
                                                          int N, float* dA, long ldda) {
  constexpr int RS = CH + 8;                              // CH columns per LDS chunk, row stride (bf16)
  constexpr int NPL = 32, PS = CH + 4;                    // proposals whose arg / GM chunk is staged in LDS, row stride
  constexpr int WPT = 2 * 128 * (CH / 8) / 512;           // 16-byte W^T pieces per thread (2 planes x 128 k rows)
  constexpr int SQN = CH / 4;                             // arg / GM staging: column quads per proposal
  __shared__ __attribute__((aligned(16))) unsigned short Ws[2][2][128 * RS];     // [buffer][plane][k row][c]
  __shared__ __attribute__((aligned(16))) int argS[2][NPL * PS];
  __shared__ __attribute__((aligned(16))) float gmS[2][NPL * PS];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, l31 = lane & 31, lhi = lane >> 5;
  const int R0 = blockIdx.x * 256;
  const int row = R0 + wave * 32 + l31;
  const int rowc = yl_min(row, N - 1);
  const int p_first = node_seg[R0];
  const int pl = node_seg[rowc] - p_first;                // this row's proposal, local to the workgroup
  // (256 rows hold ~10 proposals; a WAVE with a row beyond the 32 staged proposals reads arg / GM straight from L2 —
  // wave-uniform, so that the common path carries no masked-off global loads: they were 256 of a wave's 449 vector loads)
  const bool in_lds = __builtin_amdgcn_ballot_w64(pl >= NPL) == 0ull;
  const long pbase = (long)(p_first + pl) * F + 8 * lhi;
  const int myrow = row < N ? row : -1;
  // staging roles: W^T pieces (2 planes x 128 k rows x CH/8 pieces); arg / GM: thread = (proposal, 4 columns)
  const int sp = tid / SQN, sq = tid % SQN;
  const bool stager = sp < NPL;
  const long sbase = (long)yl_min(p_first + sp, P - 1) * F + 4 * sq;
  auto load_w = [&](int c0, ft_u32x4* rw, int4& ra, float4& rg) {
#pragma unroll
    for (int t = 0; t < WPT; ++t) {
      const int i = tid + 512 * t, plane = i / (128 * (CH / 8)), r = (i / (CH / 8)) & 127, pc = i % (CH / 8);
      rw[t] = *reinterpret_cast<const ft_u32x4*>((plane ? WTm : WTh) + (long)r * F + c0 + 8 * pc);
    }
    ra = *reinterpret_cast<const int4*>(arg + sbase + c0);
    rg = *reinterpret_cast<const float4*>(GM + sbase + c0);
  };
  auto store_w = [&](int buf, const ft_u32x4* rw, const int4& ra, const float4& rg) {
#pragma unroll
    for (int t = 0; t < WPT; ++t) {
      const int i = tid + 512 * t, plane = i / (128 * (CH / 8)), r = (i / (CH / 8)) & 127, pc = i % (CH / 8);
      *reinterpret_cast<ft_u32x4*>(&Ws[buf][plane][r * RS + 8 * pc]) = rw[t];
    }
    if (stager) {
      *reinterpret_cast<int4*>(&argS[buf][sp * PS + 4 * sq]) = ra;
      *reinterpret_cast<float4*>(&gmS[buf][sp * PS + 4 * sq]) = rg;
    }
  };
  f32x16 acc[4];
#pragma unroll
  for (int j = 0; j < 4; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
  ft_u32x4 rw[WPT];
  int4 ra;
  float4 rg;
  load_w(0, rw, ra, rg);
  store_w(0, rw, ra, rg);
  __syncthreads();
  const int nch = F / CH;
  for (int ch = 0; ch < nch; ++ch) {
    const int buf = ch & 1;
    if (ch + 1 < nch) load_w((ch + 1) * CH, rw, ra, rg);
#pragma unroll
    for (int ks = 0; ks < CH / 16; ++ks) {
      // ---- A fragments of this k step: M[row, c] = GM[p, c] where arg[p, c] == row, in two bf16 terms
      int4 a0, a1;
      float4 g0, g1;
      if (in_lds) {
        // (explicit LDS address space: left to itself the compiler selects between the LDS and the global POINTER and issues
        // flat loads — they count on vmcnt AND lgkmcnt, and every wait for them drained the weight prefetches in flight)
        typedef int ft_i32x4 __attribute__((ext_vector_type(4)));
        typedef float ft_f32x4 __attribute__((ext_vector_type(4)));
        typedef const __attribute__((address_space(3))) ft_i32x4* lds_i4;
        typedef const __attribute__((address_space(3))) ft_f32x4* lds_f4;
        const int* ap = &argS[buf][pl * PS + 16 * ks + 8 * lhi];
        const float* gp = &gmS[buf][pl * PS + 16 * ks + 8 * lhi];
        const ft_i32x4 x0 = *(lds_i4)ap, x1 = *(lds_i4)(ap + 4);
        const ft_f32x4 y0 = *(lds_f4)gp, y1 = *(lds_f4)(gp + 4);
        a0 = make_int4(x0.x, x0.y, x0.z, x0.w); a1 = make_int4(x1.x, x1.y, x1.z, x1.w);
        g0 = make_float4(y0.x, y0.y, y0.z, y0.w); g1 = make_float4(y1.x, y1.y, y1.z, y1.w);
      } else {
        const long o = pbase + ch * CH + 16 * ks;
        a0 = *reinterpret_cast<const int4*>(arg + o); a1 = *reinterpret_cast<const int4*>(arg + o + 4);
        g0 = *reinterpret_cast<const float4*>(GM + o); g1 = *reinterpret_cast<const float4*>(GM + o + 4);
      }
      const int av[8] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w};
      const float gv[8] = {g0.x, g0.y, g0.z, g0.w, g1.x, g1.y, g1.z, g1.w};
      unsigned ph[4], pm[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const float x0 = av[2 * i] == myrow ? gv[2 * i] : 0.f, x1 = av[2 * i + 1] == myrow ? gv[2 * i + 1] : 0.f;
        const unsigned h0 = __float_as_uint(x0) & 0xFFFF0000u, h1 = __float_as_uint(x1) & 0xFFFF0000u;
        const float m0 = x0 - __uint_as_float(h0), m1 = x1 - __uint_as_float(h1);
        ph[i] = __builtin_amdgcn_perm(h1, h0, 0x07060302u);
        pm[i] = __builtin_amdgcn_perm(__float_as_uint(m1), __float_as_uint(m0), 0x07060302u);
      }
      const ft_u32x4 qh = {ph[0], ph[1], ph[2], ph[3]}, qm = {pm[0], pm[1], pm[2], pm[3]};
      const ft_bf16x8 Ah = __builtin_bit_cast(ft_bf16x8, qh), Am = __builtin_bit_cast(ft_bf16x8, qm);
      const unsigned short* bh = &Ws[buf][0][l31 * RS + 16 * ks + 8 * lhi];
      const unsigned short* bm = &Ws[buf][1][l31 * RS + 16 * ks + 8 * lhi];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const ft_bf16x8 Bh = *reinterpret_cast<const ft_bf16x8*>(bh + 32 * j * RS);
        const ft_bf16x8 Bm = *reinterpret_cast<const ft_bf16x8*>(bm + 32 * j * RS);
        acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Am, Bh, acc[j], 0, 0, 0);
        acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Ah, Bm, acc[j], 0, 0, 0);
        acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Ah, Bh, acc[j], 0, 0, 0);
      }
    }
    if (ch + 1 < nch) store_w(buf ^ 1, rw, ra, rg);
    __syncthreads();
  }
  // dA += acc   (C layout: lane = column, register = row)
  const int rb = R0 + wave * 32 + 4 * lhi;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int rr = rb + (r & 3) + 8 * (r >> 2);
    if (rr < N) {
      float* o = dA + (long)rr * ldda + l31;
#pragma unroll
      for (int j = 0; j < 4; ++j) o[32 * j] += acc[j][r];
    }
  }
}

static __global__ void k_row_scale(const float* __restrict__ W, const float* s, long elems, int K, float* out) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < elems) out[i] = s[i / K] * W[i];
}

// nu[k] = - sum_c q1[c] * W[c,k];  K <= 128: 8 column groups x 128 k's in one 1024-thread workgroup,
// group partial sums combined in a fixed order
static __global__ void __launch_bounds__(1024) k_wt_vec(const float* __restrict__ W, const float* q1, int F, int K,
                                                        float* nu) {
  __shared__ float red[8][128];
  const int k = threadIdx.x & 127, grp = threadIdx.x >> 7;
  const int per = (F + 7) / 8;
  const int c0 = grp * per, c1 = (c0 + per < F) ? c0 + per : F;
  float s = 0.f;
  if (k < K) {
    int c = c0;
    for (; c + 8 <= c1; c += 8) {
      float w[8], q[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) { w[j] = W[(long)(c + j) * K + k]; q[j] = q1[c + j]; }
#pragma unroll
      for (int j = 0; j < 8; ++j) s = fmaf(q[j], w[j], s);
    }
    for (; c < c1; ++c) s = fmaf(q1[c], W[(long)c * K + k], s);
  }
  red[grp][k] = s;
  __syncthreads();
  if (grp == 0 && k < K) {
    float t = red[0][k];
#pragma unroll
    for (int g = 1; g < 8; ++g) t += red[g][k];
    nu[k] = -t;
  }
}

extern "C" int yolat_fusion_pool_train_bwd(const float* A, int64_t lda, int64_t N, int64_t K, const float* W,
                                           const float* gamma, int64_t F, const float* coef, const float* saved,
                                           const int32_t* node_seg, const int32_t* seg_ptr, int64_t P,
                                           const float* gZ, int64_t ldg, float* dW, float* dbias, float* dgamma,
                                           float* dbeta, float* dA, int64_t ldda, float* work,
                                           yolat_stream_t stream) {
  return yolat_fusion_pool_train_bwd_parts(A, lda, N, K, W, gamma, F, coef, saved, node_seg, seg_ptr, P, gZ, ldg, dW, dbias,
                                           dgamma, dbeta, dA, ldda, work, YOLAT_FUS_BWD_ALL, stream);
}

// The same backward in parts (bit mask), for a caller that runs the weight gradient beside the input gradient on a second
// stream: YOLAT_FUS_BWD_COLS (the per-column reductions: dgamma, dbeta and the coefficient vectors both other parts read)
// must be complete — stream order or an event — before _DW and _DA, which write disjoint regions of `work`.
extern "C" int yolat_fusion_pool_train_bwd_parts(const float* A, int64_t lda, int64_t N, int64_t K, const float* W,
                                                 const float* gamma, int64_t F, const float* coef, const float* saved,
                                                 const int32_t* node_seg, const int32_t* seg_ptr, int64_t P,
                                                 const float* gZ, int64_t ldg, float* dW, float* dbias, float* dgamma,
                                                 float* dbeta, float* dA, int64_t ldda, float* work, int parts,
                                                 yolat_stream_t stream) {
  (void)gamma; (void)seg_ptr;
  if ((parts & ~YOLAT_FUS_BWD_ALL) != 0 || parts == 0) return YOLAT_E_INVALID;
  if (N <= 0 || K <= 0 || F <= 0 || P <= 0 || !A || !W || !coef || !saved || !node_seg || !gZ || !dW || !dgamma ||
      !dbeta || !dA || !work)
    return YOLAT_E_INVALID;
  if (lda < K || ldg < F || ldda < K || N >= (1LL << 31)) return YOLAT_E_INVALID;
  if (K != 128 || F % 32 != 0 || lda % 4 != 0 || ldda % 4 != 0 || ((uintptr_t)dA & 15) != 0)
    return YOLAT_E_UNSUPPORTED;
  hipStream_t st = (hipStream_t)stream;
  FusSaved sv = fus_saved(const_cast<float*>(saved), K, F, P);
  const int PB = yl_cdiv(P, FB_PROWS);
  size_t o = 0;
  auto take = [&](size_t n) { float* q = work + o; o += (n + 3) / 4 * 4; return q; };
  float* GM = take((size_t)P * F);
  float* colpart = take((size_t)2 * PB * F);
  float* dwpart = take((size_t)FB_NG * F * K);
  float* q1 = take(F);
  float* nq2 = take(F);
  float* nu = take(K);
  float* Wq = take((size_t)F * K);
  float* nQ = take((size_t)K * K);
  float* tnpart = take(yolat_linear_bwd_w_work_elems(F, K, K));

  // 1. per-column reductions over the P*F sparse entries: dbeta, dgamma and the coefficient vectors
  if (parts & YOLAT_FUS_BWD_COLS) {
  hipLaunchKernelGGL(k_fus_cols_partial, dim3(yl_cdiv(F, 256), PB), dim3(256), 0, st, gZ, (long)ldg, sv.zstar, sv.arg,
                     (long)P, (int)F, (int)N, coef, GM, colpart);
  YL_LAUNCH_CHECK();
  hipLaunchKernelGGL(k_fus_cols_final, dim3(yl_cdiv(F, 64)), dim3(1024), 0, st, colpart, PB, (int)F, 1.f / (float)N, coef,
                     dgamma, dbeta, dbias, q1, nq2);
  YL_LAUNCH_CHECK();
  }
  // 2. weight gradient: sparse gather term + the two rank-structured dense terms
  if (parts & YOLAT_FUS_BWD_DW) {
  const int nb = yl_cdiv(N, 64);
  const int dw_ng = FB_NG;
  const int bpg = yl_cdiv(nb, dw_ng);
  const int ng = yl_cdiv(nb, bpg);
  hipLaunchKernelGGL(k_fus_dw_sparse, dim3(yl_cdiv(F, 128), ng), dim3(256), 0, st, A, (long)lda, (int)N, (int)K,
                     node_seg, GM, sv.arg, (int)F, bpg, dwpart);
  YL_LAUNCH_CHECK();
  hipLaunchKernelGGL(k_fus_dw_finish, dim3(yl_cdiv(F * K, 256)), dim3(256), 0, st, dwpart, ng, (long)(F * K), (int)K, q1,
                     nq2, sv.sumA, sv.T, dW);
  YL_LAUNCH_CHECK();
  }
  if (!(parts & YOLAT_FUS_BWD_DA)) return 0;
  // 3. input gradient: sparse scatter term, then  dA += (A - mean_A) . (-Q) - u   with Q = W^T diag(q2) W
  const int da_threads = 256;   // measured at N = 175 k: 256 -> 3.78, 512 -> 3.75, 1024 -> 3.86 ms per cfg-3 step (not the W re-staging: the walk)
  // the matrix-core form splits the operands into TWO bf16 terms (three products, 2^-16 per product): not under
  // YOLAT_STRICT_FP32, which promises fp32 arithmetic / IEEE propagation for every GEMM of the fp32 mode
  const int da_mfma = yl_strict_fp32() ? 0 : 1, da_ch = 32;
  if (da_mfma && F % 64 == 0 && F >= 64) {
    unsigned short* WTh = reinterpret_cast<unsigned short*>(take((size_t)F * K / 2 + 8));
    unsigned short* WTm = reinterpret_cast<unsigned short*>(take((size_t)F * K / 2 + 8));
    hipLaunchKernelGGL(k_wt_split2, dim3(yl_cdiv(F * K, 256)), dim3(256), 0, st, W, (int)F, (int)K, WTh, WTm);
    YL_LAUNCH_CHECK();
    if (da_ch == 64)
      hipLaunchKernelGGL(k_fus_da_mfma<64>, dim3(yl_cdiv(N, 256)), dim3(512), 0, st, WTh, WTm, (int)F, node_seg, (int)P, GM,
                         sv.arg, (int)N, dA, (long)ldda);
    else
      hipLaunchKernelGGL(k_fus_da_mfma<32>, dim3(yl_cdiv(N, 256)), dim3(512), 0, st, WTh, WTm, (int)F, node_seg, (int)P, GM,
                         sv.arg, (int)N, dA, (long)ldda);
  } else if (da_threads == 1024)
    hipLaunchKernelGGL((k_fus_da_sparse<8, 1024>), dim3(yl_cdiv(N, 128)), dim3(1024), 0, st, W, (int)K, (int)F, node_seg, GM,
                       sv.arg, (int)N, dA, (long)ldda);
  else if (da_threads == 512)
    hipLaunchKernelGGL((k_fus_da_sparse<8, 512>), dim3(yl_cdiv(N, 64)), dim3(512), 0, st, W, (int)K, (int)F, node_seg, GM,
                       sv.arg, (int)N, dA, (long)ldda);
  else
  hipLaunchKernelGGL((k_fus_da_sparse<8, 256>), dim3(yl_cdiv(N, 32)), dim3(256), 0, st, W, (int)K, (int)F, node_seg, GM,
                     sv.arg, (int)N, dA, (long)ldda);
  YL_LAUNCH_CHECK();
  hipLaunchKernelGGL(k_row_scale, dim3(yl_cdiv(F * K, 256)), dim3(256), 0, st, W, nq2, (long)(F * K), (int)K, Wq);
  YL_LAUNCH_CHECK();
  YL_TRY(yolat_linear_bwd_w(Wq, K, F, K, W, K, K, nullptr, nullptr, 0, nQ, K, nullptr, 0, tnpart, stream));
  hipLaunchKernelGGL(k_wt_vec, dim3(1), dim3(1024), 0, st, W, q1, (int)F, (int)K, nu);
  YL_LAUNCH_CHECK();
  YL_TRY(yolat_linear_fwd(A, lda, N, K, sv.ones, sv.negmean, 0, nQ, K, nu, K, nullptr, nullptr, 0, dA, ldda, 1, nullptr,
                          stream));
  return 0;
}
