// loss_optim.hip — nn.CrossEntropyLoss of DetectionLoss (architecture3cc_rpn_gp_iter2.py:363,376)
// and the torch.optim.Adam step of cad_recognition/train.py:212,284 over one flat buffer.
#include "common.hpp"

// One 1024-thread workgroup; thread t owns rows t, t+1024, ...; fixed-order tree reduction.
__global__ void __launch_bounds__(1024) k_softmax_ce(const float* logits, long ld,
                                                     const int64_t* labels, int P, int K,
                                                     float* loss, float* dl, long lddl) {
  __shared__ float red[1024];
  const int tid = threadIdx.x;
  const float invP = 1.f / (float)P;
  float acc = 0.f;
  for (int p = tid; p < P; p += 1024) {
    const float* z = logits + (long)p * ld;
    float m = z[0];
    for (int k = 1; k < K; ++k) m = fmaxf(m, z[k]);
    float s = 0.f;
    for (int k = 0; k < K; ++k) s += expf(z[k] - m);
    const float lse = m + logf(s);
    const int64_t yl = labels[p];
    const bool bad = yl < 0 || yl >= K;          // no out-of-bounds read; the loss is poisoned (NaN) instead
    const int y = bad ? 0 : (int)yl;
    acc += bad ? __builtin_nanf("") : lse - z[y];
    if (dl != nullptr) {
      float* d = dl + (long)p * lddl;
      const float invs = 1.f / s;
      for (int k = 0; k < K; ++k) {
        float g = expf(z[k] - m) * invs;
        if (k == y) g -= 1.f;
        d[k] = g * invP;
      }
    }
  }
  red[tid] = acc;
  __syncthreads();
  for (int s = 512; s > 0; s >>= 1) {
    if (tid < s) red[tid] += red[tid + s];
    __syncthreads();
  }
  if (tid == 0) loss[0] = red[0] * invP;
}

// Multi-workgroup variant: one row per thread, 256 rows per workgroup; the row is held in registers
// (K <= 32: all loads independent, one expf per element), the workgroup's loss sum goes to work[wg] via a
// fixed-order tree, and k_ce_final adds the per-workgroup sums in a fixed order -> run-to-run deterministic.
template <int KMAX>
__global__ void __launch_bounds__(256) k_softmax_ce_rows(const float* logits, long ld, const int64_t* labels,
                                                         int P, int K, float* work, float* dl, long lddl) {
  __shared__ float red[256];
  const int tid = threadIdx.x;
  const int p = blockIdx.x * 256 + tid;
  float row_loss = 0.f;
  if (p < P) {
    const float* z = logits + (long)p * ld;
    float v[KMAX];
#pragma unroll
    for (int k = 0; k < KMAX; ++k) v[k] = z[k < K ? k : K - 1];
    float m = v[0];
#pragma unroll
    for (int k = 1; k < KMAX; ++k) m = fmaxf(m, v[k]);
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < KMAX; ++k) {
      v[k] = expf(v[k] - m);
      s += (k < K) ? v[k] : 0.f;
    }
    // a label outside [0, K) (torch raises "Target out of bounds"; ignore_index is not used by the reference,
    // architecture3cc_rpn_gp_iter2.py:363) must not become an out-of-bounds read: the loss is poisoned with NaN
    const int64_t yl = labels[p];
    const bool bad = yl < 0 || yl >= K;
    const int y = bad ? 0 : (int)yl;
    row_loss = bad ? __builtin_nanf("") : m + logf(s) - z[y];
    if (dl != nullptr) {
      const float invs = 1.f / s, invP = 1.f / (float)P;
      float* d = dl + (long)p * lddl;
#pragma unroll
      for (int k = 0; k < KMAX; ++k)
        if (k < K) d[k] = (v[k] * invs - (k == y ? 1.f : 0.f)) * invP;
    }
  }
  red[tid] = row_loss;
  __syncthreads();
  for (int st = 128; st > 0; st >>= 1) {
    if (tid < st) red[tid] += red[tid + st];
    __syncthreads();
  }
  if (tid == 0) work[blockIdx.x] = red[0];
}

__global__ void __launch_bounds__(1024) k_ce_final(const float* work, int n, int P, float* loss) {
  __shared__ float red[1024];
  const int tid = threadIdx.x;
  float a = 0.f;
  for (int i = tid; i < n; i += 1024) a += work[i];
  red[tid] = a;
  __syncthreads();
  for (int st = 512; st > 0; st >>= 1) {
    if (tid < st) red[tid] += red[tid + st];
    __syncthreads();
  }
  if (tid == 0) loss[0] = red[0] / (float)P;
}

extern "C" size_t yolat_softmax_ce_work_elems(int64_t P) { return (size_t)((P + 255) / 256 + 1); }

extern "C" int yolat_softmax_ce(const float* logits, int64_t ld, const int64_t* labels, int64_t P,
                                int64_t K, float* loss, float* dlogits, int64_t lddl, float* work,
                                yolat_stream_t stream) {
  if (P <= 0 || K <= 0 || !logits || !labels || !loss || ld < K || P >= (1LL << 31))
    return YOLAT_E_INVALID;
  if (dlogits && lddl < K) return YOLAT_E_INVALID;
  if (work != nullptr && K <= 32) {
    const int nwg = yl_cdiv(P, 256);
    hipLaunchKernelGGL(k_softmax_ce_rows<32>, dim3(nwg), dim3(256), 0, (hipStream_t)stream, logits, (long)ld,
                       labels, (int)P, (int)K, work, dlogits, (long)lddl);
    YL_LAUNCH_CHECK();
    hipLaunchKernelGGL(k_ce_final, dim3(1), dim3(1024), 0, (hipStream_t)stream, work, nwg, (int)P, loss);
    YL_LAUNCH_CHECK();
    return 0;
  }
  hipLaunchKernelGGL(k_softmax_ce, dim3(1), dim3(1024), 0, (hipStream_t)stream, logits, (long)ld,
                     labels, (int)P, (int)K, loss, dlogits, (long)lddl);
  YL_LAUNCH_CHECK();
  return 0;
}

// torch.optim.Adam (single-tensor formulation): g += wd*p; m.lerp_(g, 1-b1);
// v = b2*v + (1-b2)*g*g; p -= step_size * m / (sqrt(v)/sqrt(bc2) + eps)
__global__ void k_adam(float* p, const float* g, float* m, float* v, long n, float step_size,
                       float beta1, float beta2, float inv_bc2_sqrt, float eps, float wd,
                       float gscale) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += (long)gridDim.x * blockDim.x) {
    const float pi = p[i];
    float gi = g[i] * gscale;
    if (wd != 0.f) gi = fmaf(wd, pi, gi);
    float mi = m[i];
    mi = mi + (gi - mi) * (1.f - beta1);
    const float vi = v[i] * beta2 + (1.f - beta2) * gi * gi;
    const float denom = sqrtf(vi) * inv_bc2_sqrt + eps;
    m[i] = mi;
    v[i] = vi;
    p[i] = pi - step_size * (mi / denom);
  }
}

extern "C" int yolat_adam_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq,
                               int64_t n, float lr, float beta1, float beta2, float eps,
                               float weight_decay, int64_t step, float grad_scale,
                               yolat_stream_t stream) {
  if (n <= 0 || step <= 0 || !param || !grad || !exp_avg || !exp_avg_sq) return YOLAT_E_INVALID;
  const double bc1 = 1.0 - pow((double)beta1, (double)step);
  const double bc2 = 1.0 - pow((double)beta2, (double)step);
  const float step_size = (float)((double)lr / bc1);
  const float inv_bc2_sqrt = (float)(1.0 / sqrt(bc2));
  int blocks = yl_cdiv(n, 256);
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(k_adam, dim3(blocks), dim3(256), 0, (hipStream_t)stream, param, grad, exp_avg,
                     exp_avg_sq, (long)n, step_size, beta1, beta2, inv_bc2_sqrt, eps, weight_decay,
                     grad_scale);
  YL_LAUNCH_CHECK();
  return 0;
}

// ------------------------------------------------------------------------------------------------
// nn.Dropout2d(p) of MLP (gcn_lib/sparse/torch_nn.py:67-68), training mode, on a [M,C] activation.  With the
// reference's torch 1.7.1 a 2-D input gets ELEMENT-wise Bernoulli(1-p) noise scaled by 1/(1-p) (feature_dropout's
// noise has the shape of a 2-D input); that is what this does.  The producer's lazy BatchNorm + ReLU is applied on
// the way: Z = relu?(Y*scale + shift) * keep / (1-p).  keep[r,c] = 1 iff hash(seed, r*C + c) >= p * 2^32 — a
// counter-based generator (splitmix64 finaliser), so the mask is a pure function of (seed, position): deterministic,
// independent of the launch geometry; the mask is stored (uint8) for the backward.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ unsigned yl_hash32(unsigned long long seed, unsigned long long idx) {
  unsigned long long z = seed + 0x9E3779B97F4A7C15ull * (idx + 1ull);
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  z = z ^ (z >> 31);
  return (unsigned)(z >> 32);
}

__global__ void k_dropout_fwd(const float* __restrict__ Y, long ldy, long M, int C, const float* __restrict__ scale,
                              const float* __restrict__ shift, int relu, unsigned thresh, float inv_keep,
                              unsigned long long seed, unsigned char* __restrict__ mask, float* __restrict__ Z, long ldz) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= M * C) return;
  const long r = i / C;
  const int c = (int)(i % C);
  float v = Y[r * ldy + c];
  if (scale != nullptr) v = fmaf(v, scale[c], shift[c]);
  if (relu) v = fmaxf(v, 0.f);
  const unsigned char keep = yl_hash32(seed, (unsigned long long)i) >= thresh ? 1 : 0;
  mask[i] = keep;
  Z[r * ldz + c] = keep ? v * inv_keep : 0.f;
}

__global__ void k_dropout_bwd(const float* __restrict__ dZ, long lddz, long M, int C,
                              const unsigned char* __restrict__ mask, float inv_keep, float* __restrict__ dX, long lddx) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= M * C) return;
  const long r = i / C;
  const int c = (int)(i % C);
  dX[r * lddx + c] = mask[i] ? dZ[r * lddz + c] * inv_keep : 0.f;
}

extern "C" int yolat_dropout_fwd(const float* Y, int64_t ldy, int64_t M, int64_t C, const float* scale,
                                 const float* shift, int relu, float p, uint64_t seed, uint8_t* mask, float* Z,
                                 int64_t ldz, yolat_stream_t stream) {
  if (M < 0 || C <= 0 || !(p >= 0.f && p < 1.f) || (scale == nullptr) != (shift == nullptr)) return YOLAT_E_INVALID;
  if (M == 0) return 0;
  if (!Y || !mask || !Z || ldy < C || ldz < C) return YOLAT_E_INVALID;
  const double t = (double)p * 4294967296.0;
  const unsigned thresh = t >= 4294967295.0 ? 0xFFFFFFFFu : (unsigned)t;
  hipLaunchKernelGGL(k_dropout_fwd, dim3(yl_cdiv(M * C, 256)), dim3(256), 0, (hipStream_t)stream, Y, (long)ldy, (long)M,
                     (int)C, scale, shift, relu, thresh, 1.f / (1.f - p), (unsigned long long)seed, mask, Z, (long)ldz);
  YL_LAUNCH_CHECK();
  return 0;
}

extern "C" int yolat_dropout_bwd(const float* dZ, int64_t lddz, int64_t M, int64_t C, const uint8_t* mask, float p,
                                 float* dX, int64_t lddx, yolat_stream_t stream) {
  if (M < 0 || C <= 0 || !(p >= 0.f && p < 1.f)) return YOLAT_E_INVALID;
  if (M == 0) return 0;
  if (!dZ || !mask || !dX || lddz < C || lddx < C) return YOLAT_E_INVALID;
  hipLaunchKernelGGL(k_dropout_bwd, dim3(yl_cdiv(M * C, 256)), dim3(256), 0, (hipStream_t)stream, dZ, (long)lddz, (long)M,
                     (int)C, mask, 1.f / (1.f - p), dX, (long)lddx);
  YL_LAUNCH_CHECK();
  return 0;
}

extern "C" int yolat_abi_version(void) { return 6; }

extern "C" const char* yolat_strerror(int code) {
  if (code == 0) return "ok";
  if (code == YOLAT_E_INVALID) return "yolat: invalid argument (size, NULL pointer or leading dimension)";
  if (code == YOLAT_E_UNSUPPORTED) return "yolat: shape not supported by the gfx950 kernels";
  if (code > 0) return hipGetErrorString((hipError_t)code);
  return "yolat: unknown error";
}
