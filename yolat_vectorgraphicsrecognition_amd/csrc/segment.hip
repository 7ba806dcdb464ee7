// segment.hip — per-proposal pooling = torch_scatter.scatter(src, bbox_idx, dim=0,
// reduce='mean'|'max') of cad_recognition/architecture3cc_rpn_gp_iter2.py:67,122.
// bbox_idx is non-decreasing (Datasets/graph_dict3.py:732) so a proposal is a contiguous row range
// [seg_ptr[p], seg_ptr[p+1]).  One thread per (proposal, column): threads of a workgroup cover 256
// consecutive columns -> every row read is a coalesced 1 KiB burst; no atomics, fixed row order.
#include "common.hpp"

__global__ void __launch_bounds__(256) k_segment_mean_fwd(const float* X, long ldx, int D,
                                                          const float* xs, const float* xb,
                                                          int relu, const int* seg_ptr, float* Y,
                                                          long ldy) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  const int p = blockIdx.y;
  if (c >= D) return;
  const int r0 = seg_ptr[p], r1 = seg_ptr[p + 1];
  const float sc = xs ? xs[c] : 1.f, sh = xs ? xb[c] : 0.f;
  float s = 0.f;
  // rows are read 8 at a time (independent loads in flight); the sum itself still runs in row order
  int r = r0;
  for (; r + 8 <= r1; r += 8) {
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = X[(long)(r + j) * ldx + c];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      float t = v[j];
      if (xs) t = fmaf(t, sc, sh);
      if (relu) t = fmaxf(t, 0.f);
      s += t;
    }
  }
  for (; r < r1; ++r) {
    float v = X[(long)r * ldx + c];
    if (xs) v = fmaf(v, sc, sh);
    if (relu) v = fmaxf(v, 0.f);
    s += v;
  }
  const int cnt = r1 - r0;
  Y[(long)p * ldy + c] = s / (float)(cnt > 1 ? cnt : 1);
}

__global__ void __launch_bounds__(256) k_segment_max_fwd(const float* X, long ldx, int D,
                                                         const float* xs, const float* xb,
                                                         int relu, const int* seg_ptr, int N,
                                                         float* Y, long ldy, int* arg) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  const int p = blockIdx.y;
  if (c >= D) return;
  const int r0 = seg_ptr[p], r1 = seg_ptr[p + 1];
  const float sc = xs ? xs[c] : 1.f, sh = xs ? xb[c] : 0.f;
  float best = 0.f;
  int a = N;  // empty segment -> 0, arg = N   (torch_scatter semantics)
  int r = r0;
  for (; r + 8 <= r1; r += 8) {                  // 8 independent loads in flight, compared in row order
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = X[(long)(r + j) * ldx + c];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      float t = v[j];
      if (xs) t = fmaf(t, sc, sh);
      if (relu) t = fmaxf(t, 0.f);
      if (a == N || t > best) { best = t; a = r + j; }
    }
  }
  for (; r < r1; ++r) {
    float v = X[(long)r * ldx + c];
    if (xs) v = fmaf(v, sc, sh);
    if (relu) v = fmaxf(v, 0.f);
    if (a == N || v > best) { best = v; a = r; }  // strict '>' : first (lowest) row wins ties
  }
  Y[(long)p * ldy + c] = best;
  if (arg) arg[(long)p * D + c] = a;
}

// Pooling prologue of the eval plan, one launch: for proposal p
//   Z[p, 0:F]              = 0                       (target of the fused GEMM+max kernel)
//   Z[p, F:F+D]            = max over rows of feats  (pass-through half of out_feat, arch:63,122)
//   Z[p, 2F+D:2F+2D]       = mean over rows of fsup  (arch:67)
__global__ void __launch_bounds__(256) k_pool_prepare(const float* feats, const float* fsup, long ld, int D,
                                                      int F, const int* seg_ptr, float* Z, long ldz, int parts) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  const int p = blockIdx.y;
  float* z = Z + (long)p * ldz;
  if (c < F) { if (parts & YL_POOL_ZERO) z[c] = 0.f; return; }
  if (c >= F + 2 * D || (fsup == nullptr && c >= F + D)) return;
  const int r0 = seg_ptr[p], r1 = seg_ptr[p + 1];
  const bool is_max = c < F + D;
  if (!(parts & (is_max ? YL_POOL_MAX : YL_POOL_MEAN))) return;
  const int k = is_max ? c - F : c - F - D;
  const float* src = (is_max ? feats : fsup) + k;
  // rows are read 8 at a time (independent loads in flight) — a one-row-per-iteration loop is a chain of
  // dependent L2 round trips; the reduction itself still runs in row order
  float best = 0.f, s = 0.f;
  bool any = false;
  int r = r0;
  for (; r + 8 <= r1; r += 8) {
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = src[(long)(r + j) * ld];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      if (!any || v[j] > best) { best = v[j]; any = true; }
      s += v[j];
    }
  }
  for (; r < r1; ++r) {
    const float v = src[(long)r * ld];
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

int yl_pool_prepare_parts(const float* feats, const float* fsup, int64_t ld, int64_t D, int64_t F, const int32_t* seg_ptr,
                          int64_t P, float* Z, int64_t ldz, int parts, yolat_stream_t stream) {
  if (P <= 0 || D <= 0 || F <= 0 || !feats || !seg_ptr || !Z || ld < D || ldz < 2 * (F + D))
    return YOLAT_E_INVALID;
  for (int64_t p0 = 0; p0 < P; p0 += 65535) {
    const int64_t np = (P - p0) < 65535 ? (P - p0) : 65535;
    hipLaunchKernelGGL(k_pool_prepare, dim3(yl_cdiv(F + (fsup ? 2 : 1) * D, 256), (unsigned)np), dim3(256), 0,
                       (hipStream_t)stream, feats, fsup, (long)ld, (int)D, (int)F, seg_ptr + p0, Z + p0 * ldz,
                       (long)ldz, parts);
    YL_LAUNCH_CHECK();
  }
  return 0;
}

extern "C" int yolat_pool_prepare(const float* feats, const float* fsup, int64_t ld, int64_t D, int64_t F,
                                  const int32_t* seg_ptr, int64_t P, float* Z, int64_t ldz,
                                  yolat_stream_t stream) {
  return yl_pool_prepare_parts(feats, fsup, ld, D, F, seg_ptr, P, Z, ldz, YL_POOL_ZERO | YL_POOL_MAX | YL_POOL_MEAN,
                               stream);
}

static int seg_args_ok(const float* X, int64_t ldx, int64_t D, const float* s, const float* b,
                       const int32_t* seg_ptr, int64_t P, float* Y, int64_t ldy) {
  if (P < 0 || D <= 0 || ldx < D || ldy < D) return 0;
  if (P > 0 && (!X || !seg_ptr || !Y)) return 0;
  if ((s == nullptr) != (b == nullptr)) return 0;
  if (P > 65535LL * 1) { /* grid.y limit handled by caller loop */ }
  return 1;
}

extern "C" int yolat_segment_mean_fwd(const float* X, int64_t ldx, int64_t D, const float* x_scale,
                                      const float* x_shift, int x_relu, const int32_t* seg_ptr,
                                      int64_t P, float* Y, int64_t ldy, yolat_stream_t stream) {
  if (!seg_args_ok(X, ldx, D, x_scale, x_shift, seg_ptr, P, Y, ldy)) return YOLAT_E_INVALID;
  for (int64_t p0 = 0; p0 < P; p0 += 65535) {
    const int64_t np = (P - p0) < 65535 ? (P - p0) : 65535;
    hipLaunchKernelGGL(k_segment_mean_fwd, dim3(yl_cdiv(D, 256), (unsigned)np), dim3(256), 0,
                       (hipStream_t)stream, X, (long)ldx, (int)D, x_scale, x_shift, x_relu,
                       seg_ptr + p0, Y + p0 * ldy, (long)ldy);
    YL_LAUNCH_CHECK();
  }
  return 0;
}

extern "C" int yolat_segment_max_fwd(const float* X, int64_t ldx, int64_t D, const float* x_scale,
                                     const float* x_shift, int x_relu, const int32_t* seg_ptr,
                                     int64_t P, int64_t N, float* Y, int64_t ldy, int32_t* arg,
                                     yolat_stream_t stream) {
  if (!seg_args_ok(X, ldx, D, x_scale, x_shift, seg_ptr, P, Y, ldy)) return YOLAT_E_INVALID;
  for (int64_t p0 = 0; p0 < P; p0 += 65535) {
    const int64_t np = (P - p0) < 65535 ? (P - p0) : 65535;
    hipLaunchKernelGGL(k_segment_max_fwd, dim3(yl_cdiv(D, 256), (unsigned)np), dim3(256), 0,
                       (hipStream_t)stream, X, (long)ldx, (int)D, x_scale, x_shift, x_relu,
                       seg_ptr + p0, (int)N, Y + p0 * ldy, (long)ldy, arg ? arg + p0 * D : nullptr);
    YL_LAUNCH_CHECK();
  }
  return 0;
}

// ---- backward: elementwise over [N, D]; threads along columns, 4 rows per block iteration
__global__ void __launch_bounds__(256) k_segment_mean_bwd(const float* dY, long lddy, int D,
                                                          const int* seg_ptr, const int* node_seg,
                                                          long N, float* dX, long lddx) {
  const int c = blockIdx.x * 64 + (threadIdx.x & 63);
  if (c >= D) return;
  for (long r = (long)blockIdx.y * 4 + (threadIdx.x >> 6); r < N; r += (long)gridDim.y * 4) {
    const int p = node_seg[r];
    const int cnt = seg_ptr[p + 1] - seg_ptr[p];
    dX[r * lddx + c] = dY[(long)p * lddy + c] / (float)(cnt > 1 ? cnt : 1);
  }
}

__global__ void __launch_bounds__(256) k_segment_max_bwd(const float* dY, long lddy, int D,
                                                         const int* arg, const int* node_seg,
                                                         long N, float* dX, long lddx) {
  const int c = blockIdx.x * 64 + (threadIdx.x & 63);
  if (c >= D) return;
  for (long r = (long)blockIdx.y * 4 + (threadIdx.x >> 6); r < N; r += (long)gridDim.y * 4) {
    const int p = node_seg[r];
    dX[r * lddx + c] = (arg[(long)p * D + c] == (int)r) ? dY[(long)p * lddy + c] : 0.f;
  }
}

// D % 128 == 0, aligned rows: 32 lanes x float4 per row (a wave writes 2 rows = 1 KiB per store)
__global__ void __launch_bounds__(256) k_segment_mean_bwd_v4(const float* __restrict__ dY, long lddy, int D,
                                                             const int* __restrict__ seg_ptr,
                                                             const int* __restrict__ node_seg, long N,
                                                             float* __restrict__ dX, long lddx) {
  const int c = blockIdx.x * 128 + (threadIdx.x & 31) * 4;
  for (long r = (long)blockIdx.y * 8 + (threadIdx.x >> 5); r < N; r += (long)gridDim.y * 8) {
    const int p = node_seg[r];
    const int cnt = seg_ptr[p + 1] - seg_ptr[p];
    const float d = (float)(cnt > 1 ? cnt : 1);
    const float4 g = *reinterpret_cast<const float4*>(dY + (long)p * lddy + c);
    *reinterpret_cast<float4*>(dX + r * lddx + c) = make_float4(g.x / d, g.y / d, g.z / d, g.w / d);
  }
}

__global__ void __launch_bounds__(256) k_segment_max_bwd_v4(const float* __restrict__ dY, long lddy, int D,
                                                            const int* __restrict__ arg,
                                                            const int* __restrict__ node_seg, long N,
                                                            float* __restrict__ dX, long lddx) {
  const int c = blockIdx.x * 128 + (threadIdx.x & 31) * 4;
  for (long r = (long)blockIdx.y * 8 + (threadIdx.x >> 5); r < N; r += (long)gridDim.y * 8) {
    const int p = node_seg[r];
    const int4 a = *reinterpret_cast<const int4*>(arg + (long)p * D + c);
    const float4 g = *reinterpret_cast<const float4*>(dY + (long)p * lddy + c);
    const int ri = (int)r;
    *reinterpret_cast<float4*>(dX + r * lddx + c) =
        make_float4(a.x == ri ? g.x : 0.f, a.y == ri ? g.y : 0.f, a.z == ri ? g.z : 0.f, a.w == ri ? g.w : 0.f);
  }
}

extern "C" int yolat_segment_mean_bwd(const float* dY, int64_t lddy, int64_t D,
                                      const int32_t* seg_ptr, const int32_t* node_seg, int64_t N,
                                      float* dX, int64_t lddx, yolat_stream_t stream) {
  if (N < 0 || D <= 0 || lddx < D || lddy < D) return YOLAT_E_INVALID;
  if (N == 0) return 0;
  if (!dY || !seg_ptr || !node_seg || !dX) return YOLAT_E_INVALID;
  int gy = yl_cdiv(N, 4);
  if (gy > 4096) gy = 4096;
  if (D % 128 == 0 && lddy % 4 == 0 && lddx % 4 == 0 && yl_aligned16(dY) && yl_aligned16(dX))
    hipLaunchKernelGGL(k_segment_mean_bwd_v4, dim3(D / 128, gy), dim3(256), 0, (hipStream_t)stream, dY, (long)lddy, (int)D,
                       seg_ptr, node_seg, (long)N, dX, (long)lddx);
  else
  hipLaunchKernelGGL(k_segment_mean_bwd, dim3(yl_cdiv(D, 64), gy), dim3(256), 0,
                     (hipStream_t)stream, dY, (long)lddy, (int)D, seg_ptr, node_seg, (long)N, dX,
                     (long)lddx);
  YL_LAUNCH_CHECK();
  return 0;
}

extern "C" int yolat_segment_max_bwd(const float* dY, int64_t lddy, int64_t D, const int32_t* arg,
                                     const int32_t* node_seg, int64_t N, float* dX, int64_t lddx,
                                     yolat_stream_t stream) {
  if (N < 0 || D <= 0 || lddx < D || lddy < D) return YOLAT_E_INVALID;
  if (N == 0) return 0;
  if (!dY || !arg || !node_seg || !dX) return YOLAT_E_INVALID;
  int gy = yl_cdiv(N, 4);
  if (gy > 4096) gy = 4096;
  if (D % 128 == 0 && lddy % 4 == 0 && lddx % 4 == 0 && yl_aligned16(dY) && yl_aligned16(dX) && yl_aligned16(arg))
    hipLaunchKernelGGL(k_segment_max_bwd_v4, dim3(D / 128, gy), dim3(256), 0, (hipStream_t)stream, dY, (long)lddy, (int)D,
                       arg, node_seg, (long)N, dX, (long)lddx);
  else
  hipLaunchKernelGGL(k_segment_max_bwd, dim3(yl_cdiv(D, 64), gy), dim3(256), 0,
                     (hipStream_t)stream, dY, (long)lddy, (int)D, arg, node_seg, (long)N, dX,
                     (long)lddx);
  YL_LAUNCH_CHECK();
  return 0;
}
