// Dev experiment: which resource saturates the K=128 64x64x32 NT GEMM?  MODE bits remove one
// component at a time (results are then wrong on purpose; only the time is of interest).
//  1: no global loads after the first k-step   2: no LDS fragment reads (MFMA on fixed registers)
//  4: no epilogue stores                        8: no LDS staging writes
//  16: no MFMA
#include "../../yolat_vectorgraphicsrecognition_amd/csrc/common.hpp"
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include <math.h>

template <int MODE>
__global__ void __launch_bounds__(256) k_exp(const float* __restrict__ A, const float* __restrict__ W, float* Y,
                                             int M, int N, int K) {
  constexpr int BM = 64, BN = 64, BK = 32, LD = BK + 1, KQ = BK / 4;
  __shared__ float As[BM * LD];
  __shared__ float Bs[BN * LD];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int wm = wave >> 1, wn = wave & 1, l31 = lane & 31, lhi = lane >> 5;
  int rt_, ct_;
  yl_xcd_tile(rt_, ct_);
  const int row0 = rt_ * BM, col0 = ct_ * BN;
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  float4 ra[2], rb[2];
  auto fetch = [&](int k0) {
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const int i = tid + t * 256;
      const int r = min(row0 + i / KQ, M - 1);
      ra[t] = *reinterpret_cast<const float4*>(A + (long)r * K + k0 + 4 * (i % KQ));
      rb[t] = *reinterpret_cast<const float4*>(W + (long)(col0 + i / KQ) * K + k0 + 4 * (i % KQ));
    }
  };
  auto stage = [&]() {
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const int i = tid + t * 256;
      float* d = As + (i / KQ) * LD + 4 * (i % KQ);
      d[0] = ra[t].x; d[1] = ra[t].y; d[2] = ra[t].z; d[3] = ra[t].w;
      float* e = Bs + (i / KQ) * LD + 4 * (i % KQ);
      e[0] = rb[t].x; e[1] = rb[t].y; e[2] = rb[t].z; e[3] = rb[t].w;
    }
  };
  fetch(0);
  for (int k0 = 0; k0 < K; k0 += BK) {
    if (!(MODE & 8) || k0 == 0) stage();
    __syncthreads();
    if (k0 + BK < K && !(MODE & 1)) fetch(k0 + BK);
    float fa = ra[0].x, fb = rb[0].y;
#pragma unroll
    for (int kk = 0; kk < BK; kk += 2) {
      float a, b;
      if (MODE & 2) { a = fa; b = fb; fa += 1.f; }
      else { a = As[(wm * 32 + l31) * LD + kk + lhi]; b = Bs[(wn * 32 + l31) * LD + kk + lhi]; }
      if (MODE & 16) acc[kk & 15] += a * b;
      else acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
    }
    __syncthreads();
  }
  const int col = col0 + wn * 32 + l31;
  if (MODE & 4) {
    float s = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) s += acc[r];
    if (s == 1234.5678f) Y[tid] = s;
  } else {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = row0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhi;
      if (row < M) Y[(long)row * N + col] = acc[r];
    }
  }
}

// v2: k-parity-interleaved LDS layout (fragment reads as ds_read_b128 covering 4 MFMA steps, staging as
// ds_write_b64) + double-buffered LDS (one barrier per k-step).  NOSTORE: skip the epilogue stores.
template <bool NOSTORE>
__global__ void __launch_bounds__(256) k_exp2(const float* __restrict__ A, const float* __restrict__ W, float* Y,
                                              int M, int N, int K) {
  constexpr int BK = 32, LDR = 36;                       // row = [parity 0: 16 floats | parity 1: 16 floats | pad 4]
  __shared__ __attribute__((aligned(16))) float As[2][64 * LDR];
  __shared__ __attribute__((aligned(16))) float Bs[2][64 * LDR];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int wm = wave >> 1, wn = wave & 1, l31 = lane & 31, lhi = lane >> 5;
  int rt_, ct_;
  yl_xcd_tile(rt_, ct_);
  const int row0 = rt_ * 64, col0 = ct_ * 64;
  // staging map: lanes 0-7 -> row a, 8-15 -> row a+4 (bank-disjoint b64 writes), 16 lanes per pair of rows
  const int sq = lane & 7;
  const int srow = wave * 8 + ((lane >> 3) & 1) * 4 + (lane >> 4);      // + 32 for the second slot
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  float4 ra[2], rb[2];
  auto fetch = [&](int k0) {
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const int r = srow + 32 * t;
      ra[t] = *reinterpret_cast<const float4*>(A + (long)min(row0 + r, M - 1) * K + k0 + 4 * sq);
      rb[t] = *reinterpret_cast<const float4*>(W + (long)(col0 + r) * K + k0 + 4 * sq);
    }
  };
  auto stage = [&](int buf) {
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const int r = srow + 32 * t;
      float* a = As[buf] + r * LDR + 2 * sq;
      *reinterpret_cast<float2*>(a) = make_float2(ra[t].x, ra[t].z);
      *reinterpret_cast<float2*>(a + 16) = make_float2(ra[t].y, ra[t].w);
      float* b = Bs[buf] + r * LDR + 2 * sq;
      *reinterpret_cast<float2*>(b) = make_float2(rb[t].x, rb[t].z);
      *reinterpret_cast<float2*>(b + 16) = make_float2(rb[t].y, rb[t].w);
    }
  };
  fetch(0);
  stage(0);
  if (BK < K) fetch(BK);
  __syncthreads();
  int buf = 0;
  for (int k0 = 0; k0 < K; k0 += BK) {
    if (k0 + BK < K) stage(buf ^ 1);
    if (k0 + 2 * BK < K) fetch(k0 + 2 * BK);
    const float* ap = As[buf] + (wm * 32 + l31) * LDR + lhi * 16;
    const float* bp = Bs[buf] + (wn * 32 + l31) * LDR + lhi * 16;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const float4 a4 = *reinterpret_cast<const float4*>(ap + 4 * g);
      const float4 b4 = *reinterpret_cast<const float4*>(bp + 4 * g);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.x, b4.x, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.y, b4.y, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.z, b4.z, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.w, b4.w, acc, 0, 0, 0);
    }
    __syncthreads();
    buf ^= 1;
  }
  const int col = col0 + wn * 32 + l31;
  if (NOSTORE) {
    float s = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) s += acc[r];
    if (s == 1234.5678f) Y[tid] = s;
  } else {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = row0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhi;
      if (row < M) Y[(long)row * N + col] = acc[r];
    }
  }
}

template <bool NOSTORE>
static void run2(int M, int N, int K, int iters, float* A, float* W, float* Y) {
  dim3 grid((M + 63) / 64, N / 64);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(k_exp2<NOSTORE>, grid, dim3(256), 0, 0, A, W, Y, M, N, K);
  hipEventRecord(e0);
  for (int i = 0; i < iters; ++i) hipLaunchKernelGGL(k_exp2<NOSTORE>, grid, dim3(256), 0, 0, A, W, Y, M, N, K);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  float us = ms * 1e3f / iters;
  printf("M=%6d v2 (b128 fragments, double-buffered LDS)%s  %8.2f us  %7.2f TF/s-equivalent\n", M, NOSTORE ? " nostore" : "",
         us, 2.0 * M * N * K / us * 1e-6);
}

template <int MODE>
static void run(int M, int N, int K, int iters, float* A, float* W, float* Y) {
  dim3 grid((M + 63) / 64, N / 64);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(k_exp<MODE>, grid, dim3(256), 0, 0, A, W, Y, M, N, K);
  hipEventRecord(e0);
  for (int i = 0; i < iters; ++i) hipLaunchKernelGGL(k_exp<MODE>, grid, dim3(256), 0, 0, A, W, Y, M, N, K);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  float us = ms * 1e3f / iters;
  printf("M=%6d mode %2d [%s%s%s%s%s]  %8.2f us  %7.2f TF/s-equivalent\n", M, MODE, (MODE & 1) ? "noglobal " : "",
         (MODE & 2) ? "noldsread " : "", (MODE & 4) ? "nostore " : "", (MODE & 8) ? "noldswrite " : "",
         (MODE & 16) ? "nomfma " : "", us, 2.0 * M * N * K / us * 1e-6);
}

int main() {
  const int MAXM = 1 << 18;
  float *A, *W, *Y;
  hipMalloc(&A, (size_t)MAXM * 128 * 4); hipMalloc(&W, (size_t)1024 * 128 * 4); hipMalloc(&Y, (size_t)MAXM * 1024 * 4);
  std::vector<float> h((size_t)MAXM * 128);
  for (auto& v : h) v = (rand() % 2001 - 1000) * 1e-3f;
  hipMemcpy(A, h.data(), h.size() * 4, hipMemcpyHostToDevice);
  hipMemcpy(W, h.data(), (size_t)1024 * 128 * 4, hipMemcpyHostToDevice);
  for (int M : {10000, 200000}) {
    const int it = M > 50000 ? 5 : 30;
    run<0>(M, 1024, 128, it, A, W, Y);
    {
      std::vector<float> y0((size_t)4096), y1((size_t)4096);
      hipMemcpy(y0.data(), Y + (size_t)(M - 5) * 1024, 4096 * 4, hipMemcpyDeviceToHost);
      run2<false>(M, 1024, 128, it, A, W, Y);
      hipMemcpy(y1.data(), Y + (size_t)(M - 5) * 1024, 4096 * 4, hipMemcpyDeviceToHost);
      double md = 0; for (int i = 0; i < 4096; ++i) md = fmax(md, fabs((double)y0[i] - y1[i]));
      printf("   v2 vs v1 max abs diff on the last rows: %g\n", md);
      run2<true>(M, 1024, 128, it, A, W, Y);
    }
    run<1>(M, 1024, 128, it, A, W, Y);
    run<2>(M, 1024, 128, it, A, W, Y);
    run<4>(M, 1024, 128, it, A, W, Y);
    run<8>(M, 1024, 128, it, A, W, Y);
    run<16>(M, 1024, 128, it, A, W, Y);
    run<1 | 4>(M, 1024, 128, it, A, W, Y);
    run<1 | 2 | 4 | 8>(M, 1024, 128, it, A, W, Y);
    run<2 | 8>(M, 1024, 128, it, A, W, Y);
    run<1 | 2 | 8>(M, 1024, 128, it, A, W, Y);
  }
  return 0;
}
