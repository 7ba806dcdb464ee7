// Dev experiment (NOT product code): fp32 GEMM emulated with bf16 MFMAs on pre-split operands.
//   a = a_h + a_l (+ a_m), each piece a bf16;  a*b ~= a_h*b_h + a_h*b_l + a_l*b_h   (bf16x3: ~2^-16 relative)
// v_mfma_f32_32x32x16_bf16 runs at 16x the rate of v_mfma_f32_32x32x2_f32, so 3 (or 6) of them per fp32 product are
// still 5.3x (2.7x) faster on the matrix pipe.  Measures time and accuracy on the fusion-GEMM shape.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <math.h>
#include <vector>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef unsigned short u16;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;

__global__ void k_split(const float* __restrict__ X, long n, u16* H, u16* L) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float x = X[i];
  const unsigned hb = __float_as_uint(x) & 0xFFFF0000u;            // truncation: the remainder is exact
  const float r = x - __uint_as_float(hb);
  const unsigned lb = __float_as_uint(r) & 0xFFFF0000u;
  H[i] = (u16)(hb >> 16);
  L[i] = (u16)(lb >> 16);
}

__global__ void k_split3(const float* __restrict__ X, long n, u16* H, u16* Mi, u16* L) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float x = X[i];
  const unsigned hb = __float_as_uint(x) & 0xFFFF0000u;
  const float r = x - __uint_as_float(hb);
  const unsigned mb = __float_as_uint(r) & 0xFFFF0000u;
  const float r2 = r - __uint_as_float(mb);
  const unsigned lb = __float_as_uint(r2) & 0xFFFF0000u;
  H[i] = (u16)(hb >> 16); Mi[i] = (u16)(mb >> 16); L[i] = (u16)(lb >> 16);
}

// 3-term split, 6 products: hh + hm + mh + hl + lh + mm  (fp32-level accuracy).  Planes are separate [rows][K] bf16
// arrays at stride aps / bps elements.  Block = 2x2 waves, each wave (32*WM) x (32*WN) outputs.
template <int WM, int WN, int K>
__global__ void __launch_bounds__(256) k_gemm_x6(const u16* __restrict__ A, long aps, const u16* __restrict__ B, long bps,
                                                 float* Y, int M, int N, int) {
  constexpr int RS = 40, BM = 64 * WM, BN = 64 * WN;
  __shared__ __attribute__((aligned(16))) u16 sA[3 * BM * RS], sB[3 * BN * RS];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int wm = wave >> 1, wn = wave & 1, l31 = lane & 31, lhi = lane >> 5;
  const int row0 = blockIdx.x * BM, col0 = blockIdx.y * BN;
  f32x16 acc[WM][WN];
#pragma unroll
  for (int i = 0; i < WM; ++i)
#pragma unroll
    for (int j = 0; j < WN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  const int sr = tid >> 2, sq = tid & 3;
  u32x4 ra[3 * WM], rb[3 * WN];
  long aoff[WM], boff[WN];
#pragma unroll
  for (int i = 0; i < WM; ++i) aoff[i] = (long)min(row0 + sr + 64 * i, M - 1) * K + 8 * sq;
#pragma unroll
  for (int j = 0; j < WN; ++j) boff[j] = (long)(col0 + sr + 64 * j) * K + 8 * sq;
#pragma unroll
  for (int p = 0; p < 3; ++p) {
#pragma unroll
    for (int i = 0; i < WM; ++i) ra[p * WM + i] = *reinterpret_cast<const u32x4*>(A + p * aps + aoff[i]);
#pragma unroll
    for (int j = 0; j < WN; ++j) rb[p * WN + j] = *reinterpret_cast<const u32x4*>(B + p * bps + boff[j]);
  }
#pragma unroll
  for (int k0 = 0; k0 < K; k0 += 32) {
#pragma unroll
    for (int p = 0; p < 3; ++p) {
#pragma unroll
      for (int i = 0; i < WM; ++i) *reinterpret_cast<u32x4*>(sA + (p * BM + sr + 64 * i) * RS + 8 * sq) = ra[p * WM + i];
#pragma unroll
      for (int j = 0; j < WN; ++j) *reinterpret_cast<u32x4*>(sB + (p * BN + sr + 64 * j) * RS + 8 * sq) = rb[p * WN + j];
    }
    __syncthreads();
    if (k0 + 32 < K) {
#pragma unroll
      for (int p = 0; p < 3; ++p) {
#pragma unroll
        for (int i = 0; i < WM; ++i) ra[p * WM + i] = *reinterpret_cast<const u32x4*>(A + p * aps + aoff[i] + k0 + 32);
#pragma unroll
        for (int j = 0; j < WN; ++j) rb[p * WN + j] = *reinterpret_cast<const u32x4*>(B + p * bps + boff[j] + k0 + 32);
      }
    }
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      bf16x8 a[3][WM], b[3][WN];
#pragma unroll
      for (int p = 0; p < 3; ++p) {
#pragma unroll
        for (int i = 0; i < WM; ++i)
          a[p][i] = *reinterpret_cast<const bf16x8*>(sA + (p * BM + wm * 32 * WM + 32 * i + l31) * RS + ks * 16 + lhi * 8);
#pragma unroll
        for (int j = 0; j < WN; ++j)
          b[p][j] = *reinterpret_cast<const bf16x8*>(sB + (p * BN + wn * 32 * WN + 32 * j + l31) * RS + ks * 16 + lhi * 8);
      }
#pragma unroll
      for (int i = 0; i < WM; ++i)
#pragma unroll
        for (int j = 0; j < WN; ++j) {
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1][i], b[1][j], acc[i][j], 0, 0, 0);   // mm
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[2][i], b[0][j], acc[i][j], 0, 0, 0);   // lh
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0][i], b[2][j], acc[i][j], 0, 0, 0);   // hl
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1][i], b[0][j], acc[i][j], 0, 0, 0);   // mh
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0][i], b[1][j], acc[i][j], 0, 0, 0);   // hm
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0][i], b[0][j], acc[i][j], 0, 0, 0);   // hh
        }
    }
    __syncthreads();
  }
#pragma unroll
  for (int i = 0; i < WM; ++i)
#pragma unroll
    for (int j = 0; j < WN; ++j) {
      const int col = col0 + wn * 32 * WN + 32 * j + l31;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = row0 + wm * 32 * WM + 32 * i + (r & 3) + 8 * (r >> 2) + 4 * lhi;
        if (row < M) Y[(long)row * N + col] = acc[i][j][r];
      }
    }
}

// 64x64 tile, K step 32; LDS planes [64 rows][32 k] bf16 with an 80-byte row stride (conflict-free b128 reads)
template <int NPROD>   // 3: hh + hl + lh
__global__ void __launch_bounds__(256) k_gemm_x3(const u16* __restrict__ Ah, const u16* __restrict__ Al,
                                                 const u16* __restrict__ Bh, const u16* __restrict__ Bl, float* Y, int M,
                                                 int N, int K) {
  constexpr int RS = 40;                                   // row stride in u16 (80 B)
  __shared__ __attribute__((aligned(16))) u16 sAh[64 * RS], sAl[64 * RS], sBh[64 * RS], sBl[64 * RS];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int wm = wave >> 1, wn = wave & 1, l31 = lane & 31, lhi = lane >> 5;
  const int row0 = blockIdx.x * 64, col0 = blockIdx.y * 64;
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  // staging: 64 rows x 32 k x 2 B = 4 KB per plane = 256 x 16 B: one uint4 per thread per plane
  const int sr = tid >> 2, sq = tid & 3;                   // row, 8-element chunk
  uint4 ra_h, ra_l, rb_h, rb_l;
  auto fetch = [&](int k0) {
    const long ao = (long)min(row0 + sr, M - 1) * K + k0 + 8 * sq;
    const long bo = (long)(col0 + sr) * K + k0 + 8 * sq;
    ra_h = *reinterpret_cast<const uint4*>(Ah + ao); ra_l = *reinterpret_cast<const uint4*>(Al + ao);
    rb_h = *reinterpret_cast<const uint4*>(Bh + bo); rb_l = *reinterpret_cast<const uint4*>(Bl + bo);
  };
  auto stage = [&]() {
    *reinterpret_cast<uint4*>(sAh + sr * RS + 8 * sq) = ra_h; *reinterpret_cast<uint4*>(sAl + sr * RS + 8 * sq) = ra_l;
    *reinterpret_cast<uint4*>(sBh + sr * RS + 8 * sq) = rb_h; *reinterpret_cast<uint4*>(sBl + sr * RS + 8 * sq) = rb_l;
  };
  fetch(0);
  for (int k0 = 0; k0 < K; k0 += 32) {
    stage();
    __syncthreads();
    if (k0 + 32 < K) fetch(k0 + 32);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const int ao = (wm * 32 + l31) * RS + ks * 16 + lhi * 8;
      const int bo = (wn * 32 + l31) * RS + ks * 16 + lhi * 8;
      const bf16x8 ah = *reinterpret_cast<const bf16x8*>(sAh + ao), al = *reinterpret_cast<const bf16x8*>(sAl + ao);
      const bf16x8 bh = *reinterpret_cast<const bf16x8*>(sBh + bo), bl = *reinterpret_cast<const bf16x8*>(sBl + bo);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc, 0, 0, 0);
    }
    __syncthreads();
  }
  const int col = col0 + wn * 32 + l31;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int row = row0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhi;
    if (row < M) Y[(long)row * N + col] = acc[r];
  }
}

int main() {
  const int N = 1024, K = 128;
  for (int M : {10000, 200000}) {
    std::vector<float> hA((size_t)M * K), hW((size_t)N * K);
    srand(1);
    for (auto& v : hA) v = (rand() % 20001 - 10000) * 1e-4f * ((rand() & 3) ? 1.f : 3.7f);
    for (auto& v : hW) v = (rand() % 20001 - 10000) * 1e-4f / 11.3f;
    float *A, *W, *Y; u16 *Ah, *Al, *Wh, *Wl;
    hipMalloc(&A, hA.size() * 4); hipMalloc(&W, hW.size() * 4); hipMalloc(&Y, (size_t)M * N * 4);
    hipMalloc(&Ah, hA.size() * 2); hipMalloc(&Al, hA.size() * 2); hipMalloc(&Wh, hW.size() * 2); hipMalloc(&Wl, hW.size() * 2);
    hipMemcpy(A, hA.data(), hA.size() * 4, hipMemcpyHostToDevice); hipMemcpy(W, hW.data(), hW.size() * 4, hipMemcpyHostToDevice);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k_split, dim3((hW.size() + 255) / 256), dim3(256), 0, 0, W, (long)hW.size(), Wh, Wl);
    hipEventRecord(e0);
    for (int i = 0; i < 10; ++i) hipLaunchKernelGGL(k_split, dim3((hA.size() + 255) / 256), dim3(256), 0, 0, A, (long)hA.size(), Ah, Al);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("M=%d: split of A (M x %d fp32 -> 2 bf16 planes): %.2f us\n", M, K, ms * 100.f);
    dim3 grid((M + 63) / 64, N / 64);
    const int it = M > 50000 ? 5 : 30;
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(k_gemm_x3<3>, grid, dim3(256), 0, 0, Ah, Al, Wh, Wl, Y, M, N, K);
    hipEventRecord(e0);
    for (int i = 0; i < it; ++i) hipLaunchKernelGGL(k_gemm_x3<3>, grid, dim3(256), 0, 0, Ah, Al, Wh, Wl, Y, M, N, K);
    hipEventRecord(e1); hipEventSynchronize(e1);
    hipEventElapsedTime(&ms, e0, e1);
    const float us = ms * 1e3f / it;
    printf("M=%d: bf16x3 GEMM 64x64x32 tiles: %.2f us = %.1f fp32-equivalent TFLOP/s\n", M, us, 2.0 * M * N * K / us * 1e-6);
    // accuracy on a sample of rows vs float64
    std::vector<float> hY((size_t)64 * N);
    hipMemcpy(hY.data(), Y + (size_t)(M - 64) * N, hY.size() * 4, hipMemcpyDeviceToHost);
    double maxerr = 0, maxref = 0, sumsq = 0, sumref = 0;
    for (int r = 0; r < 64; ++r)
      for (int c = 0; c < N; ++c) {
        double ref = 0;
        for (int k = 0; k < K; ++k) ref += (double)hA[(size_t)(M - 64 + r) * K + k] * (double)hW[(size_t)c * K + k];
        const double d = fabs(ref - hY[(size_t)r * N + c]);
        maxerr = fmax(maxerr, d); maxref = fmax(maxref, fabs(ref)); sumsq += d * d; sumref += ref * ref;
      }
    printf("M=%d: max abs err %.3e, max |ref| %.3e -> %.2e of scale;  rms err / rms ref = %.2e\n", M, maxerr, maxref,
           maxerr / maxref, sqrt(sumsq / sumref));
    {
      u16 *A3, *W3;
      const long aps = (long)hA.size(), bps = (long)hW.size();
      hipMalloc(&A3, aps * 6); hipMalloc(&W3, bps * 6);
      hipLaunchKernelGGL(k_split3, dim3((bps + 255) / 256), dim3(256), 0, 0, W, bps, W3, W3 + bps, W3 + 2 * bps);
      hipEventRecord(e0);
      for (int i = 0; i < 10; ++i) hipLaunchKernelGGL(k_split3, dim3((aps + 255) / 256), dim3(256), 0, 0, A, aps, A3, A3 + aps, A3 + 2 * aps);
      hipEventRecord(e1); hipEventSynchronize(e1);
      hipEventElapsedTime(&ms, e0, e1);
      printf("M=%d: 3-plane split of A: %.2f us\n", M, ms * 100.f);
      auto run = [&](auto kern, int BM, int BN, const char* name) {
        dim3 g((M + BM - 1) / BM, N / BN);
        for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(kern, g, dim3(256), 0, 0, A3, aps, W3, bps, Y, M, N, K);
        hipEventRecord(e0);
        for (int i = 0; i < it; ++i) hipLaunchKernelGGL(kern, g, dim3(256), 0, 0, A3, aps, W3, bps, Y, M, N, K);
        hipEventRecord(e1); hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
        const float us6 = ms * 1e3f / it;
        hipMemcpy(hY.data(), Y + (size_t)(M - 64) * N, hY.size() * 4, hipMemcpyDeviceToHost);
        double me = 0, mr = 0, ss = 0, sr2 = 0;
        for (int r = 0; r < 64; ++r)
          for (int c = 0; c < N; ++c) {
            double ref = 0;
            for (int k = 0; k < K; ++k) ref += (double)hA[(size_t)(M - 64 + r) * K + k] * (double)hW[(size_t)c * K + k];
            const double d = fabs(ref - hY[(size_t)r * N + c]);
            me = fmax(me, d); mr = fmax(mr, fabs(ref)); ss += d * d; sr2 += ref * ref;
          }
        printf("M=%d: bf16x6 GEMM %s tiles: %.2f us = %.1f fp32-equivalent TFLOP/s; max err %.2e of scale, rms %.2e\n", M, name,
               us6, 2.0 * M * N * K / us6 * 1e-6, me / mr, sqrt(ss / sr2));
      };
      run(k_gemm_x6<1, 1, 128>, 64, 64, "64x64");
      run(k_gemm_x6<2, 1, 128>, 128, 64, "128x64");
      run(k_gemm_x6<1, 2, 128>, 64, 128, "64x128");
      run(k_gemm_x6<2, 2, 128>, 128, 128, "128x128");
      hipFree(A3); hipFree(W3);
    }
    hipFree(A); hipFree(W); hipFree(Y); hipFree(Ah); hipFree(Al); hipFree(Wh); hipFree(Wl);
  }
  return 0;
}
