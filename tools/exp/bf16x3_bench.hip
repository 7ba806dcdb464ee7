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
    hipFree(A); hipFree(W); hipFree(Y); hipFree(Ah); hipFree(Al); hipFree(Wh); hipFree(Wl);
  }
  return 0;
}
