// Dev experiment: BK sweep for the K=128 fusion GEMM shape (plain-store epilogue).
#include "../../yolat_vectorgraphicsrecognition_amd/csrc/common.hpp"
#include <stdio.h>
#include <stdlib.h>
#include <vector>
template <int BM, int BN, int BK>
static void run(int M, int N, int K, int iters, float* A, float* W, float* Y) {
  DenseOp a = yl_dense(A, K, M, K), b = yl_dense(W, K, N, K);
  Epilogue ep; ep.bias = nullptr; ep.scale = nullptr; ep.shift = nullptr; ep.relu = 0; ep.Y = Y; ep.ldy = N;
  ep.accumulate = 0; ep.stats = nullptr; ep.seg = nullptr; ep.pool = nullptr; ep.ldpool = 0;
  dim3 grid((M + BM - 1) / BM, (N + BN - 1) / BN);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((k_gemm_nt<BM, BN, BK, DenseOp, DenseOp, false>), grid, dim3(256), 0, 0, a, b, ep, M, N, K);
  hipEventRecord(e0);
  for (int i = 0; i < iters; ++i) hipLaunchKernelGGL((k_gemm_nt<BM, BN, BK, DenseOp, DenseOp, false>), grid, dim3(256), 0, 0, a, b, ep, M, N, K);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  float us = ms * 1e3f / iters;
  printf("M=%6d N=%5d K=%4d tile %3dx%3dx%2d grid %6d  %8.2f us  %7.2f TF/s\n", M, N, K, BM, BN, BK, grid.x * grid.y, us,
         2.0 * M * N * K / us * 1e-6);
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
    run<64, 64, 16>(M, 1024, 128, it, A, W, Y);
    run<64, 64, 32>(M, 1024, 128, it, A, W, Y);
    run<64, 64, 64>(M, 1024, 128, it, A, W, Y);
    run<64, 64, 128>(M, 1024, 128, it, A, W, Y);
    run<128, 64, 64>(M, 1024, 128, it, A, W, Y);
  }
  return 0;
}
