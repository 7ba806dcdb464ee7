// Dev experiment: how does the K=128 fusion GEMM respond to the number of resident workgroups per CU?
// Extra dynamic LDS at launch caps residency without touching the kernel.
#include "../../yolat_vectorgraphicsrecognition_amd/csrc/common.hpp"
#include <stdio.h>
#include <stdlib.h>
#include <vector>

template <int BM, int BN, int BK>
static void run(int M, int N, int K, int iters, float* A, float* W, float* Y, int extra_lds) {
  DenseOp a = yl_dense(A, K, M, K), b = yl_dense(W, K, N, K);
  Epilogue ep; ep.bias = nullptr; ep.scale = nullptr; ep.shift = nullptr; ep.relu = 0; ep.Y = Y; ep.ldy = N;
  ep.accumulate = 0; ep.stats = nullptr; ep.seg = nullptr; ep.pool = nullptr; ep.ldpool = 0;
  dim3 grid((M + BM - 1) / BM, (N + BN - 1) / BN);
  hipFuncSetAttribute(reinterpret_cast<const void*>(k_gemm_nt<BM, BN, BK, DenseOp, DenseOp, false>),
                      hipFuncAttributeMaxDynamicSharedMemorySize, 140 * 1024);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((k_gemm_nt<BM, BN, BK, DenseOp, DenseOp, false>), grid, dim3(256), extra_lds, 0, a, b, ep, M, N, K);
  hipEventRecord(e0);
  for (int i = 0; i < iters; ++i) hipLaunchKernelGGL((k_gemm_nt<BM, BN, BK, DenseOp, DenseOp, false>), grid, dim3(256), extra_lds, 0, a, b, ep, M, N, K);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  float us = ms * 1e3f / iters;
  const int lds = (BM + BN) * (BK + 1) * 4 + extra_lds;
  printf("M=%6d N=%5d K=%4d tile %3dx%3dx%2d grid %6d  lds/WG %6d B (<= %d WG/CU)  %8.2f us  %7.2f TF/s\n", M, N, K, BM, BN, BK,
         grid.x * grid.y, lds, 160 * 1024 / lds, us, 2.0 * M * N * K / us * 1e-6);
}

int main() {
  const int MAXM = 1 << 18;
  float *A, *W, *Y;
  hipMalloc(&A, (size_t)MAXM * 128 * 4); hipMalloc(&W, (size_t)1024 * 128 * 4); hipMalloc(&Y, (size_t)MAXM * 1024 * 4);
  std::vector<float> h((size_t)MAXM * 128);
  for (auto& v : h) v = (rand() % 2001 - 1000) * 1e-3f;
  hipMemcpy(A, h.data(), h.size() * 4, hipMemcpyHostToDevice);
  hipMemcpy(W, h.data(), (size_t)1024 * 128 * 4, hipMemcpyHostToDevice);
  const int extras[] = {0, 4096, 10 * 1024, 16 * 1024, 24 * 1024, 37 * 1024, 64 * 1024};
  for (int M : {10000, 200000}) {
    for (int e : extras) run<64, 64, 32>(M, 1024, 128, M > 50000 ? 5 : 30, A, W, Y, e);
    for (int e : {0, 20 * 1024, 46 * 1024}) run<128, 128, 32>(M, 1024, 128, M > 50000 ? 5 : 30, A, W, Y, e);
  }
  return 0;
}
