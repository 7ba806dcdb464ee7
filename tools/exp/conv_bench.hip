// Dev experiment: phase ablation of k_conv_fused_eval (dbg bits: 1 skip chunk loop, 2 skip weight staging,
// 4 skip root/node, 8 gather only).
#include "../../yolat_vectorgraphicsrecognition_amd/csrc/conv_fused.hip"
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include <algorithm>
int main() {
  const int N = 10000, E = 40000, C = 64;
  for (int Cin : {64, 5}) {
    std::vector<int> dst(E), src(E), rp(N + 1, 0);
    srand(1);
    for (int e = 0; e < E; ++e) { dst[e] = rand() % N; src[e] = rand() % N; }
    std::sort(dst.begin(), dst.end());
    for (int e = 0; e < E; ++e) rp[dst[e] + 1]++;
    for (int i = 0; i < N; ++i) rp[i + 1] += rp[i];
    float *x, *attr, *w, *fo, *so; int *drp, *ds, *dd;
    hipMalloc(&x, N * 64 * 4); hipMalloc(&attr, E * 16); hipMalloc(&w, 1 << 20); hipMalloc(&fo, N * 64 * 4); hipMalloc(&so, N * 64 * 4);
    hipMalloc(&drp, (N + 1) * 4); hipMalloc(&ds, E * 4); hipMalloc(&dd, E * 4);
    hipMemset(x, 0, N * 64 * 4); hipMemset(attr, 0, E * 16); hipMemset(w, 0, 1 << 20);
    hipMemcpy(drp, rp.data(), (N + 1) * 4, hipMemcpyHostToDevice);
    hipMemcpy(ds, src.data(), E * 4, hipMemcpyHostToDevice); hipMemcpy(dd, dst.data(), E * 4, hipMemcpyHostToDevice);
    yolat_conv_eval cv; cv.Cin = Cin;
    cv.W1 = w; cv.b1 = w; cv.s1 = w; cv.t1 = w; cv.W2 = w; cv.b2 = w; cv.s2 = w; cv.t2 = w; cv.Wr = w; cv.br = w; cv.Wn = w; cv.bn = w; cv.sn = w; cv.tn = w;
    for (int dbg : {0, 1, 8, 16, 32, 64, 112}) {
      g_conv_dbg = dbg;
      hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
      for (int i = 0; i < 3; ++i) yolat_conv_eval_fused(x, Cin, x, Cin, N, Cin, drp, ds, dd, attr, E, &cv, C, fo, 64, so, 64, 0);
      hipEventRecord(e0);
      for (int i = 0; i < 20; ++i) yolat_conv_eval_fused(x, Cin, x, Cin, N, Cin, drp, ds, dd, attr, E, &cv, C, fo, 64, so, 64, 0);
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      printf("Cin=%2d dbg=%2d  %8.2f us\n", Cin, dbg, ms * 1e3f / 20);
    }
  }
  return 0;
}
