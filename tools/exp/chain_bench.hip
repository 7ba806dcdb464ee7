// Dev experiment: k_conv_chain timing at cfg-2 and cfg-5 sizes (compile with -DCH_RING=n to vary the ring).
#include "../../yolat_vectorgraphicsrecognition_amd/csrc/conv_chain.hip"
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include <algorithm>
int main() {
  struct Cfg { int N, E; } cfgs[] = {{10000, 40000}, {200000, 1200000}};
  for (auto c : cfgs) for (int Cin : {64, 5}) {
    const int N = c.N, E = c.E, C = 64;
    std::vector<int> dst(E), src(E), rp(N + 1, 0);
    srand(1);
    for (int e = 0; e < E; ++e) { int p = rand() % (N / 25); dst[e] = p * 25 + rand() % 25; src[e] = p * 25 + rand() % 25; }
    std::vector<int> order(E);
    for (int e = 0; e < E; ++e) order[e] = e;
    std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return dst[a] < dst[b]; });
    std::vector<int> d2(E), s2(E);
    for (int e = 0; e < E; ++e) { d2[e] = dst[order[e]]; s2[e] = src[order[e]]; rp[d2[e] + 1]++; }
    for (int i = 0; i < N; ++i) rp[i + 1] += rp[i];
    float *x, *attr, *w, *fo, *so; int *drp, *ds, *dd;
    hipMalloc(&x, (size_t)N * 64 * 4); hipMalloc(&attr, (size_t)E * 16); hipMalloc(&w, 1 << 20);
    hipMalloc(&fo, (size_t)N * 64 * 4); hipMalloc(&so, (size_t)N * 64 * 4);
    hipMalloc(&drp, (N + 1) * 4); hipMalloc(&ds, E * 4); hipMalloc(&dd, E * 4);
    hipMemset(x, 0, (size_t)N * 64 * 4); hipMemset(attr, 0, (size_t)E * 16); hipMemset(w, 0, 1 << 20);
    hipMemcpy(drp, rp.data(), (N + 1) * 4, hipMemcpyHostToDevice);
    hipMemcpy(ds, s2.data(), E * 4, hipMemcpyHostToDevice); hipMemcpy(dd, d2.data(), E * 4, hipMemcpyHostToDevice);
    yolat_conv_eval cv; cv.Cin = Cin; float* pk; hipMalloc(&pk, yolat_conv_pack_elems(Cin) * 4); hipMemset(pk, 0, yolat_conv_pack_elems(Cin) * 4);
    cv.W1 = w; cv.b1 = w; cv.s1 = w; cv.t1 = w; cv.W2 = w; cv.b2 = w; cv.s2 = w; cv.t2 = w; cv.Wr = w; cv.br = w; cv.Wn = w; cv.bn = w; cv.sn = w; cv.tn = w;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 3; ++i) yolat_conv_eval_chain(x, Cin, x, Cin, N, Cin, drp, ds, dd, attr, E, &cv, pk, C, fo, 64, so, 64, 0);
    hipEventRecord(e0);
    const int it = 20;
    for (int i = 0; i < it; ++i) yolat_conv_eval_chain(x, Cin, x, Cin, N, Cin, drp, ds, dd, attr, E, &cv, pk, C, fo, 64, so, 64, 0);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double us = ms * 1e3 / it, fl = 2.0 * E * ((2.0 * Cin + 4) * 64 + 64 * 64 + 32 * 64) ;
    printf("RING=%d N=%7d E=%8d Cin=%2d  %9.2f us  %6.1f TF/s (incl. agg MFMA)  %6.2f ns/edge\n", CH_RING, N, E, Cin, us, fl / us * 1e-6, us * 1e3 / E);
    hipFree(x); hipFree(attr); hipFree(w); hipFree(fo); hipFree(so); hipFree(drp); hipFree(ds); hipFree(dd);
  }
  return 0;
}
