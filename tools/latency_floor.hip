// latency_floor.hip — the primitives a latency-structured launch chain is made of, measured on this part (round 4):
//   (a) back-to-back EMPTY launches on one stream (cost of a launch boundary),
//   (b) one workgroup-wide chain of k DEPENDENT global loads (pointer chase; every launch starts elsewhere, so the lines are
//       cold in L1) out of a 1 MiB footprint (L2 / MALL resident) / a 1 GiB one (HBM): cost of a round trip,
//   (c) the same chain in 768 co-resident workgroups (a round trip under load).
// Build + run: hipcc --offload-arch=gfx950 -O3 tools/latency_floor.hip -o /tmp/latency_floor && /tmp/latency_floor
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
__global__ void k_empty(int* p) { if (p && threadIdx.x == 9999) *p = 1; }
__global__ void __launch_bounds__(256) k_chase(const int* __restrict__ next, int hops, int stride_wg, int* out, int base, int n) {
  int i = (int)(((long)base + (long)blockIdx.x * stride_wg + threadIdx.x) % n);
  for (int h = 0; h < hops; ++h) i = next[i];
  if (i == -1) *out = i;
}
static float run(void (*launch)(hipStream_t), hipStream_t st, int n) {
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  for (int i = 0; i < 20; ++i) launch(st);
  hipStreamSynchronize(st);
  hipEventRecord(a, st);
  for (int i = 0; i < n; ++i) launch(st);
  hipEventRecord(b, st); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  return ms * 1e3f / n;
}
static int* g_next; static int g_hops, g_wgs, g_stride, g_n, g_base; static int* g_out;
int main() {
  hipStream_t st; hipStreamCreate(&st);
  hipMalloc(&g_out, 4);
  printf("empty kernel, 1 workgroup of 64   : %.2f us per launch (back to back on one stream)\n",
         run([](hipStream_t s) { hipLaunchKernelGGL(k_empty, dim3(1), dim3(64), 0, s, (int*)nullptr); }, st, 2000));
  printf("empty kernel, 768 workgroups of 256: %.2f us per launch\n",
         run([](hipStream_t s) { hipLaunchKernelGGL(k_empty, dim3(768), dim3(256), 0, s, (int*)nullptr); }, st, 2000));
  // pointer chase: a random permutation cycle over n ints; n * 4 bytes = the footprint
  for (long n : {262144L, 268435456L}) {            // 1 MiB (L2 resident after the warm-up), 1 GiB (HBM: beyond the 256 MB MALL)
    std::vector<int> h(n);
    // a stride walk with a large odd stride: every hop lands on a different cache line / channel
    const long step = 1048573 % n ? 1048573 : 1048571;
    for (long i = 0; i < n; ++i) h[i] = (int)((i + step * 33) % n);
    hipMalloc(&g_next, n * 4); hipMemcpy(g_next, h.data(), n * 4, hipMemcpyHostToDevice);
    for (int wgs : {1, 768}) {
      float t[2];
      int hop[2] = {4, 36};
      for (int k = 0; k < 2; ++k) {
        g_hops = hop[k]; g_wgs = wgs; g_stride = (int)(n / 1024);
        g_n = (int)n; g_base = 0;
        // every launch starts its chains somewhere else: lines cold in the CU's L1 (and, for the large footprint, in L2)
        t[k] = run([](hipStream_t s) { g_base = (int)(((long)g_base + 7340033) % g_n);
                                       hipLaunchKernelGGL(k_chase, dim3(g_wgs), dim3(256), 0, s, g_next, g_hops, g_stride, g_out, g_base, g_n); }, st, 300);
      }
      printf("dependent global loads, footprint %4ld MiB, %3d workgroup(s): %.2f us per round trip (kernel %.1f us at 4 hops, %.1f us at 36)\n",
             n * 4 >> 20, wgs, (t[1] - t[0]) / 32.f, t[0], t[1]);
    }
    (void)hipFree(g_next);
  }
  return 0;
}
