// Dev experiment: where does the hardware dispatcher put the workgroups of a 632-WG launch whose
// kernel could fit 4 WGs per CU?  Records (xcc_id, hw_id) per workgroup while all spin ~30 us.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <map>
#include <vector>
__global__ void __launch_bounds__(256) probe(unsigned* out, long spin, int lds_words) {
  extern __shared__ float sm[];
  if (threadIdx.x < lds_words) sm[threadIdx.x] = 1.f;
  __syncthreads();
  unsigned hw, xcc;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
  long t0 = clock64();
  while (clock64() - t0 < spin) {}
  if (threadIdx.x == 0) { out[2 * blockIdx.x] = hw; out[2 * blockIdx.x + 1] = xcc | (unsigned)(sm[0] > 2.f); }
}
int main(int argc, char** argv) {
  int nwg = argc > 1 ? atoi(argv[1]) : 632;
  int lds = argc > 2 ? atoi(argv[2]) : 17408;
  unsigned* d; hipMalloc(&d, nwg * 8);
  std::vector<unsigned> h(nwg * 2);
  for (int rep = 0; rep < 2; ++rep) {
    hipLaunchKernelGGL(probe, dim3(nwg), dim3(256), lds, 0, d, 60000L, 64);
    hipDeviceSynchronize();
  }
  hipMemcpy(h.data(), d, nwg * 8, hipMemcpyDeviceToHost);
  std::map<unsigned long, int> per_cu;
  for (int i = 0; i < nwg; ++i) {
    unsigned hw = h[2 * i], xcc = h[2 * i + 1] & 0xf;
    unsigned cu = (hw >> 8) & 0xf, sh = (hw >> 12) & 1, se = (hw >> 13) & 0x7;
    per_cu[((unsigned long)xcc << 16) | (se << 8) | (sh << 4) | cu]++;
  }
  std::map<int, int> hist;
  for (auto& kv : per_cu) hist[kv.second]++;
  printf("nwg=%d lds=%d distinct CUs used=%zu; histogram (WGs per CU -> #CUs):", nwg, lds, per_cu.size());
  for (auto& kv : hist) printf(" %d->%d", kv.first, kv.second);
  printf("\n");
  return 0;
}
