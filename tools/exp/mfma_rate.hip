// Issue rate of v_mfma_f32_32x32x16_bf16 on one SIMD as a function of the number of independent accumulator chains
// per wave (1, 2, 4) and of waves per SIMD (1, 2), all 256 CUs busy.  Prints ns and cycles (at the measured wall
// clock s_memtime rate) per MFMA per SIMD.
// Build: hipcc --offload-arch=gfx950 -O3 tools/exp/mfma_rate.hip -o tools/exp/mfma_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int CH>
__global__ void __launch_bounds__(512) k(float* out, long long* cyc, int n, int rnd) {
  f32x16 acc[CH];
  for (int c = 0; c < CH; ++c)
    for (int i = 0; i < 16; ++i) acc[c][i] = 0.f;
  bf16x8 a, b;
  if (rnd) {
    // pseudo-random operands (every lane and element different, full-entropy significands): data-dependent power
    unsigned h = threadIdx.x * 2654435761u + blockIdx.x * 40503u + 12345u;
    for (int i = 0; i < 8; ++i) {
      h = h * 1664525u + 1013904223u; a[i] = (__bf16)(((int)(h >> 8) & 0xffff) * (1.f / 32768.f) - 1.f);
      h = h * 1664525u + 1013904223u; b[i] = (__bf16)(((int)(h >> 8) & 0xffff) * (1.f / 32768.f) - 1.f);
    }
  } else {
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(threadIdx.x * 1e-3f); b[i] = (__bf16)1.0f; }
  }
  const long long t0 = clock64();
  for (int i = 0; i < n; ++i) {
#pragma unroll
    for (int c = 0; c < CH; ++c) {
      acc[c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[c], 0, 0, 0);
      if (CH == 2) asm volatile("" : "+v"(acc[0]), "+v"(acc[CH - 1]));
    }
  }
  const long long t1 = clock64();
  float r = 0.f;
  for (int c = 0; c < CH; ++c)
    for (int i = 0; i < 16; ++i) r += acc[c][i];
  if (r == 12345.678f) out[threadIdx.x] = r;
  if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}

template <int CH>
void run(float* out, long long* cyc, int threads, int total, int rnd) {
  const int n = total / CH;
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  hipLaunchKernelGGL(k<CH>, dim3(256), dim3(threads), 0, 0, out, cyc, n, rnd);
  hipDeviceSynchronize();
  hipEventRecord(a);
  for (int i = 0; i < 5; ++i) hipLaunchKernelGGL(k<CH>, dim3(256), dim3(threads), 0, 0, out, cyc, n, rnd);
  hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
  const int wps = threads / 256;
  const double ns = ms * 1e6 / 5 / ((double)total * wps);
  printf("%s operands, chains %d, waves/SIMD %d: %.2f ns per MFMA per SIMD, %.1f clock64 ticks per MFMA per SIMD (kernel %.1f us)\n", rnd ? "random" : "constant", CH, wps, ns,
         (double)c / ((double)total * wps), ms * 1e3 / 5);
}

int main() {
  float* out; long long* cyc;
  hipMalloc(&out, 4096); hipMalloc(&cyc, 8);
  const int total = 32768;
  for (int rnd = 0; rnd < 2; ++rnd) {
    run<1>(out, cyc, 256, total, rnd); run<2>(out, cyc, 256, total, rnd); run<4>(out, cyc, 256, total, rnd);
    run<1>(out, cyc, 512, total, rnd); run<2>(out, cyc, 512, total, rnd); run<4>(out, cyc, 512, total, rnd);
  }
  return 0;
}
