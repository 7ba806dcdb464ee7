// VALU issue rate on gfx950 by waves per SIMD: v_fma_f32 (8 independent chains per wave) and v_pk_fma_f32.
// One workgroup of W waves per CU (W = 4, 8, 16 -> 1, 2, 4 waves per SIMD), nv instructions per wave.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x2 __attribute__((ext_vector_type(2)));
template <int PK>
__global__ void k(float* out, int nv) {
  float r = 0.f;
  if (PK) {
    f32x2 c[8];
    for (int j = 0; j < 8; ++j) { c[j].x = threadIdx.x + j; c[j].y = j; }
    const f32x2 m = {1.0001f, 0.999f}, d = {0.5f, 0.25f};
    for (int i = 0; i < nv; i += 8) {
#pragma unroll
      for (int j = 0; j < 8; ++j) c[j] = __builtin_elementwise_fma(c[j], m, d);
    }
    for (int j = 0; j < 8; ++j) r += c[j].x + c[j].y;
  } else {
    float c[8];
    for (int j = 0; j < 8; ++j) c[j] = threadIdx.x + j;
    const float m = 1.0001f, d = 0.5f;
    for (int i = 0; i < nv; i += 8) {
#pragma unroll
      for (int j = 0; j < 8; ++j) c[j] = __builtin_fmaf(c[j], m, d);
    }
    for (int j = 0; j < 8; ++j) r += c[j];
  }
  if (r == 12345.678f) out[threadIdx.x] = r;
}
template <int PK>
float run(float* out, int threads, int nv) {
  hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
  hipLaunchKernelGGL(k<PK>, dim3(256), dim3(threads), 0, 0, out, nv);
  (void)hipDeviceSynchronize();
  (void)hipEventRecord(a);
  for (int i = 0; i < 5; ++i) hipLaunchKernelGGL(k<PK>, dim3(256), dim3(threads), 0, 0, out, nv);
  (void)hipEventRecord(b); (void)hipEventSynchronize(b);
  float ms; (void)hipEventElapsedTime(&ms, a, b);
  return ms / 5 * 1e3f;
}
int main() {
  float* out; (void)hipMalloc(&out, 8192);
  const int nv = 65536;
  for (int threads : {256, 512, 1024}) {
    const float t0 = run<0>(out, threads, nv), t1 = run<1>(out, threads, nv);
    const double wps = threads / 256.0;
    printf("%d waves/SIMD: v_fma_f32 %.1f us = %.2f ns per instr per SIMD;  v_pk_fma_f32 %.1f us = %.2f ns per instr per SIMD\n",
           (int)wps, t0, t0 * 1e3 / (nv * wps), t1, t1 * 1e3 / (nv * wps));
  }
  return 0;
}
