// Do fp32-input MFMAs and plain VALU instructions of two different waves on the same SIMD overlap?
// 512-thread workgroups (2 waves per SIMD), one per CU: waves 0-3 issue `nm` dependent v_mfma_f32_32x32x2_f32
// (or bf16 32x32x16 with -DBF16), waves 4-7 issue `nv` v_fma_f32 in 8 independent chains.  Timed: MFMA only,
// VALU only, both.  Build: hipcc --offload-arch=gfx950 -O3 tools/exp/pipe_overlap.hip -o tools/exp/pipe_overlap
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <bool BF>
__global__ void __launch_bounds__(512) k(float* out, int nm, int nv, int mode) {
  const int wave = threadIdx.x >> 6;
  float r = 0.f;
  if (wave < 4) {
    if (mode & 1) {
      f32x16 acc;
      for (int i = 0; i < 16; ++i) acc[i] = 0.f;
      float a = threadIdx.x * 1e-3f, b = 1.0001f;
      bf16x8 ab, bb;
      for (int i = 0; i < 8; ++i) { ab[i] = (__bf16)a; bb[i] = (__bf16)b; }
      for (int i = 0; i < nm; ++i) {
        if (BF) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ab, bb, acc, 0, 0, 0);
        else acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
      }
      for (int i = 0; i < 16; ++i) r += acc[i];
    }
  } else {
    if (mode & 2) {
      float c[8];
      for (int j = 0; j < 8; ++j) c[j] = threadIdx.x + j;
      const float m = 1.0001f, d = 0.5f;
      for (int i = 0; i < nv; i += 8) {
#pragma unroll
        for (int j = 0; j < 8; ++j) c[j] = __builtin_fmaf(c[j], m, d);
      }
      for (int j = 0; j < 8; ++j) r += c[j];
    }
  }
  if (r == 12345.678f) out[threadIdx.x] = r;
}

template <bool BF>
float run(float* out, int nm, int nv, int mode) {
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  hipLaunchKernelGGL(k<BF>, dim3(256), dim3(512), 0, 0, out, nm, nv, mode);
  hipDeviceSynchronize();
  hipEventRecord(a);
  for (int i = 0; i < 5; ++i) hipLaunchKernelGGL(k<BF>, dim3(256), dim3(512), 0, 0, out, nm, nv, mode);
  hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  return ms / 5 * 1e3f;
}

int main() {
  float* out; hipMalloc(&out, 4096);
  const int nm = 4096;
  for (int nv : {0, 32768, 65536, 131072}) {
    printf("fp32 MFMA x%d | v_fma x%d :  mfma only %.1f us, valu only %.1f us, both %.1f us\n", nm, nv,
           run<false>(out, nm, nv, 1), run<false>(out, nm, nv, 2), run<false>(out, nm, nv, 3));
  }
  for (int nv : {0, 32768, 65536}) {
    printf("bf16 MFMA x%d | v_fma x%d :  mfma only %.1f us, valu only %.1f us, both %.1f us\n", 2 * nm, nv,
           run<true>(out, 2 * nm, nv, 1), run<true>(out, 2 * nm, nv, 2), run<true>(out, 2 * nm, nv, 3));
  }
  return 0;
}
