// Dev experiment: aggregate L2 -> register load bandwidth for the staging patterns of the GEMM kernels.
// Every workgroup streams `bytes_per_wg` out of a buffer of `footprint` bytes (L2 / MALL resident), 4 x 16-B
// loads per thread in flight, and folds the data into a checksum.
#include <hip/hip_runtime.h>
#include <stdio.h>
template <int SEG>   // lanes per contiguous segment: 8 -> 128-B row pieces (rows `stride` apart), 64 -> 1 KiB contiguous
__global__ void __launch_bounds__(256) k_stream(const float4* __restrict__ buf, long footprint_f4, long stride_f4, int iters,
                                                float* out) {
  const int tid = threadIdx.x;
  const long wg_base = ((long)blockIdx.x * 7919) % (footprint_f4 / 2);
  float4 acc = make_float4(0, 0, 0, 0);
  for (int it = 0; it < iters; ++it) {
    float4 v[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const int i = tid + t * 256;                         // 1024 float4 slots per iteration = 16 KB
      long off;
      if (SEG == 64) off = wg_base + (long)it * 1024 + i;
      else off = wg_base + (long)(i / SEG) * stride_f4 + (i % SEG) + (long)it * SEG;
      v[t] = buf[off % footprint_f4];
    }
#pragma unroll
    for (int t = 0; t < 4; ++t) { acc.x += v[t].x; acc.y += v[t].y; acc.z += v[t].z; acc.w += v[t].w; }
  }
  if (acc.x + acc.y + acc.z + acc.w == 12345.678f) out[0] = acc.x;
}
template <int SEG>
static void run(const char* tag, long footprint, long stride_bytes, int wgs, int iters, float4* buf, float* out) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int i = 0; i < 2; ++i) hipLaunchKernelGGL(k_stream<SEG>, dim3(wgs), dim3(256), 0, 0, buf, footprint / 16, stride_bytes / 16, iters, out);
  hipEventRecord(e0);
  for (int i = 0; i < 5; ++i) hipLaunchKernelGGL(k_stream<SEG>, dim3(wgs), dim3(256), 0, 0, buf, footprint / 16, stride_bytes / 16, iters, out);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double bytes = 5.0 * wgs * (double)iters * 16384;
  printf("%-34s footprint %6.1f MB  wgs %5d  %7.2f TB/s\n", tag, footprint / 1048576.0, wgs, bytes / (ms * 1e-3) / 1e12);
}
int main() {
  float4* buf; float* out;
  hipMalloc(&buf, (size_t)1 << 30); hipMalloc(&out, 64);
  hipMemset(buf, 0, (size_t)1 << 30);
  for (long fp : {2L << 20, 16L << 20, 128L << 20, 1024L << 20}) {
    for (int wgs : {256, 1024, 4096}) {
      run<64>("1 KiB contiguous per wave", fp, 0, wgs, 256, buf, out);
      run<8>("128-B pieces, rows 512 B apart", fp, 512, wgs, 256, buf, out);
      run<8>("128-B pieces, rows 9216 B apart", fp, 9216, wgs, 256, buf, out);
    }
  }
  return 0;
}
