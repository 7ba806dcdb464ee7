// nms.hip — torchvision.ops.nms(boxes, scores, iou_threshold) as the reference's post-processing calls it
// (cad_recognition/train.py:105 inside non_max_suppression :34-121, detect.py:118): greedy suppression in
// descending score order — a box is dropped when its IoU with an already kept, higher-scored box is > threshold.
// torchvision is a third-party dependency (unpinned next to pytorch 1.7.1, deepgcn_env_install.sh:21) and absent
// here; this restates its published kernels (csrc/cpu/nms_kernel.cpp / cuda/nms_kernel.cu):
//   area = (x2-x1)*(y2-y1);  inter = max(0, min(x2)-max(x1)) * max(0, min(y2)-max(y1));
//   suppressed  <=>  inter / (area_a + area_b - inter) > threshold          (fp32, every product rounded on its own)
// Score ties: torchvision sorts with an unstable sort; here ties keep ascending index order (stable radix sort).
//
// Three device steps, no host round trip (torchvision's CUDA path copies the n x n/64 mask to the host):
//   1. rocPRIM stable radix sort of (score, index) pairs, descending;
//   2. k_nms_mask: 64 x 64 tiles of the upper triangle, one 64-bit word per (sorted box, column tile);
//   3. k_nms_reduce: ONE workgroup walks the sorted boxes 64 at a time — the intra-chunk dependency is resolved by
//      one wave on the diagonal 64 x 64 bit block, then 16 waves OR the kept rows' words into the removed-set
//      (LDS) of the later chunks.  n <= 524 288 (the reference caps at max_nms = 30 000, train.py:47).
#include "common.hpp"

#include <string.h>
#include <rocprim/device/device_radix_sort.hpp>

typedef unsigned long long u64;

static __global__ void k_nms_iota(int* idx, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) idx[i] = i;
}

__device__ __forceinline__ bool nms_over(const float4& a, const float4& b, float thr) {
  const float left = fmaxf(a.x, b.x), right = fminf(a.z, b.z);
  const float top = fmaxf(a.y, b.y), bottom = fminf(a.w, b.w);
  const float w = fmaxf(right - left, 0.f), h = fmaxf(bottom - top, 0.f);
  const float inter = yl_mul_rn(w, h);
  const float sa = yl_mul_rn(a.z - a.x, a.w - a.y), sb = yl_mul_rn(b.z - b.x, b.w - b.y);
  return inter / (sa + sb - inter) > thr;
}

// mask[i * cb + j] bit k = sorted box i suppresses sorted box 64 j + k  (only k beyond i, only tiles j >= i / 64)
static __global__ void __launch_bounds__(64) k_nms_mask(const float4* __restrict__ boxes, const int* __restrict__ order,
                                                       int n, float thr, int cb, u64* __restrict__ mask) {
  const int rt = blockIdx.y, ct = blockIdx.x;
  if (rt > ct) return;
  __shared__ float4 cbox[64];
  const int t = threadIdx.x;
  const int cj = ct * 64 + t;
  cbox[t] = boxes[order[cj < n ? cj : n - 1]];
  __syncthreads();
  const int ri = rt * 64 + t;
  if (ri >= n) return;
  const float4 a = boxes[order[ri]];
  const int csize = (n - ct * 64) < 64 ? (n - ct * 64) : 64;
  u64 bits = 0ull;
  for (int k = (rt == ct) ? t + 1 : 0; k < csize; ++k)
    if (nms_over(a, cbox[k], thr)) bits |= 1ull << k;
  mask[(size_t)ri * cb + ct] = bits;
}

// One workgroup of 16 waves.  Per chunk c of 64 sorted boxes: wave 0 resolves the chunk on its diagonal bit block
// (loaded one chunk ahead), then the kept rows' words of the LATER chunks are OR-ed into the removed-set: wave v
// takes the kept rows v, v+16, ... (at most 4), its 64 lanes cover 64 consecutive words (512 contiguous bytes of
// a mask row), all of a block's loads are issued before the first is used, one LDS atomic OR per (lane, block).
// (A first version walked each word's kept rows in a dependent load loop: 12-20 us per chunk, 9.7 ms at n = 30 000.)
static __global__ void __launch_bounds__(1024) k_nms_reduce(const u64* __restrict__ mask, const int* __restrict__ order,
                                                           int n, int cb, long long* __restrict__ keep,
                                                           int* __restrict__ num_keep) {
  extern __shared__ u64 remv[];          // [cb] removed-set, one bit per sorted box
  __shared__ u64 kept_s;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int w = tid; w < cb; w += 1024) remv[w] = 0ull;
  u64 diag = 0ull;                        // wave 0: lane l holds the diagonal word of sorted box 64 c + l
  if (wave == 0 && lane < n) diag = mask[(size_t)lane * cb];
  __syncthreads();
  int out = 0;                            // number of boxes kept so far (same value in every thread)
  for (int c = 0; c < cb; ++c) {
    if (wave == 0) {
      u64 next = 0ull;                    // next chunk's diagonal: independent of the removed-set, fetched early
      if (c + 1 < cb && (c + 1) * 64 + lane < n) next = mask[(size_t)((c + 1) * 64 + lane) * cb + c + 1];
      u64 r = remv[c];
      if (n - c * 64 < 64) r |= ~0ull << (n - c * 64);          // boxes beyond n count as removed
      u64 kept = 0ull;
#pragma unroll 8
      for (int k = 0; k < 64; ++k) {
        const u64 dk = __shfl(diag, k);
        if (!((r >> k) & 1ull)) { kept |= 1ull << k; r |= dk; }
      }
      if ((kept >> lane) & 1ull) keep[out + __popcll(kept & ((1ull << lane) - 1ull))] = order[c * 64 + lane];
      if (lane == 0) kept_s = kept;
      diag = next;
    }
    __syncthreads();
    const u64 kept = kept_s;
    out += __popcll(kept);
    // this wave's kept rows: the (wave)-th, (wave+16)-th, ... set bits of `kept`
    int rows[4], nr = 0;
    {
      u64 kk = kept;
      int idx = 0;
      while (kk) {
        const int k = __ffsll((long long)kk) - 1;
        kk &= kk - 1;
        if ((idx & 15) == wave && nr < 4) rows[nr++] = c * 64 + k;
        ++idx;
      }
    }
    if (nr > 0) {
      for (int w0 = c + 1; w0 < cb; w0 += 64) {
        const int w = w0 + lane;
        const int wc = w < cb ? w : cb - 1;
        u64 v[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = mask[(size_t)rows[j < nr ? j : 0] * cb + wc];
        u64 acc = 0ull;
#pragma unroll
        for (int j = 0; j < 4; ++j) acc |= (j < nr) ? v[j] : 0ull;
        if (w < cb && acc) atomicOr(&remv[w], acc);
      }
    }
    __syncthreads();
  }
  if (tid == 0) *num_keep = out;
}

namespace {
struct NmsPlan { size_t off_idx, off_order, off_skeys, off_mask, off_tmp, tmp_bytes, total; int cb; };
NmsPlan nms_plan(int64_t n) {
  NmsPlan p;
  p.cb = (int)((n + 63) / 64);
  size_t tmp = 0;
  (void)rocprim::radix_sort_pairs_desc(nullptr, tmp, (const float*)nullptr, (float*)nullptr, (const int*)nullptr,
                                       (int*)nullptr, (size_t)n, 0, 32, (hipStream_t)0);
  auto al = [](size_t v) { return (v + 255) & ~(size_t)255; };
  size_t off = 0;
  p.off_idx = off; off = al(off + sizeof(int) * n);
  p.off_order = off; off = al(off + sizeof(int) * n);
  p.off_skeys = off; off = al(off + sizeof(float) * n);
  p.off_mask = off; off = al(off + sizeof(u64) * (size_t)n * p.cb);
  p.off_tmp = off; off = al(off + tmp);
  p.tmp_bytes = tmp; p.total = off + 256;
  return p;
}
}  // namespace

extern "C" size_t yolat_nms_work_bytes(int64_t n) {
  if (n <= 0 || n > 524288) return 0;
  return nms_plan(n).total;
}

extern "C" int yolat_nms(const float* boxes, const float* scores, int64_t n, float iou_threshold, int64_t* keep,
                         int32_t* num_keep, void* work, size_t work_bytes, yolat_stream_t stream) {
  if (n < 0 || !num_keep) return YOLAT_E_INVALID;
  hipStream_t st = (hipStream_t)stream;
  if (n == 0) return hipMemsetAsync(num_keep, 0, sizeof(int32_t), st) == hipSuccess ? 0 : YOLAT_E_INVALID;
  if (!boxes || !scores || !keep || !work) return YOLAT_E_INVALID;
  if (n > 524288) return YOLAT_E_UNSUPPORTED;
  if (!yl_aligned16(boxes) || (((uintptr_t)work) & 255) != 0) return YOLAT_E_UNSUPPORTED;
  const NmsPlan p = nms_plan(n);
  if (work_bytes < p.total) return YOLAT_E_INVALID;
  char* base = reinterpret_cast<char*>(work);
  int* idx = reinterpret_cast<int*>(base + p.off_idx);
  int* order = reinterpret_cast<int*>(base + p.off_order);
  float* skeys = reinterpret_cast<float*>(base + p.off_skeys);
  u64* mask = reinterpret_cast<u64*>(base + p.off_mask);
  hipLaunchKernelGGL(k_nms_iota, dim3(yl_cdiv(n, 256)), dim3(256), 0, st, idx, (int)n);
  YL_LAUNCH_CHECK();
  size_t tmp = p.tmp_bytes;
  if (rocprim::radix_sort_pairs_desc(base + p.off_tmp, tmp, scores, skeys, idx, order, (size_t)n, 0, 32, st) != hipSuccess)
    return YOLAT_E_INVALID;
  hipLaunchKernelGGL(k_nms_mask, dim3(p.cb, p.cb), dim3(64), 0, st, reinterpret_cast<const float4*>(boxes), order, (int)n,
                     iou_threshold, p.cb, mask);
  YL_LAUNCH_CHECK();
  hipLaunchKernelGGL(k_nms_reduce, dim3(1), dim3(1024), sizeof(u64) * (size_t)p.cb, st, mask, order, (int)n, p.cb,
                     reinterpret_cast<long long*>(keep), num_keep);
  YL_LAUNCH_CHECK();
  return 0;
}
