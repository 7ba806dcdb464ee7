// cls_tail.hip — the last two layers of the classifier in ONE launch, for batches of a few thousand proposals (round 6).
//
// Reference: prediction_cls.1 (Linear 512 -> 256 + BatchNorm1d(eval) + ReLU) and prediction_cls.2 (Linear 256 -> n_classes),
// cad_recognition/architecture3cc_rpn_gp_iter2.py:91-93,127-128 over gcn_lib/sparse/torch_nn.py:50-71, eval mode.
//
// Why: at P = 400 (the headline configuration) the two layers are two latency-structured launches — 9.3 + 5.8 us for 0.11
// GFLOP — and the second one exists only because a 32 x 32 output tile of the first does not hold whole rows.  Here a
// workgroup owns 16 WHOLE rows: it stages them (16 x H1 fp32) in LDS, computes all H2 columns of layer 2 on the fp32 matrix
// cores (v_mfma_f32_16x16x4_f32: 8 waves x H2 / 128 column tiles x H1 / 4 steps, the weight rows streamed from L2 as 16-byte
// pieces — the k order inside a 16-block is permuted identically for both operands so that a lane's four steps are one
// 16-byte load), applies bias / folded BatchNorm / ReLU, keeps the 16 x H2 activations in LDS and finishes with layer 3's
// n_classes dot products per row on the vector ALU.  fp32 products and sums throughout (1e-6 of the two-launch result:
// different summation order only).  One round of workgroups up to P = 4096; the caller keeps the two launches beyond.
#include "common.hpp"

namespace {
constexpr int CT_ROWS = 16;

template <int NT2>   // column tiles of 16 per wave: H2 = 128 * NT2
__global__ void __launch_bounds__(512) k_cls_tail(const float* __restrict__ X, long ldx, int P, int H1,
                                                  const float* __restrict__ W2, const float* __restrict__ b2,
                                                  const float* __restrict__ s2, const float* __restrict__ t2,
                                                  const float* __restrict__ W3, const float* __restrict__ b3, int K3,
                                                  float* __restrict__ logits, long ldl) {
  constexpr int H2 = 128 * NT2;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int LDX = H1 + 4;                       // (+ 4 floats: rows 16 bytes apart in the bank pattern)
  float* Xs = smem;                             // [16][H1 + 4]
  float* Ys = smem + CT_ROWS * LDX;             // [16][H2 + 4]
  constexpr int LDY = H2 + 4;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, fr = lane & 15, fk = lane >> 4;
  const int r0 = blockIdx.x * CT_ROWS;
  // ---- stage the 16 rows
  for (int i = tid; i < CT_ROWS * (H1 >> 2); i += 512) {
    const int r = i / (H1 >> 2), c4 = i - r * (H1 >> 2);
    const int row = yl_min(r0 + r, P - 1);
    *reinterpret_cast<float4*>(Xs + r * LDX + 4 * c4) = *reinterpret_cast<const float4*>(X + (long)row * ldx + 4 * c4);
  }
  __syncthreads();
  // ---- layer 2: wave w owns column tiles w + 8 j
  f32x4 acc[NT2];
#pragma unroll
  for (int j = 0; j < NT2; ++j) acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
  const float* xrow = Xs + fr * LDX + 4 * fk;
  const float* wrow[NT2];
#pragma unroll
  for (int j = 0; j < NT2; ++j) wrow[j] = W2 + (long)((wave + 8 * j) * 16 + fr) * H1 + 4 * fk;
  const int nblk = H1 >> 4;
  float4 bq[NT2];
#pragma unroll
  for (int j = 0; j < NT2; ++j) bq[j] = *reinterpret_cast<const float4*>(wrow[j]);
  for (int i = 0; i < nblk; ++i) {
    float4 bn[NT2];
    const int inx = i + 1 < nblk ? i + 1 : i;
#pragma unroll
    for (int j = 0; j < NT2; ++j) bn[j] = *reinterpret_cast<const float4*>(wrow[j] + 16 * inx);     // next block's pieces
    const float4 a = *reinterpret_cast<const float4*>(xrow + 16 * i);
#pragma unroll
    for (int j = 0; j < NT2; ++j) {
      acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, bq[j].x, acc[j], 0, 0, 0);
      acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, bq[j].y, acc[j], 0, 0, 0);
      acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.z, bq[j].z, acc[j], 0, 0, 0);
      acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.w, bq[j].w, acc[j], 0, 0, 0);
    }
#pragma unroll
    for (int j = 0; j < NT2; ++j) bq[j] = bn[j];
  }
  // epilogue: relu(s2 * (acc + b2) + t2) -> Ys   (C layout: rows 4 fk + r, column fr of the tile)
#pragma unroll
  for (int j = 0; j < NT2; ++j) {
    const int col = (wave + 8 * j) * 16 + fr;
    const float bias = b2 ? b2[col] : 0.f, sc = s2 ? s2[col] : 1.f, sh = s2 ? t2[col] : 0.f;
#pragma unroll
    for (int r = 0; r < 4; ++r) Ys[(4 * fk + r) * LDY + col] = fmaxf(fmaf(acc[j][r] + bias, sc, sh), 0.f);
  }
  __syncthreads();
  // ---- layer 3: logits[row][c] = b3[c] + sum_k Ys[row][k] W3[c][k]
  for (int i = tid; i < CT_ROWS * K3; i += 512) {
    const int r = i / K3, c = i - r * K3;
    if (r0 + r >= P) continue;
    const float* y = Ys + r * LDY;
    const float* w = W3 + (long)c * H2;
    float s0 = 0.f, s1 = 0.f, s2_ = 0.f, s3 = 0.f;
#pragma unroll 4
    for (int k = 0; k < H2; k += 4) {
      const float4 yv = *reinterpret_cast<const float4*>(y + k);
      const float4 wv = *reinterpret_cast<const float4*>(w + k);
      s0 = fmaf(yv.x, wv.x, s0); s1 = fmaf(yv.y, wv.y, s1); s2_ = fmaf(yv.z, wv.z, s2_); s3 = fmaf(yv.w, wv.w, s3);
    }
    logits[(long)(r0 + r) * ldl + c] = ((s0 + s1) + (s2_ + s3)) + (b3 ? b3[c] : 0.f);
  }
}
}  // namespace

extern "C" int yolat_cls_tail_eval(const float* X, int64_t ldx, int64_t P, int64_t H1, const float* W2, const float* b2,
                                   const float* s2, const float* t2, int64_t H2, const float* W3, const float* b3, int64_t K3,
                                   float* logits, int64_t ld_logits, yolat_stream_t stream) {
  if (!X || !W2 || !W3 || !logits || P <= 0 || H1 <= 0 || H2 <= 0 || K3 <= 0 || ldx < H1 || ld_logits < K3) return YOLAT_E_INVALID;
  if ((s2 == nullptr) != (t2 == nullptr)) return YOLAT_E_INVALID;
  if (H1 % 16 != 0 || (H2 != 128 && H2 != 256) || K3 > 64 || ldx % 4 != 0 || !yl_aligned16(X) || !yl_aligned16(W2) ||
      !yl_aligned16(W3) || P > (1LL << 24))
    return YOLAT_E_UNSUPPORTED;
  const size_t lds = (size_t)(CT_ROWS * (H1 + 4) + CT_ROWS * (H2 + 4)) * sizeof(float);
  if (lds > 160 * 1024) return YOLAT_E_UNSUPPORTED;
  const dim3 grid((unsigned)yl_cdiv(P, CT_ROWS));
  if (H2 == 256)
    hipLaunchKernelGGL(k_cls_tail<2>, grid, dim3(512), lds, (hipStream_t)stream, X, (long)ldx, (int)P, (int)H1, W2, b2, s2, t2, W3,
                       b3, (int)K3, logits, (long)ld_logits);
  else
    hipLaunchKernelGGL(k_cls_tail<1>, grid, dim3(512), lds, (hipStream_t)stream, X, (long)ldx, (int)P, (int)H1, W2, b2, s2, t2, W3,
                       b3, (int)K3, logits, (long)ld_logits);
  YL_LAUNCH_CHECK();
  return 0;
}
