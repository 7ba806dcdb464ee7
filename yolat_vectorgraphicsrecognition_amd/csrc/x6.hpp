// x6.hpp — shared pieces of the bf16x6-emulated fp32 GEMM kernels (fusion_x6.hip, gemm_x6.hip): exact 3-way bfloat16
// split of fp32 values.  x = h + m + l with h = the top 8 significand bits of x, m the next 8, l the last 8
// (truncation splits: every term is exactly representable in bfloat16); six of the nine products a_i * b_j — all but
// the three O(2^-24) ones — reproduce the fp32 product to ~3e-7 relative with fp32 accumulation.
//
// Non-finite and tiny inputs.  The split is exact for every finite x whose three terms are normal or zero.  It is NOT a
// drop-in for an IEEE fp32 product at the edges of the format:
//   * x = +-Inf: h = +-Inf, m = hm - h = Inf - Inf = NaN — an Inf operand turns its output row / column into NaN where a
//     true fp32 GEMM would give +-Inf (or NaN only against a zero); NaN stays NaN.  The model's activations are finite
//     (BatchNorm'd features, bounded Bezier attributes); yolat_check_last_status-style input checks do not look at values.
//   * |x| < 2^-110 or so: the low terms underflow bfloat16's normal range only when x itself is within 2^16 of the
//     smallest normal fp32 (bf16 shares fp32's exponent range), i.e. products that are far below fp32's own rounding of
//     any sum they enter.
// Strict-parity runs select the fp32-input MFMA kernels everywhere with ONE switch, YOLAT_STRICT_FP32=1 (common.hpp
// yl_strict_fp32; plan.py / ops.py read the same variable): IEEE propagation and the fp32 MFMA summation order, at the
// fp32 vector-ALU rate (DESIGN.md "fp32 MFMA and the vector ALU").
#pragma once
#include "common.hpp"

typedef __bf16 fx_bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned fx_u32x4 __attribute__((ext_vector_type(4)));

namespace {
// exact 3-way bfloat16 split of 8 fp32 values (h = top 8 significand bits, hm = top 16: m = hm - h and l = x - hm are
// exact and need 8 bits each)
__device__ __forceinline__ void fx_split8(const float x[8], fx_bf16x8& h, fx_bf16x8& m, fx_bf16x8& l) {
  fx_u32x4 ph, pm, pl;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const unsigned x0 = __float_as_uint(x[2 * i]), x1 = __float_as_uint(x[2 * i + 1]);
    const yl_f32x2 xv = {x[2 * i], x[2 * i + 1]};
    const yl_f32x2 hv = {__uint_as_float(x0 & 0xffff0000u), __uint_as_float(x1 & 0xffff0000u)};
    const yl_f32x2 hmv = {__uint_as_float(x0 & 0xffffff00u), __uint_as_float(x1 & 0xffffff00u)};
    const yl_f32x2 mv = hmv - hv, lv = xv - hmv;
    ph[i] = __builtin_amdgcn_perm(x1, x0, 0x07060302u);
    pm[i] = __builtin_amdgcn_perm(__float_as_uint(mv.y), __float_as_uint(mv.x), 0x07060302u);
    pl[i] = __builtin_amdgcn_perm(__float_as_uint(lv.y), __float_as_uint(lv.x), 0x07060302u);
  }
  h = *reinterpret_cast<fx_bf16x8*>(&ph);
  m = *reinterpret_cast<fx_bf16x8*>(&pm);
  l = *reinterpret_cast<fx_bf16x8*>(&pl);
}
}  // namespace
