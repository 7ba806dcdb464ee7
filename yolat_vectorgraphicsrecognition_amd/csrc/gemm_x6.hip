// gemm_x6.hip — eval-mode Linear (+ folded BatchNorm + ReLU) as an LDS-tiled fp32 GEMM EMULATED on the bf16 matrix
// cores (round 2): out [M, N] = act(A [M, K] . W'^T + shift), W' = scale (rows) * W.  Built for the per-proposal
// classifier head (architecture3cc_rpn_gp_iter2.py:91-93,127-128: P x 2304 -> 512), whose fp32-MFMA kernels are
// bound by the fp32 matrix rate (64 cycles per 32x32x2 step; DESIGN.md "fp32 MFMA and the vector ALU").
//
// Operands: x = h + m + l exactly (x6.hpp), six bf16 products per k (32 cycles per 32x32x16 step: 2.7x the fp32
// rate), fp32 accumulation: ~3e-7 relative to the fp32 product.  The weights are split ONCE per weight version and
// stored in the kernel's LDS image order (yolat_gemm_x6_pack), so staging a weight tile is a straight 12 KB copy; the
// activations are loaded as fp32 and split on their way into LDS (one fx_split8 per thread and k step).
//
// Tiling: 512 threads = 2 x 4 waves (two per SIMD: one wave's LDS / VMEM / barrier waits hide under the other's
// MFMAs), workgroup tile 128 x 128, wave tile 64 x 32 (2 accumulators: 12 MFMAs per 9 fragment reads per k step of
// 16), two k steps per LDS stage, THREE stage buffers (3 x 48 KB), one barrier per stage; the fragment reads of k
// step 1 are issued among the MFMAs of k step 0; the split + LDS write of stage s+2 and the first fragment reads of
// stage s+1 among the MFMAs of k step 1; the global loads of stage s+3 right after (hand-placed with sched_barrier:
// the loop is one basic block and no LDS read waits behind a barrier).
// LDS image of one operand part: [row][16 k] bf16 = 32 B per row, the two 16-byte halves of rows 8..15 (mod 16)
// swapped, which makes both the ds_write_b128 of the staging threads and the ds_read_b128 of the MFMA operand
// fragments conflict-free.  Few rows (P = a few hundred): the launch splits K over gridDim.y workgroups writing fp32
// partials, summed in a fixed order by k_gemm_x6_reduce (deterministic; no atomics).
#include "x6.hpp"

#define GX_BM 128
#define GX_BN 128
#define GX_PART (GX_BN * 16)          // bf16 elements of one part of one (column tile, k step) image
#define GX_KS 2                       // k steps (of 16) per LDS stage and barrier

namespace {
// element offset of (row, k half) inside a part image
__device__ __forceinline__ int gx_slot(int row, int khalf) { return row * 16 + ((khalf ^ ((row >> 3) & 1)) << 3); }

template <bool SPLITK>
__global__ void __launch_bounds__(512) k_gemm_x6(const float* __restrict__ A, long lda, int M, int K,
                                                 const yl_bf16_t* __restrict__ Wp, const float* __restrict__ shift,
                                                 int relu, int N, float* __restrict__ out, long ldo, int tn,
                                                 int st_per_split, float* __restrict__ stats) {
  // stats != NULL (!SPLITK only): BatchNorm partial statistics of the stored pre-activation values, per 32-row group and
  // column (sum, M2) in the layout yolat_bn_finalize reads (training-mode Linear)
  // one stage = GX_KS k steps of 16; images [k step][part][row slot]
  __shared__ __attribute__((aligned(16))) yl_bf16_t As[3][GX_KS * 3 * GX_BM * 16];
  __shared__ __attribute__((aligned(16))) yl_bf16_t Bs[3][GX_KS * 3 * GX_PART];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, l31 = lane & 31, lhi = lane >> 5;
  // consecutive workgroup ids go to different XCDs: deal the tiles so that each XCD gets a contiguous range of
  // (row tile, column tile) pairs, column tile fastest — the column tiles of a row tile share its A rows in ONE L2
  int logical;
  {
    const int total = gridDim.x, chunk = total >> 3, rem = total & 7;
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    logical = xcd * chunk + (xcd < rem ? xcd : rem) + slot;
  }
  const int ct = logical % tn, rt = logical / tn;
  const int row0 = rt * GX_BM, col0 = ct * GX_BN;
  const int nks = K >> 4, nst = (nks + GX_KS - 1) / GX_KS;
  const int s0 = blockIdx.y * st_per_split, s1 = yl_min(nst, s0 + st_per_split);
  // staging roles: thread t splits the 8 k values of (k step su, row srow, k half shalf) of A per stage and copies
  // three 16-byte pieces of the stage's weight image (contiguous in Wp: k steps are adjacent)
  const int su = tid >> 8, srow = (tid & 255) >> 1, shalf = tid & 1;
  const float* ap = A + (long)yl_min(row0 + srow, M - 1) * lda + 8 * shalf;
  const int a_slot = su * (3 * GX_BM * 16) + gx_slot(srow, shalf);
  const yl_bf16_t* wp = Wp + (long)ct * nks * (3 * GX_PART);
  // 2 x 4 waves: wave tile 64 rows x 32 columns
  const int wr = wave >> 2, wc = wave & 3;
  const int a_off0 = gx_slot(64 * wr + l31, lhi), a_off1 = gx_slot(64 * wr + 32 + l31, lhi);
  const int b_off = gx_slot(32 * wc + l31, lhi);
  f32x16 acc[2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;

  float4 ra0, ra1;
  fx_u32x4 rw[3];
  // a stage / k step past the end (the last iterations' prefetch, odd K / 16): loads clamped to the last k step, A
  // values zeroed — everything in the loop is unconditional, so that it stays one basic block
  auto gload = [&](int st) {
    const int ks = yl_min(st * GX_KS + su, nks - 1);
    ra0 = *reinterpret_cast<const float4*>(ap + 16 * ks);
    ra1 = *reinterpret_cast<const float4*>(ap + 16 * ks + 4);
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const int piece = tid + 512 * c;                       // 16-byte piece of the stage image: [k step][part][2048 / 8]
      const int ks_w = yl_min(st * GX_KS + piece / 768, nks - 1);
      rw[c] = *reinterpret_cast<const fx_u32x4*>(wp + (long)ks_w * (3 * GX_PART) + (piece % 768) * 8);
    }
  };
  fx_u32x4 sh_, sm_, sl_;
  auto split_pair = [&](int p, float g) {
    const float xa[8] = {ra0.x, ra0.y, ra0.z, ra0.w, ra1.x, ra1.y, ra1.z, ra1.w};
    const float v0 = xa[2 * p] * g, v1 = xa[2 * p + 1] * g;
    const unsigned x0 = __float_as_uint(v0), x1 = __float_as_uint(v1);
    const yl_f32x2 xv = {v0, v1};
    const yl_f32x2 hv = {__uint_as_float(x0 & 0xffff0000u), __uint_as_float(x1 & 0xffff0000u)};
    const yl_f32x2 hmv = {__uint_as_float(x0 & 0xffffff00u), __uint_as_float(x1 & 0xffffff00u)};
    const yl_f32x2 mv = hmv - hv, lv = xv - hmv;
    sh_[p] = __builtin_amdgcn_perm(x1, x0, 0x07060302u);
    sm_[p] = __builtin_amdgcn_perm(__float_as_uint(mv.y), __float_as_uint(mv.x), 0x07060302u);
    sl_[p] = __builtin_amdgcn_perm(__float_as_uint(lv.y), __float_as_uint(lv.x), 0x07060302u);
  };
  auto write_a = [&](int buf) {
    *reinterpret_cast<fx_u32x4*>(&As[buf][a_slot]) = sh_;
    *reinterpret_cast<fx_u32x4*>(&As[buf][a_slot + GX_BM * 16]) = sm_;
    *reinterpret_cast<fx_u32x4*>(&As[buf][a_slot + 2 * GX_BM * 16]) = sl_;
  };
  auto write_b = [&](int buf) {
#pragma unroll
    for (int c = 0; c < 3; ++c) *reinterpret_cast<fx_u32x4*>(&Bs[buf][(tid + 512 * c) * 8]) = rw[c];
  };
  struct Frags { fx_bf16x8 a[2][3], b[3]; };
  // fragment g (0..8) of k step u in the order the MFMAs below first need them
  auto read_frag = [&](int buf, int u, Frags& f, int g) {
    constexpr int kind[9] = {0, 1, 2, 0, 1, 2, 0, 1, 2}, part[9] = {2, 2, 0, 0, 0, 2, 1, 1, 1};
    const int q = part[g];
    if (kind[g] == 0) f.a[0][q] = *reinterpret_cast<const fx_bf16x8*>(&As[buf][(u * 3 + q) * (GX_BM * 16) + a_off0]);
    else if (kind[g] == 1) f.a[1][q] = *reinterpret_cast<const fx_bf16x8*>(&As[buf][(u * 3 + q) * (GX_BM * 16) + a_off1]);
    else f.b[q] = *reinterpret_cast<const fx_bf16x8*>(&Bs[buf][(u * 3 + q) * GX_PART + b_off]);
  };
  // MFMA m (0..11): six products (small terms first), the two accumulators in turn inside each
  auto mfma_one = [&](const Frags& f, int m) {
    constexpr int qa[6] = {2, 0, 1, 1, 0, 0}, qb[6] = {0, 2, 1, 0, 1, 0};
    const int t = m >> 1, i = m & 1;
    acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.a[i][qa[t]], f.b[qb[t]], acc[i], 0, 0, 0);
  };
  static_assert(GX_KS == 2, "the loop below is written for two k steps per stage");
  auto stage_all = [&](int st, int buf) {
    const float g = (st * GX_KS + su < nks) ? 1.f : 0.f;
#pragma unroll
    for (int p = 0; p < 4; ++p) split_pair(p, g);
    write_a(buf);
    write_b(buf);
  };
  // Three LDS stage buffers: stage s+2 is written during stage s, so stage s+1 is complete (one barrier old) while
  // stage s computes and its first fragments can be read BEFORE the barrier that ends stage s — no fragment read is
  // ever exposed behind a barrier.
  if (s0 < s1) {
    gload(s0);
    stage_all(s0, 0);
    gload(s0 + 1);
    stage_all(s0 + 1, 1);
    gload(s0 + 2);
  }
  __syncthreads();
  Frags f0, f1;
  if (s0 < s1) {
#pragma unroll
    for (int g = 0; g < 9; ++g) read_frag(0, 0, f0, g);
  }
  int buf = 0;
  for (int st = s0; st < s1; ++st) {
    const int buf1 = buf == 2 ? 0 : buf + 1, buf2 = buf1 == 2 ? 0 : buf1 + 1;
    // sched_barrier(0) after every group: the order written here is the order issued.
    // k step 0: its 12 MFMAs cover the 9 fragment reads of k step 1
#pragma unroll
    for (int g = 0; g < 6; ++g) {
      read_frag(buf, 1, f1, g);
      if (g < 3) read_frag(buf, 1, f1, 6 + g);
      mfma_one(f0, 2 * g);
      mfma_one(f0, 2 * g + 1);
      __builtin_amdgcn_sched_barrier(0);
    }
    // k step 1: its MFMAs cover (a) the split + LDS write of stage st+2 (its global loads went out an iteration ago)
    // and (b) the first fragment reads of stage st+1; then the loads of stage st+3 go out
    const float gn = ((st + 2) * GX_KS + su < nks) ? 1.f : 0.f;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      mfma_one(f1, 3 * g);
      mfma_one(f1, 3 * g + 1);
      mfma_one(f1, 3 * g + 2);
      split_pair(g, gn);
      if (g == 0) write_b(buf2);
      read_frag(buf1, 0, f0, 2 * g);
      read_frag(buf1, 0, f0, 2 * g + 1);
      if (g == 3) read_frag(buf1, 0, f0, 8);
      __builtin_amdgcn_sched_barrier(0);
    }
    write_a(buf2);
    gload(st + 3);
    __syncthreads();
    buf = buf1;
  }
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int col = col0 + 32 * wc + l31;
    const float sh = (!SPLITK && shift && col < N) ? shift[col] : 0.f;
    if (!SPLITK && stats != nullptr) {
      const int rbase = row0 + 64 * wr + 32 * i;
      int cnt = M - rbase;
      cnt = cnt > 32 ? 32 : cnt;
      if (cnt > 0) {
        float sum = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) sum += (rbase + (r & 3) + 8 * (r >> 2) + 4 * lhi < M) ? acc[i][r] + sh : 0.f;
        sum += __shfl_xor(sum, 32);
        const float mu = sum / (float)cnt;
        float m2 = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float d = acc[i][r] + sh - mu;
          m2 += (rbase + (r & 3) + 8 * (r >> 2) + 4 * lhi < M) ? d * d : 0.f;
        }
        m2 += __shfl_xor(m2, 32);
        if (lhi == 0 && col < N) reinterpret_cast<float2*>(stats)[(long)(rbase >> 5) * N + col] = make_float2(sum, m2);
      }
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = row0 + 64 * wr + 32 * i + (r & 3) + 8 * (r >> 2) + 4 * lhi;
      if (row < M && col < N) {
        if (SPLITK) {
          out[((long)blockIdx.y * M + row) * N + col] = acc[i][r];
        } else {
          float v = acc[i][r] + sh;
          if (relu) v = fmaxf(v, 0.f);
          out[(long)row * ldo + col] = v;
        }
      }
    }
  }
}

// out[row, col] = act(sum_s partial[s][row][col] + shift[col]), s ascending
__global__ void k_gemm_x6_reduce(const float* __restrict__ partial, int S, long M, int N, const float* __restrict__ shift,
                                 int relu, float* __restrict__ out, long ldo) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= M * N) return;
  const long row = i / N;
  const int col = (int)(i - row * N);
  float v = partial[i];
  for (int s = 1; s < S; ++s) v += partial[(long)s * M * N + i];
  if (shift) v += shift[col];
  if (relu) v = fmaxf(v, 0.f);
  out[row * ldo + col] = v;
}

// W [N, K] fp32 (optionally scaled per row) -> packed[ct][ks][part][gx_slot(col, k half)]; columns beyond N are zero
// transposed != 0: W is given as [K, N] row-major (the GEMM's weight is its transpose: dX = dY . W of a Linear)
// kvalid: k indices at or beyond it read as zero (the transposed form of a matrix whose row count is not a multiple of 16)
__global__ void k_gemm_x6_pack(const float* __restrict__ W, long ldw, int N, int K, const float* __restrict__ row_scale,
                               int transposed, int kvalid, yl_bf16_t* __restrict__ packed) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;          // (ct, ks, col j, k half)
  const int nks = K >> 4, tn = (N + GX_BN - 1) / GX_BN;
  if (i >= (long)tn * nks * GX_BN * 2) return;
  const int half = (int)(i & 1), j = (int)((i >> 1) % GX_BN);
  const long cs = (i >> 1) / GX_BN;
  const int ks = (int)(cs % nks), ct = (int)(cs / nks);
  const int n = ct * GX_BN + j;
  float x[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const int k = 16 * ks + 8 * half + e;
    x[e] = (n < N && k < kvalid) ? (transposed ? W[(long)k * ldw + n] : W[(long)n * ldw + k]) * (row_scale ? row_scale[n] : 1.f)
                                 : 0.f;
  }
  fx_bf16x8 h, m, l;
  fx_split8(x, h, m, l);
  yl_bf16_t* o = packed + ((long)ct * nks + ks) * (3 * GX_PART) + gx_slot(j, half);
  *reinterpret_cast<fx_bf16x8*>(o) = h;
  *reinterpret_cast<fx_bf16x8*>(o + GX_PART) = m;
  *reinterpret_cast<fx_bf16x8*>(o + 2 * GX_PART) = l;
}

// K split for few output tiles: enough workgroups to cover the chip, at least 2 stages each
int gx_splits(long M, long N, long K) {
  const long tiles = yl_cdiv(M, GX_BM) * yl_cdiv(N, GX_BN), nst = yl_cdiv(K / 16, GX_KS);
  if (tiles >= 128) return 1;
  long s = 256 / tiles;                 // one workgroup per CU (147 KB of LDS): more than 256 would run in two rounds
  if (s > nst / 2) s = nst / 2;
  return s < 1 ? 1 : (int)s;
}
}  // namespace

extern "C" size_t yolat_gemm_x6_packed_elems(int64_t N, int64_t K) {
  return (N <= 0 || K <= 0) ? 0 : (size_t)yl_cdiv(N, GX_BN) * (size_t)(K / 16) * 3 * GX_PART;
}
// packed: yolat_gemm_x6_packed_elems(N, K) bfloat16 values, 16-byte aligned; K % 16 == 0.  Once per weight version.
static int gx_pack(const float* W, int64_t ldw, int64_t N, int64_t K, const float* row_scale, int transposed,
                   uint16_t* packed, yolat_stream_t stream, int64_t kvalid = -1) {
  if (N <= 0 || K <= 0 || !W || !packed || ldw < (transposed ? N : K) || N >= (1LL << 31) - GX_BN || K >= (1LL << 31))
    return YOLAT_E_INVALID;
  if (K % 16 != 0 || !yl_aligned16(packed)) return YOLAT_E_UNSUPPORTED;
  const long items = (long)yl_cdiv(N, GX_BN) * (K / 16) * GX_BN * 2;
  hipLaunchKernelGGL(k_gemm_x6_pack, dim3((unsigned)yl_cdiv(items, 256)), dim3(256), 0, (hipStream_t)stream, W, (long)ldw,
                     (int)N, (int)K, row_scale, transposed, (int)(kvalid < 0 ? K : kvalid), reinterpret_cast<yl_bf16_t*>(packed));
  YL_LAUNCH_CHECK();
  return 0;
}
extern "C" int yolat_gemm_x6_pack(const float* W, int64_t ldw, int64_t N, int64_t K, const float* row_scale,
                                  uint16_t* packed, yolat_stream_t stream) {
  return gx_pack(W, ldw, N, K, row_scale, 0, packed, stream);
}
// the weight given transposed: Wt [K, N] row-major (ldw >= N) — for  dX [M, N] = dY [M, K] . Wt  of a Linear's backward
extern "C" int yolat_gemm_x6_pack_t(const float* Wt, int64_t ldw, int64_t N, int64_t K, uint16_t* packed,
                                    yolat_stream_t stream) {
  return gx_pack(Wt, ldw, N, K, nullptr, 1, packed, stream);
}
// fp32 elements of the split-K workspace yolat_gemm_x6 needs for this shape (0: none)
extern "C" size_t yolat_gemm_x6_work_elems(int64_t M, int64_t N, int64_t K) {
  if (M <= 0 || N <= 0 || K < 16) return 0;
  const int s = gx_splits(M, N, K);
  return s > 1 ? (size_t)s * (size_t)M * (size_t)N : 0;
}
// out [M, N] = act(A [M, K] . W'^T + shift): W' packed by yolat_gemm_x6_pack, shift NULL = none, relu != 0 = ReLU.
// K % 16 == 0, lda % 4 == 0, A / Wp 16-byte aligned; work: yolat_gemm_x6_work_elems(M, N, K) floats (may be NULL
// when that is 0).
static int gx_run(const float* A, int64_t lda, int64_t M, int64_t K, const uint16_t* Wp, const float* shift, int relu,
                  int64_t N, float* out, int64_t ldo, float* work, float* stats, yolat_stream_t stream);
extern "C" int yolat_gemm_x6(const float* A, int64_t lda, int64_t M, int64_t K, const uint16_t* Wp, const float* shift,
                             int relu, int64_t N, float* out, int64_t ldo, float* work, yolat_stream_t stream) {
  return gx_run(A, lda, M, K, Wp, shift, relu, N, out, ldo, work, nullptr, stream);
}
// the same with the BatchNorm partial statistics of the (pre-activation) output, yolat_linear_fwd's `stats` layout; only
// for shapes that need no K split (yolat_gemm_x6_work_elems(M, N, K) == 0), YOLAT_E_UNSUPPORTED otherwise
extern "C" int yolat_gemm_x6_stats(const float* A, int64_t lda, int64_t M, int64_t K, const uint16_t* Wp, const float* bias,
                                   int64_t N, float* out, int64_t ldo, float* stats, yolat_stream_t stream) {
  if (!stats) return YOLAT_E_INVALID;
  if (M > 0 && N > 0 && K >= 16 && gx_splits(M, N, K) != 1) return YOLAT_E_UNSUPPORTED;
  return gx_run(A, lda, M, K, Wp, bias, 0, N, out, ldo, nullptr, stats, stream);
}
static int gx_run(const float* A, int64_t lda, int64_t M, int64_t K, const uint16_t* Wp, const float* shift, int relu,
                  int64_t N, float* out, int64_t ldo, float* work, float* stats, yolat_stream_t stream) {
  if (M <= 0 || N <= 0 || K <= 0 || !A || !Wp || !out) return YOLAT_E_INVALID;
  if (lda < K || ldo < N || M >= (1LL << 31) - GX_BM || N >= (1LL << 31) - GX_BN) return YOLAT_E_INVALID;
  if (K % 16 != 0 || lda % 4 != 0 || !yl_aligned16(A) || !yl_aligned16(Wp)) return YOLAT_E_UNSUPPORTED;
  const long tm = yl_cdiv(M, GX_BM), tn = yl_cdiv(N, GX_BN);
  if (tm * tn >= (1LL << 31)) return YOLAT_E_UNSUPPORTED;
  const int S = gx_splits(M, N, K);
  const int nst = (int)yl_cdiv(K / 16, GX_KS);
  hipStream_t st = (hipStream_t)stream;
  const yl_bf16_t* wp = reinterpret_cast<const yl_bf16_t*>(Wp);
  if (S == 1) {
    hipLaunchKernelGGL(k_gemm_x6<false>, dim3((unsigned)(tm * tn), 1), dim3(512), 0, st, A, (long)lda, (int)M, (int)K, wp,
                       shift, relu, (int)N, out, (long)ldo, (int)tn, nst, stats);
    YL_LAUNCH_CHECK();
    return 0;
  }
  if (!work) return YOLAT_E_INVALID;
  const int per = yl_cdiv(nst, S), S2 = yl_cdiv(nst, per);
  hipLaunchKernelGGL(k_gemm_x6<true>, dim3((unsigned)(tm * tn), (unsigned)S2), dim3(512), 0, st, A, (long)lda, (int)M,
                     (int)K, wp, shift, relu, (int)N, work, (long)N, (int)tn, per, (float*)nullptr);
  YL_LAUNCH_CHECK();
  hipLaunchKernelGGL(k_gemm_x6_reduce, dim3((unsigned)yl_cdiv(M * N, 256)), dim3(256), 0, st, work, S2, (long)M, (int)N,
                     shift, relu, out, (long)ldo);
  YL_LAUNCH_CHECK();
  return 0;
}

// ------------------------------------------------------------------------------------------------
// Weight gradient of a wide Linear on the same kernel (round 3):  dW [Nout, K] = dY [M, Nout]^T . A [M, K],  db = column
// sums of dY (torch.nn.Linear's backward; the per-proposal classifier's first layer, architecture3cc_rpn_gp_iter2.py:91,
// 127: M = P proposals, K = 2304, Nout = 512).  Both operands have the reduction index M as their ROW index, so dY is
// transposed into the GEMM's row operand ([Nout, Mp], Mp = M rounded up to 16, zero filled; its 64-row tiles also give the
// column sums, reduced in a fixed order) and A is packed as the "weight given transposed" (yolat_gemm_x6_pack_t's form).
// Replaces the split-row fp32 v_mfma_f32_32x32x2 kernel (k_gemm_tn: 231 us at P = 8000).
// ------------------------------------------------------------------------------------------------
namespace {
// out [C][ldo] = in [R][C]^T for the 64 x 64 tile (blockIdx.x: row tile, blockIdx.y: column tile); colpart [row tile][C] =
// the tile's column sums, rows ascending
__global__ void __launch_bounds__(256) k_x6_transpose_colsum(const float* __restrict__ in, long ldi, int R, int C,
                                                             float* __restrict__ out, long ldo, int Rp,
                                                             float* __restrict__ colpart) {
  __shared__ float t[64][65];
  const int r0 = blockIdx.x * 64, c0 = blockIdx.y * 64, tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const int r = r0 + ty + 4 * i, c = c0 + tx;
    t[ty + 4 * i][tx] = (r < R && c < C) ? in[(long)r * ldi + c] : 0.f;
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const int c = c0 + ty + 4 * i, r = r0 + tx;
    if (c < C && r < Rp) out[(long)c * ldo + r] = t[tx][ty + 4 * i];
  }
  if (colpart != nullptr && threadIdx.x < 64 && c0 + tx < C) {
    float s = 0.f;
#pragma unroll 8
    for (int i = 0; i < 64; ++i) s += t[i][tx];
    colpart[(long)blockIdx.x * C + c0 + tx] = s;
  }
}
// db[c] (+)= sum over the T row tiles: 64 columns x 16 tile lanes per workgroup, lane q sums tiles q, q + 16, ... and the
// sixteen lane sums are added in lane order (fixed order: deterministic)
__global__ void __launch_bounds__(1024) k_x6_colsum_final(const float* __restrict__ colpart, int T, int C,
                                                          float* __restrict__ db, int accumulate) {
  __shared__ float part[16][64];
  const int cx = threadIdx.x & 63, q = threadIdx.x >> 6, c = blockIdx.x * 64 + cx;
  float s = 0.f;
  if (c < C)
    for (int t = q; t < T; t += 16) s += colpart[(long)t * C + c];
  part[q][cx] = s;
  __syncthreads();
  if (q == 0 && c < C) {
    float v = 0.f;
#pragma unroll
    for (int j = 0; j < 16; ++j) v += part[j][cx];
    db[c] = accumulate ? db[c] + v : v;
  }
}
inline size_t gx_al4(size_t n) { return (n + 3) / 4 * 4; }
}  // namespace

// worthwhile and possible for this shape?  (the split path pays two extra passes over the operands)
bool yl_bwd_w_x6_ok(int64_t M, int64_t Nout, int64_t K) {
  return !yl_strict_fp32() && M >= 2048 && Nout >= 128 && K >= 128 && (double)M * (double)Nout * (double)K >= 4.0e9 && M < (1LL << 30);
}
size_t yl_bwd_w_x6_work_elems(int64_t M, int64_t Nout, int64_t K) {
  const int64_t Mp = (M + 15) / 16 * 16;
  return gx_al4((size_t)Nout * Mp) + gx_al4((yolat_gemm_x6_packed_elems(K, Mp) + 1) / 2) +
         gx_al4(yolat_gemm_x6_work_elems(Nout, K, Mp)) + gx_al4((size_t)yl_cdiv(M, 64) * Nout);
}
int yl_bwd_w_x6(const float* dY, int64_t lddy, int64_t M, int64_t Nout, const float* A, int64_t lda, int64_t K, float* dW,
                int64_t lddw, float* db, int accumulate_db, float* work, hipStream_t st) {
  const int64_t Mp = (M + 15) / 16 * 16;
  float* dYT = work;
  uint16_t* packed = reinterpret_cast<uint16_t*>(dYT + gx_al4((size_t)Nout * Mp));
  float* gwork = reinterpret_cast<float*>(packed) + gx_al4((yolat_gemm_x6_packed_elems(K, Mp) + 1) / 2);
  float* colpart = gwork + gx_al4(yolat_gemm_x6_work_elems(Nout, K, Mp));
  const int T = yl_cdiv(M, 64);
  hipLaunchKernelGGL(k_x6_transpose_colsum, dim3((unsigned)yl_cdiv(Mp, 64), (unsigned)yl_cdiv(Nout, 64)), dim3(256), 0, st, dY,
                     (long)lddy, (int)M, (int)Nout, dYT, (long)Mp, (int)Mp, db ? colpart : (float*)nullptr);
  YL_LAUNCH_CHECK();
  if (db) {
    hipLaunchKernelGGL(k_x6_colsum_final, dim3((unsigned)yl_cdiv(Nout, 64)), dim3(1024), 0, st, colpart, T, (int)Nout, db,
                       accumulate_db);
    YL_LAUNCH_CHECK();
  }
  { const int rc = gx_pack(A, lda, K, Mp, nullptr, 1, packed, (yolat_stream_t)st, M); if (rc != 0) return rc; }
  return gx_run(dYT, Mp, Nout, Mp, packed, nullptr, 0, K, dW, lddw, gwork, nullptr, (yolat_stream_t)st);
}
