// fusion_x6.hip — eval-mode fusion block + per-proposal max pooling (architecture3cc_rpn_gp_iter2.py:61-63,122) with
// the [N,128] x [128,1024] GEMM as an fp32 GEMM EMULATED on the bf16 matrix cores (round 2).
//
// Why: the fp32-input MFMA of gfx950 executes on the SIMD's vector ALUs at the fp32 vector rate ("fp32 MFMA
// and the vector ALU"): the round-1 kernel (k_gemm_nt_two, 64x64 tiles) sits at 0.48 of that peak at cfg 2 and its
// epilogue VALU work adds to, not under, its MFMA time.  Both operands split exactly into three bfloat16 terms
// (truncation: 8 + 8 + 8 significand bits) and six products per 16 k — a_h b_l, a_l b_h, a_m b_m, a_m b_h, a_h b_m,
// a_h b_h, the three O(2^-24) ones dropped — give the fp32 result up to the summation order at 16/6 = 2.7x the
// fp32-MFMA rate, on a pipe that runs beside the VALU.
//
// Structure: a 512-thread workgroup (8 waves) owns 256 rows.  Every wave splits ITS 32 rows x K of A once into
// registers (3 x K/16 fragments = 96 VGPRs at K = 128) and keeps them for all the column tiles it walks; the weights —
// split once per weight version by yolat_split_bf16x3, BatchNorm scale folded into their rows — stream through a
// double-buffered LDS tile (3 x 64 columns x K bf16 = 48 KB, row stride K+8: conflict-free ds_read_b128), the next
// tile's 16-byte pieces loaded into registers while the current tile's 96 MFMAs per wave and its pooling epilogue
// run; one barrier per column tile.  W is re-streamed once per 256 rows (4x less L2 -> LDS traffic than 64-row
// tiles: at N = 200 k, 0.65 GB instead of 2.6 GB).  Pooling epilogue: the run-length integer atomicMax of
// bf16_eval.hip's rows kernel (values >= 0: int order == float order, exact, order-independent).
// fusion_block_super (P rows, its own weights, plain relu store) is a second problem of the same launch.
#include "x6.hpp"
#include "segmax.hpp"
#include <algorithm>
#include <map>
#include <queue>
#include <vector>

#ifdef YOLAT_FX_STAMPS
// debug build only (tools/exp/r06_fx_stamps.sh): wall-clock stamps (100 MHz) of thread 0 of every workgroup of the K = 128
// rows kernel, read back through yolat_debug_fx_stamps — where a workgroup's time goes
__device__ long long fx_stamps_d[4096 * 32];
#define FX_STAMP(k) do { if (KD == 128 && threadIdx.x == 0 && blockIdx.x < 4096) fx_stamps_d[blockIdx.x * 32 + (k)] = wall_clock64(); } while (0)
extern "C" int yolat_debug_fx_stamps(long long* out, int n) {
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(fx_stamps_d), sizeof(long long) * (size_t)n);
}
#define FX_STAMP_META()                                                                          \
  do {                                                                                           \
    if (KD == 128 && threadIdx.x == 0 && blockIdx.x < 4096) {                                    \
      fx_stamps_d[blockIdx.x * 32 + 21] = (long long)ngl;                                        \
      fx_stamps_d[blockIdx.x * 32 + 22] = (long long)(small ? 1 : 0);                            \
      unsigned xcc_;                                                                             \
      asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc_));                        \
      fx_stamps_d[blockIdx.x * 32 + 23] = (long long)xcc_;                                       \
    }                                                                                            \
  } while (0)
#else
#define FX_STAMP(k) do { } while (0)
#define FX_STAMP_META() do { } while (0)
#endif
namespace {
// training-mode epilogue: the extreme of s*z per run with its (lowest) row, as the key of wave_epilogue's key64 branch
// (common.hpp) — same key, same tie rule
__device__ __forceinline__ unsigned long long fx_key(float z, bool neg, unsigned row) {
  unsigned u = __float_as_uint(neg ? -z : z);
  u = (u & 0x80000000u) ? ~u : (u | 0x80000000u);                // order-preserving float -> uint
  return ((unsigned long long)u << 32) | (unsigned long long)(0xFFFFFFFFu - row);
}
// Round 6: the run's extreme is tracked as a FLOAT with its register index (xor with the column's sign mask, one compare,
// two selects per element) and turned into the 64-bit key only where a run ends — the key per element (sign select,
// order-preserving transform, 64-bit compare + two selects) was ~8 vector instructions per element, a quarter of the
// kernel's MFMA time at two waves per SIMD.  Same key, same tie rule (strict >: the lowest row of equal values wins);
// differences only where IEEE comparison and the key order differ: -0 / +0 inside one lane's run count as equal (the
// lower row wins, as in torch_scatter's `>` walk), and a NaN never replaces a number.
struct FxKeyRun { float cur0, cur1; int at0, at1; };
__device__ __forceinline__ void fx_key_begin(FxKeyRun& k) { k.cur0 = -INFINITY; k.cur1 = -INFINITY; k.at0 = 0; k.at1 = 0; }
// one row of the walk (r is a compile-time constant after unrolling); m0 / m1: sign-bit masks of the two columns
__device__ __forceinline__ void fx_key_row(int r, float v0, float v1, FxKeyRun& k, unsigned long long* keys, unsigned ldk,
                                           const int* segs, int lhi, unsigned c0, bool ok0, bool ok1, unsigned m0, unsigned m1,
                                           unsigned rbase, const FxRuns& sr) {
  const float x0 = __uint_as_float(__float_as_uint(v0) ^ m0), x1 = __uint_as_float(__float_as_uint(v1) ^ m1);
  const bool b0 = x0 > k.cur0, b1 = x1 > k.cur1;
  k.cur0 = b0 ? x0 : k.cur0; k.at0 = b0 ? r : k.at0;
  k.cur1 = b1 ? x1 : k.cur1; k.at1 = b1 ? r : k.at1;
  if ((sr.uflush >> r) & 1u) {
    if ((sr.flush_bits >> r) & 1u) {
      unsigned long long* o = keys + ((unsigned long)(unsigned)segs[(r & 3) + 8 * (r >> 2) + 4 * lhi] * ldk + c0);
      if (ok0) atomicMax(o, fx_key(k.cur0, false, rbase + (k.at0 & 3) + 8 * (k.at0 >> 2)));
      if (ok1) atomicMax(o + 32, fx_key(k.cur1, false, rbase + (k.at1 & 3) + 8 * (k.at1 >> 2)));
      k.cur0 = -INFINITY; k.cur1 = -INFINITY;            // the lane's next row starts a new run
      k.at0 = r + 1; k.at1 = r + 1;
    }
  }
}
__device__ __forceinline__ void fx_key64(const f32x16& acc0, const f32x16& acc1, unsigned long long* keys, unsigned ldk,
                                         const int* segs, int lhi, unsigned c0, bool ok0, bool ok1, unsigned m0, unsigned m1,
                                         unsigned rbase, const FxRuns& sr) {
  FxKeyRun k;
  fx_key_begin(k);
#pragma unroll
  for (int r = 0; r < 16; ++r) fx_key_row(r, acc0[r], acc1[r], k, keys, ldk, segs, lhi, c0, ok0, ok1, m0, m1, rbase, sr);
}
}  // namespace

// One GEMM problem of the launch: rows of A against the pre-split weights [F, KD]; seg != NULL -> pooling epilogue into
// `out` (pool); else plain store of act(.) — column tiles [0, ct2) into out [rows, ldo], column tiles [ct2, ...) into
// out2 [rows, ldo2] (its column 0 = column 64 * ct2 of the product), act = ReLU when relu != 0.
struct FxProb {
  const float* A; long lda; int N;
  const yl_bf16_t *Wh, *Wm, *Wl;
  const float* tfold;
  const int* seg;
  float* out; long ldo;
  float* out2; long ldo2; int ct2;
  int F, relu;
  int tm, groups, ng;
  int full = 0;          // the first `full` row tiles (a multiple of 8) are walked WHOLE by one workgroup each (fx_plan)
  // training-mode pooling (fusion_train.hip): key64 != NULL (with seg) -> for every (proposal, column) the row with the
  // largest s*z (s = sign of sgn[column], z = the accumulator) is recorded by a 64-bit atomicMax on
  // orderable(s*z) << 32 | ~row (ties -> lowest row); nothing else is written
  unsigned long long* key64; const float* sgn;
  // training-mode Linear (yolat_linear_fwd_rows_x6): optional prologue on A, v = max(a * a_scale[k] + a_shift[k],
  // a_floor) (BatchNorm + ReLU of the producer), and optional BatchNorm partial statistics of the stored values:
  // stats[(row / 32) * F + col] = (sum, M2) of the 32-row group (the layout yolat_bn_finalize reads)
  const float *a_scale, *a_shift; float a_floor;
  float* stats;
};

// RIDER = false: the rider argument is never read (an instantiation that reads it keeps more kernel arguments live:
// 75 -> 121 spilled SGPRs, and the cfg-5 fusion launch went 332 -> 365 us until the two were separated)
// Threads per workgroup (32 rows per wave).  K = 128 (fusion blocks): 512 threads = 256 rows, one workgroup per CU (248
// registers, 105 KB of LDS).  K = 64 (node side, training Linear: one to three column tiles per workgroup, store-heavy):
// 256 threads = 128 rows and 75 KB of LDS, so that TWO workgroups share a CU and one's prologue / stores run under the
// other's MFMAs.
template <int KD> struct FxShape { static constexpr int T = KD == 64 ? 256 : 512, ROWS = T / 2; };
template <int KD, bool RIDER = false>
__global__ void __launch_bounds__(FxShape<KD>::T, 2) k_fusion_rows_x6(FxProb p0, FxProb p1, PoolRider rider) {
  constexpr int T = FxShape<KD>::T, NWAVE = T / 64;
  constexpr int KS = KD / 16, RS = KD + 8, CPR = KD / 8;    // k steps, LDS row stride (bf16), 16-byte chunks per row
  constexpr int CHUNKS = 3 * 64 * CPR, NW = CHUNKS / T;     // 16-byte chunks of one W tile, per thread
  static_assert(CHUNKS % T == 0, "W tile / thread mismatch");
  constexpr int TK = 64, TS = TK + 4;                       // the prologue's transposition tile: [32 rows][TK + 4] floats per wave
  constexpr int SMEM_W = 2 * 3 * 64 * RS * 2, SMEM_T = NWAVE * 32 * TS * 4;
  __shared__ __attribute__((aligned(16))) char smem[SMEM_W > SMEM_T ? SMEM_W : SMEM_T];
  yl_bf16_t (*Ws)[3 * 64 * RS] = reinterpret_cast<yl_bf16_t (*)[3 * 64 * RS]>(smem);
  __shared__ int seg_s[T / 2];
  constexpr int FX_STG_LD = 36;
  // per-wave staging tile of the plain-store epilogue — KD = 64 only (node side, training Linear): the KD = 128 kernel
  // sits at 248 registers and spilled with it
  __shared__ __attribute__((aligned(16))) float st_s[KD == 64 ? NWAVE * 32 * FX_STG_LD : 4];
  const int tid = threadIdx.x;
  // the small problem's workgroups come first (padded to a multiple of 8 so that the big problem keeps its
  // id % 8 = XCD alignment): they start with the first round of workgroups instead of forming a tail
  const int n1 = p1.tm * p1.groups, n1p = (n1 + 7) & ~7;
  const int id = blockIdx.x;
  // workgroups past both problems: pooling-prologue rider (common.hpp; reads A like the GEMM, writes columns of the
  // pooled rows that nothing in this launch reads)
  // the big problem: p0.full whole-row-tile workgroups first, then the remaining row tiles split into column groups
  const int n0t = (p0.tm - p0.full) * p0.groups, n0 = p0.full + n0t;
  if constexpr (RIDER) {
    if (id >= n1p + n0) {
      yl_pool_rider(rider, id - (n1p + n0), rider.blocks, tid, T);
      return;
    }
  }
  int logical;
  bool whole = false;
  if (id < n1p) {
    if (id >= n1) return;
    logical = id;
  } else if (id - n1p < p0.full) {
    logical = id - n1p;
    whole = true;
  } else {
    // (row tile, column group) pairs, column group fastest, dealt to the XCDs in contiguous ranges
    const int j0 = id - n1p - p0.full;
    const int chunk = n0t >> 3, rem = n0t & 7;
    const int xcd = j0 & 7, slot = j0 >> 3;
    logical = xcd * chunk + (xcd < rem ? xcd : rem) + slot;
  }
  // the problem's fields as wave-uniform scalars (a reference to "p0 or p1" would make the compiler index the
  // kernel arguments through memory)
  const bool small = id < n1p;
  struct { const float* A; long lda; int N; const float* tfold; const int* seg; float* out; long ldo; float* out2;
           long ldo2; int ct2, relu, groups, ng; unsigned long long* key64; const float* sgn; const float *a_scale,
           *a_shift; float a_floor; float* stats; } P;
  P.A = small ? p1.A : p0.A;
  P.lda = small ? p1.lda : p0.lda;
  P.N = small ? p1.N : p0.N;
  P.tfold = small ? p1.tfold : p0.tfold;
  P.seg = small ? p1.seg : p0.seg;
  P.out = small ? p1.out : p0.out;
  P.ldo = small ? p1.ldo : p0.ldo;
  P.out2 = small ? p1.out2 : p0.out2;
  P.ldo2 = small ? p1.ldo2 : p0.ldo2;
  P.ct2 = small ? p1.ct2 : p0.ct2;
  P.relu = small ? p1.relu : p0.relu;
  P.key64 = small ? p1.key64 : p0.key64;
  P.sgn = small ? p1.sgn : p0.sgn;
  P.a_scale = small ? p1.a_scale : p0.a_scale;
  P.a_shift = small ? p1.a_shift : p0.a_shift;
  P.a_floor = small ? p1.a_floor : p0.a_floor;
  P.stats = small ? p1.stats : p0.stats;
  const int F = small ? p1.F : p0.F;
  P.groups = small ? p1.groups : p0.groups;
  P.ng = small ? p1.ng : p0.ng;
  const float* __restrict__ A = P.A;
  const int N = P.N;
  const int tn = (F + 63) >> 6;
  const int rt = whole ? logical : (small ? 0 : p0.full) + logical / P.groups;
  const int cg = whole ? 0 : logical % P.groups;
  const int ct0 = cg * P.ng;
  const int ngl = whole ? tn : yl_min(P.ng, tn - ct0);
  if (ngl <= 0) return;

  const int wave = tid >> 6, lane = tid & 63, l31 = lane & 31, lhi = lane >> 5;
  const int row0 = rt * (T / 2) + wave * 32;
  FX_STAMP(0);
  // ---- this wave's 32 rows of A, split once.  Loaded row by row (a row's 64 floats of the pass = 16 lanes x 16 bytes:
  // whole cache lines per instruction, eight loads in flight) and turned into the MFMA operand order (lane = row, 8
  // consecutive k) through a [32][TK + 4] fp32 tile in LDS — the weights' buffers, not yet in use.  (Reading the operand
  // order straight from memory, every lane its own row, as a chain of dependent 16-byte loads was 126 of the kernel's
  // 350 us at N = 200 k: profiles/r03_fusion_x6_ablation.txt.)
  fx_bf16x8 Ah[KS], Am[KS], Al[KS];
  {
    float* Tt = reinterpret_cast<float*>(smem) + wave * (32 * TS);
    const int lr = lane >> 4, lc = (lane & 15) * 4;
#pragma unroll
    for (int q = 0; q < KD / TK; ++q) {
      float4 v[8];
#pragma unroll
      for (int i = 0; i < 8; ++i)
        v[i] = *reinterpret_cast<const float4*>(A + (long)yl_min(row0 + 4 * i + lr, N - 1) * P.lda + TK * q + lc);
#pragma unroll
      for (int i = 0; i < 8; ++i) *reinterpret_cast<float4*>(Tt + (4 * i + lr) * TS + lc) = v[i];
#pragma unroll
      for (int u = 0; u < TK / 16; ++u) {
        const int ks = (TK / 16) * q + u;
        const float4 a0 = *reinterpret_cast<const float4*>(Tt + l31 * TS + 16 * u + 8 * lhi);
        const float4 a1 = *reinterpret_cast<const float4*>(Tt + l31 * TS + 16 * u + 8 * lhi + 4);
        float x[8] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w};
        if (P.a_scale != nullptr) {
          const float* sp = P.a_scale + 16 * ks + 8 * lhi;
          const float* hp = P.a_shift + 16 * ks + 8 * lhi;
#pragma unroll
          for (int e = 0; e < 8; ++e) x[e] = fmaxf(fmaf(x[e], sp[e], hp[e]), P.a_floor);
        }
        fx_split8(x, Ah[ks], Am[ks], Al[ks]);
      }
    }
  }
  FX_STAMP(1);
  const bool pooling = P.seg != nullptr;
  FxRuns runs;
  {
    const int sv = (pooling && row0 + l31 < N) ? P.seg[row0 + l31] : -1;
    if (lhi == 0) seg_s[wave * 32 + l31] = sv;          // read back by the same wave only, after the barrier below
    fx_seg_runs(sv, lhi, runs);
  }
  const int* segs = seg_s + wave * 32;
  const yl_bf16_t* const wparts[3] = {small ? p1.Wh : p0.Wh, small ? p1.Wm : p0.Wm, small ? p1.Wl : p0.Wl};
  // W tile pieces: thread tid moves 16-byte piece (tid + T t) of the [3][64][KD] tile; 64 * CPR is a multiple of
  // T, so the part (hi / mid / lo) of piece t is a compile-time constant
  constexpr int PER = 64 * CPR / T;                         // pieces per thread and part
  static_assert((64 * CPR) % T == 0, "part boundary inside a thread's pieces");
  const unsigned wr0 = (unsigned)tid / CPR, wk = ((unsigned)tid % CPR) * 8;   // row (of the first piece), k offset
  auto load_w = [&](int ct, fx_u32x4* rw) {
#pragma unroll
    for (int t = 0; t < NW; ++t) {
      const int part = t / PER;
      const unsigned r = wr0 + (unsigned)(t % PER) * (T / CPR);
      rw[t] = *reinterpret_cast<const fx_u32x4*>(wparts[part] + (unsigned)yl_min(ct * 64 + (int)r, F - 1) * KD + wk);
    }
  };
  auto store_w = [&](int buf, const fx_u32x4* rw) {
#pragma unroll
    for (int t = 0; t < NW; ++t) {
      const int part = t / PER;
      const unsigned r = wr0 + (unsigned)(t % PER) * (T / CPR);
      *reinterpret_cast<fx_u32x4*>(&Ws[buf][part * 64 * RS + r * RS + wk]) = rw[t];
    }
  };
  fx_u32x4 rw[NW];
  load_w(ct0, rw);
  // shifts of the first tile; later tiles: fetched one tile ahead, BEFORE that tile's W loads, so that no wait on
  // them ever sits behind younger loads or the epilogue's atomics (vmcnt retires in order)
  float t0 = P.tfold[yl_min(ct0 * 64 + l31, F - 1)], t1 = P.tfold[yl_min(ct0 * 64 + 32 + l31, F - 1)];
  FX_STAMP(2);
  __syncthreads();                                          // every wave is done with its transposition tile
  store_w(0, rw);
  __syncthreads();
  FX_STAMP(3);
  // B fragments: column block 0 / 1 x hi / mid / lo; the compiler issues the next k step's reads as registers free up.
  // Small terms first.  The empty asm ties both accumulators after every MFMA: it pins the MFMAs' program order
  // (instruction selection otherwise pairs up MFMAs on the same accumulator) and emits nothing
#define FX_MFMA(acc, a, b)                                         \
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0); \
  asm volatile("" : "+v"(acc0), "+v"(acc1))
#define FX_MFMA_A(ks)          \
  FX_MFMA(acc0, Al[ks], bf[0]); \
  FX_MFMA(acc1, Al[ks], bf[1]); \
  FX_MFMA(acc0, Ah[ks], bf[4]); \
  FX_MFMA(acc1, Ah[ks], bf[5]); \
  FX_MFMA(acc0, Am[ks], bf[2]); \
  FX_MFMA(acc1, Am[ks], bf[3])
#define FX_MFMA_B(ks)          \
  FX_MFMA(acc0, Am[ks], bf[0]); \
  FX_MFMA(acc1, Am[ks], bf[1]); \
  FX_MFMA(acc0, Ah[ks], bf[2]); \
  FX_MFMA(acc1, Ah[ks], bf[3]); \
  FX_MFMA(acc0, Ah[ks], bf[0]); \
  FX_MFMA(acc1, Ah[ks], bf[1])
  if (pooling) {
    // ---- pooling modes, software-pipelined (round 6).  Stamps of one cfg-2 workgroup (tools/exp/r06_fx_stamps.py): per
    // column tile 2.0-3.0 us of MFMAs (the matrix pipe's rate at two waves per SIMD), then 1.0-2.0 us of pooling epilogue
    // (run walk + global atomics, matrix pipe idle), 0.3-0.5 us of W staging / waits and 0.4-0.8 us of barrier — 47 % of a
    // tile without an MFMA in flight.  A wave issues in order, so its epilogue only hides under MFMAs that FOLLOW it in its
    // own instruction stream: the finished tile's accumulators move to a second pair (32 v_mov) and their walk is spread,
    // RPS rows per k step, over the FIRST HALF of the next tile's k loop, between its MFMAs; the atomics are acknowledged
    // during the second half, so the vmcnt(0) in front of the barrier still costs nothing.  Only the last tile's
    // epilogue runs in the open.
    const bool train = P.key64 != nullptr;
    constexpr int ES = KS / 2, RPS = 16 / ES;          // k steps that carry epilogue rows, rows per step
    f32x16 acc0, acc1, prv0, prv1;
#pragma unroll
    for (int r = 0; r < 16; ++r) { acc0[r] = 0.f; acc1[r] = 0.f; }
    // sign masks of the columns (training): nm = the tile about to start (fetched one tile ahead, like the shifts), m = the
    // tile whose products accumulate, pm = the tile whose walk runs
    unsigned m0 = 0u, m1 = 0u, pm0 = 0u, pm1 = 0u, nm0 = 0u, nm1 = 0u;
    if (train) {
      nm0 = P.sgn[yl_min(ct0 * 64 + l31, F - 1)] < 0.f ? 0x80000000u : 0u;
      nm1 = P.sgn[yl_min(ct0 * 64 + 32 + l31, F - 1)] < 0.f ? 0x80000000u : 0u;
    }
    const unsigned rbase = (unsigned)(row0 + 4 * lhi);
    for (int j = 0; j < ngl; ++j) {
      const int ct = ct0 + j, buf = j & 1;
      const int c0 = ct * 64 + l31, c1 = c0 + 32;
      const bool prev = j > 0;
#pragma unroll
      for (int r = 0; r < 16; ++r) { prv0[r] = acc0[r]; prv1[r] = acc1[r]; acc0[r] = t0; acc1[r] = t1; }
      pm0 = m0; pm1 = m1; m0 = nm0; m1 = nm1;
      if (j + 1 < ngl) {
        t0 = P.tfold[yl_min(c0 + 64, F - 1)];
        t1 = P.tfold[yl_min(c1 + 64, F - 1)];
        if (train) {
          nm0 = P.sgn[yl_min(c0 + 64, F - 1)] < 0.f ? 0x80000000u : 0u;
          nm1 = P.sgn[yl_min(c1 + 64, F - 1)] < 0.f ? 0x80000000u : 0u;
        }
      }
      float cur0 = 0.f, cur1 = 0.f;
      FxKeyRun kr;
      fx_key_begin(kr);
      const unsigned pc0 = (unsigned)(c0 - 64);
      const bool pok0 = c0 - 64 < F, pok1 = c1 - 64 < F;
      // rows [r_lo, r_lo + n) of the previous tile's walk
      auto epi_rows = [&](int r_lo, int n) {
#pragma unroll
        for (int i = 0; i < n; ++i) {
          const int r = r_lo + i;
          if (train) fx_key_row(r, prv0[r], prv1[r], kr, P.key64, (unsigned)F, segs, lhi, pc0, pok0, pok1, pm0, pm1, rbase, runs);
          else fx_segmax_row(r, prv0[r], prv1[r], cur0, cur1, P.out, (unsigned)P.ldo, segs, lhi, pc0, pok0, pok1, runs);
        }
      };
      const yl_bf16_t* wb = &Ws[buf][l31 * RS + 8 * lhi];
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        fx_bf16x8 bf[6];                               // [0] b0h [1] b1h [2] b0m [3] b1m [4] b0l [5] b1l
#pragma unroll
        for (int q = 0; q < 6; ++q) bf[q] = *reinterpret_cast<const fx_bf16x8*>(wb + q * 32 * RS + 16 * ks);
        // the next tile's W pieces go out when the walk is through: their 24 registers are the ones the previous tile's
        // accumulators just left (with both live the kernel needs 271); half a tile of MFMAs covers their latency
        if (ks == ES && j + 1 < ngl) load_w(ct + 1, rw);
        FX_MFMA_A(ks);
        if (ks < ES && prev) epi_rows(ks * RPS, RPS / 2);
        FX_MFMA_B(ks);
        if (ks < ES && prev) epi_rows(ks * RPS + RPS / 2, RPS / 2);
      }
      FX_STAMP(4 + 4 * (j & 3));
      if (j + 1 < ngl) store_w(buf ^ 1, rw);
      __builtin_amdgcn_s_waitcnt(0x0F70);              // vmcnt(0): shifts, masks, W pieces, the walk's atomics (half a tile old)
      FX_STAMP(5 + 4 * (j & 3));
      FX_STAMP(6 + 4 * (j & 3));
      __syncthreads();
      FX_STAMP(7 + 4 * (j & 3));
    }
    {   // the last tile's epilogue
      const unsigned lc0 = (unsigned)((ct0 + ngl - 1) * 64 + l31);
      const bool lok0 = (int)lc0 < F, lok1 = (int)lc0 + 32 < F;
      if (train) fx_key64(acc0, acc1, P.key64, (unsigned)F, segs, lhi, lc0, lok0, lok1, m0, m1, rbase, runs);
      else fx_segmax2(acc0, acc1, P.out, (unsigned)P.ldo, segs, lhi, lc0, lok0, lok1, runs);
    }
    FX_STAMP(20);
    FX_STAMP_META();
    return;
  }
  for (int j = 0; j < ngl; ++j) {
    const int ct = ct0 + j, buf = j & 1;
    const int c0 = ct * 64 + l31, c1 = c0 + 32;
    f32x16 acc0, acc1;
#pragma unroll
    for (int r = 0; r < 16; ++r) { acc0[r] = t0; acc1[r] = t1; }
    if (j + 1 < ngl) {
      t0 = P.tfold[yl_min(c0 + 64, F - 1)];
      t1 = P.tfold[yl_min(c1 + 64, F - 1)];
      load_w(ct + 1, rw);                              // in flight while the MFMAs below run
    }
    const yl_bf16_t* wb = &Ws[buf][l31 * RS + 8 * lhi];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      fx_bf16x8 bf[6];                                 // [0] b0h [1] b1h [2] b0m [3] b1m [4] b0l [5] b1l
#pragma unroll
      for (int q = 0; q < 6; ++q) {
        bf[q] = *reinterpret_cast<const fx_bf16x8*>(wb + q * 32 * RS + 16 * ks);
      }
      FX_MFMA_A(ks);
      FX_MFMA_B(ks);
    }
    // next W tile into the other buffer (its readers finished before the last barrier) BEFORE the epilogue: the
    // wait for its loads then never includes the epilogue's atomics / stores
    FX_STAMP(4 + 4 * (j & 3));
    if (j + 1 < ngl) store_w(buf ^ 1, rw);
    // everything still in flight here (next tile's shifts and W pieces, the previous tile's atomics) was issued a
    // whole MFMA phase ago: waiting now costs nothing and keeps the compiler from waiting on the shifts at the top
    // of the next tile, where the wait would also cover this tile's atomics (vmcnt retires in order)
    __builtin_amdgcn_s_waitcnt(0x0F70);                // vmcnt(0)
    FX_STAMP(5 + 4 * (j & 3));
    {
      // the row base goes through an opaque asm so that the 16 row addresses are recomputed here instead of being
      // kept in 32 registers across the MFMA loop
      unsigned rb = (unsigned)(row0 + 4 * lhi);
      asm volatile("" : "+v"(rb));
      if (P.stats != nullptr) {
        // BatchNorm partial statistics of this wave's 32-row group, as wave_epilogue (common.hpp) takes them
        int cnt = N - row0;
        cnt = cnt > 32 ? 32 : cnt;
        if (cnt > 0) {
          float s0 = 0.f, s1 = 0.f;
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const bool ok = row0 + (r & 3) + 8 * (r >> 2) + 4 * lhi < N;
            s0 += ok ? acc0[r] : 0.f;
            s1 += ok ? acc1[r] : 0.f;
          }
          s0 += __shfl_xor(s0, 32); s1 += __shfl_xor(s1, 32);
          const float m0 = s0 / (float)cnt, m1 = s1 / (float)cnt;
          float q0 = 0.f, q1 = 0.f;
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const bool ok = row0 + (r & 3) + 8 * (r >> 2) + 4 * lhi < N;
            const float d0 = acc0[r] - m0, d1 = acc1[r] - m1;
            q0 += ok ? d0 * d0 : 0.f;
            q1 += ok ? d1 * d1 : 0.f;
          }
          q0 += __shfl_xor(q0, 32); q1 += __shfl_xor(q1, 32);
          if (lhi == 0) {
            float2* sp = reinterpret_cast<float2*>(P.stats) + (long)(row0 >> 5) * F;
            if (c0 < F) sp[c0] = make_float2(s0, q0);
            if (c1 < F) sp[c1] = make_float2(s1, q1);
          }
        }
      }
      const bool second = ct >= P.ct2;
      float* const ob = second ? P.out2 : P.out;
      const unsigned long old = (unsigned long)(second ? P.ldo2 : P.ldo);
      const int cb = second ? 64 * P.ct2 : 0;
      const float lo = P.relu ? 0.f : -INFINITY;
      // full tile, 16-byte aligned rows: the two 32 x 32 blocks go row-major through this wave's staging tile, so that
      // every lane stores 16 bytes and one instruction covers 8 rows x 128 bytes (in the accumulator layout a store
      // instruction is 4 bytes per lane, 2 rows x 128 bytes: 4x the instructions for the node side's 205 MB)
      if (KD == 64 && row0 + 32 <= N && ct * 64 + 64 <= F && (old & 3) == 0 && ((uintptr_t)ob & 15) == 0) {
        float* stg = st_s + wave * (32 * FX_STG_LD);
#pragma unroll
        for (int blk = 0; blk < 2; ++blk) {
#pragma unroll
          for (int r = 0; r < 16; ++r)
            stg[((r & 3) + 8 * (r >> 2) + 4 * lhi) * FX_STG_LD + l31] = fmaxf(blk == 0 ? acc0[r] : acc1[r], lo);
#pragma unroll
          for (int t = 0; t < 4; ++t) {
            const int chunk = lane + 64 * t, R = chunk >> 3, c = (chunk & 7) * 4;
            *reinterpret_cast<float4*>(ob + (unsigned long)(row0 + R) * old - cb + ct * 64 + 32 * blk + c) =
                *reinterpret_cast<const float4*>(stg + R * FX_STG_LD + c);
          }
        }
      } else
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const unsigned row = rb + (r & 3) + 8 * (r >> 2);
        if ((int)row < N) {
          float* o = ob + (unsigned long)row * old - cb;
          if (c0 < F) o[c0] = fmaxf(acc0[r], lo);
          if (c1 < F) o[c1] = fmaxf(acc1[r], lo);
        }
      }
    }
    FX_STAMP(6 + 4 * (j & 3));
    __syncthreads();
    FX_STAMP(7 + 4 * (j & 3));
  }
  FX_STAMP(20);
  FX_STAMP_META();
#undef FX_MFMA_A
#undef FX_MFMA_B
#undef FX_MFMA
}

// W [rows, cols] fp32 (optionally scaled per row) -> three bfloat16 matrices hi / mid / lo [rows, cols] with
// hi + mid + lo == fl(row_scale * W) exactly.
static __global__ void k_split_bf16x3(const float* __restrict__ W, long ldw, long rows, int cols,
                                      const float* __restrict__ row_scale, yl_bf16_t* __restrict__ hi,
                                      yl_bf16_t* __restrict__ mid, yl_bf16_t* __restrict__ lo) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= rows * cols) return;
  const long r = i / cols;
  float x = W[r * ldw + (i % cols)];
  if (row_scale != nullptr) x *= row_scale[r];
  const unsigned u = __float_as_uint(x);
  const float h = __uint_as_float(u & 0xffff0000u), hm = __uint_as_float(u & 0xffffff00u);
  const float m = hm - h, l = x - hm;
  hi[i] = (yl_bf16_t)(u >> 16);
  mid[i] = (yl_bf16_t)(__float_as_uint(m) >> 16);
  lo[i] = (yl_bf16_t)(__float_as_uint(l) >> 16);
}

extern "C" int yolat_split_bf16x3(const float* W, int64_t ldw, int64_t rows, int64_t cols, const float* row_scale,
                                  uint16_t* hi, uint16_t* mid, uint16_t* lo, yolat_stream_t stream) {
  if (rows <= 0 || cols <= 0 || !W || !hi || !mid || !lo || ldw < cols) return YOLAT_E_INVALID;
  hipLaunchKernelGGL(k_split_bf16x3, dim3(yl_cdiv(rows * cols, 256)), dim3(256), 0, (hipStream_t)stream, W, (long)ldw,
                     (long)rows, (int)cols, row_scale, hi, mid, lo);
  YL_LAUNCH_CHECK();
  return 0;
}

// Work split of the big problem (round 4).  One launch runs in rounds of `slots` resident workgroups; a workgroup costs one
// prologue (load + three-way split of its rows of A, measured ~ one column tile's time) plus its column tiles.  Rounds 2-3
// picked ONE power-of-two column split for all row tiles, i.e. paid either the round quantisation of whole-row workgroups
// (782 row tiles on 256 slots: 4 rounds of 17 units for 3.05 rounds of work) or a prologue per few column tiles (4 column
// tiles per workgroup at cfg 5: 5 units for 4 of work).  Now the first `full` row tiles — whole rounds — go to one workgroup
// each, and only the remainder is split into column groups: the plan with the smallest makespan of a list schedule in
// dispatch order (small problem's workgroups, cost 2; whole-row workgroups, cost 1 + tn; tail workgroups, cost 1 + ng).
namespace {
struct FxPlan { int full, groups, ng; };
double fx_makespan(int slots, long n_small, long n_full, int c_full, long n_tail, int c_tail) {
  // list schedule in dispatch order on identical slots (the next job goes to the slot that frees first), simulated on GROUPS
  // of slots with equal finish time instead of job by job: the jobs of a class are identical, so the k = min(jobs left,
  // group size) jobs that go to the earliest group move k of its slots to (time + cost) in one step.  A step per
  // (class, round) instead of a heap operation per job: the plan of a 200 k-row launch costs microseconds, not 0.6 ms
  // (every new (N, P) of a data loader is a cache miss in fx_plan).
  std::map<double, long> t;
  t[0.0] = slots;
  double last = 0.0;
  auto run = [&](long n, double c) {
    while (n > 0) {
      auto it = t.begin();
      const double at = it->first;
      const long k = n < it->second ? n : it->second;
      it->second -= k;
      if (it->second == 0) t.erase(it);
      t[at + c] += k;
      if (at + c > last) last = at + c;
      n -= k;
    }
  };
  run(n_small, 2.0);
  run(n_full, (double)c_full);
  run(n_tail, (double)c_tail);
  return last;
}
FxPlan fx_plan(int tm, int tn, long n_small, int slots) {
  struct Key { int tm, tn, slots; long n_small; FxPlan plan; };
  static thread_local Key cache[8];
  static thread_local int used = 0, next = 0;
  for (int i = 0; i < used; ++i)
    if (cache[i].tm == tm && cache[i].tn == tn && cache[i].slots == slots && cache[i].n_small == n_small) return cache[i].plan;
  FxPlan best{0, 1, tn};
  double best_t = 1e300;
  // whole rounds only (slots is a multiple of 8); candidates: none, all whole rounds, one round fewer — the schedule is
  // simulated job by job, and very large inputs would otherwise try hundreds of values
  const int f1 = (tm / slots) * slots;
  const int cand[3] = {0, f1, f1 >= slots ? f1 - slots : 0};
  for (int ci = 0; ci < 3; ++ci) {
    const int full = cand[ci];
    if (ci > 0 && full == cand[ci - 1]) continue;
    for (int g = 1; g <= tn; g *= 2) {
      const int ng = yl_cdiv(tn, g), groups = yl_cdiv(tn, ng);
      if (full == tm && g > 1) break;
      const double t = fx_makespan(slots, n_small, full, 1 + tn, (long)(tm - full) * groups, 1 + ng);
      if (t < best_t - 1e-9) { best_t = t; best = FxPlan{full, groups, ng}; }
    }
  }
  cache[next] = Key{tm, tn, slots, n_small, best};
  next = (next + 1) & 7;
  if (used < 8) ++used;
  return best;
}
}  // namespace

// pool[p, 0:F] = max over the rows of proposal p of relu(A . (sf (.) Wf)^T + tfold),  Ys = relu(S . (sfs (.) Wfs)^T
// + tsfold): both fusion blocks with pre-split weights (yolat_split_bf16x3 of the scaled rows) and folded shifts
// (s*b + t).  `pool` must be zero-filled first (yolat_pool_prepare).  D in {64, 128}, F % 64 == 0.
int yl_fusion_pair_eval_x6_impl(const float* A, int64_t lda, int64_t N, int64_t D, const uint16_t* Wh, const uint16_t* Wm,
                                const uint16_t* Wl, const float* tfold, int64_t F, const int32_t* node_seg, float* pool,
                                int64_t ldpool, const float* S, int64_t lds, int64_t P, const uint16_t* Wsh,
                                const uint16_t* Wsm, const uint16_t* Wsl, const float* tsfold, float* Ys, int64_t ldys,
                                const PoolRider* rider, yolat_stream_t stream) {
  if (N <= 0 || P <= 0 || F <= 0 || !A || !Wh || !Wm || !Wl || !tfold || !node_seg || !pool || !S || !Wsh || !Wsm || !Wsl ||
      !tsfold || !Ys)
    return YOLAT_E_INVALID;
  if (N >= (1LL << 31) - 256 || lda < D || lds < D || ldpool < F || ldys < F) return YOLAT_E_INVALID;
  if ((D != 64 && D != 128) || F % 64 != 0 || lda % 4 != 0 || lds % 4 != 0 || !yl_aligned16(A) || !yl_aligned16(S) ||
      !yl_aligned16(Wh) || !yl_aligned16(Wm) || !yl_aligned16(Wl) || !yl_aligned16(Wsh) || !yl_aligned16(Wsm) ||
      !yl_aligned16(Wsl) || (int64_t)P * ldpool >= (1LL << 32))
    return YOLAT_E_UNSUPPORTED;
  const int tn = (int)(F / 64);
  // Column groups.  One workgroup per CU is resident (8 waves x ~240 VGPRs), so the launch runs in rounds of 256
  // workgroups; a workgroup costs one prologue (load + split of its 256 rows of A, measured ~ one column tile's time)
  // plus its column tiles.  Pick the power-of-two split with the fewest (rounds x workgroup cost); the small problem
  // always gets the finest split (its few row tiles must not become the longest workgroups).
  FxProb p0, p1;
  p0.A = A; p0.lda = lda; p0.N = (int)N; p0.Wh = Wh; p0.Wm = Wm; p0.Wl = Wl; p0.tfold = tfold; p0.seg = node_seg;
  p0.out = pool; p0.ldo = ldpool; p0.out2 = nullptr; p0.ldo2 = 0; p0.ct2 = 1 << 30; p0.F = (int)F; p0.relu = 1; p0.key64 = nullptr; p0.sgn = nullptr; p0.a_scale = nullptr; p0.a_shift = nullptr; p0.a_floor = 0.f; p0.stats = nullptr;
  p1.A = S; p1.lda = lds; p1.N = (int)P; p1.Wh = Wsh; p1.Wm = Wsm; p1.Wl = Wsl; p1.tfold = tsfold; p1.seg = nullptr;
  p1.out = Ys; p1.ldo = ldys; p1.out2 = nullptr; p1.ldo2 = 0; p1.ct2 = 1 << 30; p1.F = (int)F; p1.relu = 1; p1.key64 = nullptr; p1.sgn = nullptr; p1.a_scale = nullptr; p1.a_shift = nullptr; p1.a_floor = 0.f; p1.stats = nullptr;
  const int rows_wg = D == 64 ? FxShape<64>::ROWS : FxShape<128>::ROWS, wg_round = D == 64 ? 512 : 256;
  p0.tm = yl_cdiv(N, rows_wg);
  p1.tm = yl_cdiv(P, rows_wg);
  p1.groups = tn; p1.ng = 1;
  const long n1 = (((long)p1.tm * tn + 7) & ~7L);
  const FxPlan plan = fx_plan(p0.tm, tn, n1, wg_round);
  p0.full = plan.full; p0.groups = plan.groups; p0.ng = plan.ng;
  PoolRider pr{};
  if (rider && rider->blocks > 0) pr = *rider;
  const long total = (long)p0.full + (long)(p0.tm - p0.full) * p0.groups + n1 + pr.blocks;
  if (total >= (1LL << 31)) return YOLAT_E_UNSUPPORTED;
  hipStream_t st = (hipStream_t)stream;
  if (pr.blocks > 0) {
    if (D == 128) hipLaunchKernelGGL((k_fusion_rows_x6<128, true>), dim3((unsigned)total), dim3(FxShape<128>::T), 0, st, p0, p1, pr);
    else hipLaunchKernelGGL((k_fusion_rows_x6<64, true>), dim3((unsigned)total), dim3(FxShape<64>::T), 0, st, p0, p1, pr);
  } else {
    if (D == 128) hipLaunchKernelGGL((k_fusion_rows_x6<128>), dim3((unsigned)total), dim3(FxShape<128>::T), 0, st, p0, p1, pr);
    else hipLaunchKernelGGL((k_fusion_rows_x6<64>), dim3((unsigned)total), dim3(FxShape<64>::T), 0, st, p0, p1, pr);
  }
  YL_LAUNCH_CHECK();
  return 0;
}

extern "C" int yolat_fusion_pair_eval_x6(const float* A, int64_t lda, int64_t N, int64_t D, const uint16_t* Wh,
                                         const uint16_t* Wm, const uint16_t* Wl, const float* tfold, int64_t F,
                                         const int32_t* node_seg, float* pool, int64_t ldpool, const float* S,
                                         int64_t lds, int64_t P, const uint16_t* Wsh, const uint16_t* Wsm,
                                         const uint16_t* Wsl, const float* tsfold, float* Ys, int64_t ldys,
                                         yolat_stream_t stream) {
  return yl_fusion_pair_eval_x6_impl(A, lda, N, D, Wh, Wm, Wl, tfold, F, node_seg, pool, ldpool, S, lds, P, Wsh, Wsm, Wsl,
                                     tsfold, Ys, ldys, nullptr, stream);
}

// Training-mode fusion GEMM (fusion_train.hip step 4) on the rows kernel: z = A . W^T + bias never stored, per
// (proposal, column) extreme-of-z keys.  W [F, K] fp32 is split here (it changes every step) into `wsplit`
// (3 * F * K bfloat16, 16-byte aligned).  K in {64, 128}, F % 64 == 0; returns YOLAT_E_UNSUPPORTED otherwise.
int yl_fusion_rows_x6_key64(const float* A, long lda, long N, long K, const float* W, const float* bias, long F,
                            const float* sgn, const int* node_seg, unsigned long long* keys, uint16_t* wsplit,
                            yolat_stream_t stream) {
  if ((K != 64 && K != 128) || F % 64 != 0 || lda % 4 != 0 || !yl_aligned16(A) || !yl_aligned16(wsplit) || !bias)
    return YOLAT_E_UNSUPPORTED;
  if (N >= (1LL << 31) - 256) return YOLAT_E_UNSUPPORTED;
  uint16_t *wh = wsplit, *wm = wsplit + F * K, *wl = wsplit + 2 * F * K;
  const int rc = yolat_split_bf16x3(W, K, F, K, nullptr, wh, wm, wl, stream);
  if (rc != 0) return rc;
  const int tn = (int)(F / 64);
  FxProb p0, p1;
  p0.A = A; p0.lda = lda; p0.N = (int)N; p0.Wh = wh; p0.Wm = wm; p0.Wl = wl; p0.tfold = bias; p0.seg = node_seg;
  p0.out = nullptr; p0.ldo = F; p0.out2 = nullptr; p0.ldo2 = 0; p0.ct2 = 1 << 30; p0.F = (int)F; p0.relu = 0;
  p0.key64 = keys; p0.sgn = sgn;
  p0.a_scale = nullptr; p0.a_shift = nullptr; p0.a_floor = 0.f; p0.stats = nullptr;
  const int rows_wg = K == 64 ? FxShape<64>::ROWS : FxShape<128>::ROWS, wg_round = K == 64 ? 512 : 256;
  p0.tm = yl_cdiv(N, rows_wg);
  const FxPlan plan = fx_plan(p0.tm, tn, 0, wg_round);
  p0.full = plan.full; p0.groups = plan.groups; p0.ng = plan.ng;
  p1 = p0;
  p1.tm = 0; p1.groups = 1; p1.ng = 1; p1.full = 0;
  const long total = (long)p0.full + (long)(p0.tm - p0.full) * p0.groups;
  if (total >= (1LL << 31)) return YOLAT_E_UNSUPPORTED;
  hipStream_t st = (hipStream_t)stream;
  if (K == 128) hipLaunchKernelGGL(k_fusion_rows_x6<128>, dim3((unsigned)total), dim3(FxShape<128>::T), 0, st, p0, p1, PoolRider{});
  else hipLaunchKernelGGL(k_fusion_rows_x6<64>, dim3((unsigned)total), dim3(FxShape<64>::T), 0, st, p0, p1, PoolRider{});
  YL_LAUNCH_CHECK();
  return 0;
}

// Training-mode Linear on the rows kernel:  Y [M, Nout] = pro(A) [M, K] . W^T + bias  (pre-activation), with the optional
// BatchNorm+ReLU prologue on A and the optional BatchNorm partial statistics of Y (yolat_linear_fwd's `stats`).  W [Nout,
// K] fp32 (contiguous rows, ldw) is split here — it changes every step — into `wsplit` (3 * Nout * K bfloat16, 16-byte
// aligned).  K in {64, 128}, Nout % 64 == 0, lda % 4 == 0.  For many rows (one workgroup per 256 rows): the second edge
// Linear of a training conv layer, [E, 64] -> [E, 64] (torch_vertex.py:311,335 nn.3).
extern "C" int yolat_linear_fwd_rows_x6(const float* A, int64_t lda, int64_t M, int64_t K, const float* a_scale,
                                        const float* a_shift, int a_relu, const float* W, int64_t ldw, const float* bias,
                                        int64_t Nout, float* Y, int64_t ldy, float* stats, uint16_t* wsplit,
                                        yolat_stream_t stream) {
  if (M <= 0 || !A || !W || !bias || !Y || !wsplit || lda < K || ldw < K || ldy < Nout) return YOLAT_E_INVALID;
  if ((a_scale == nullptr) != (a_shift == nullptr) || (a_relu && !a_scale)) return YOLAT_E_INVALID;
  if ((K != 64 && K != 128) || Nout <= 0 || Nout % 64 != 0 || lda % 4 != 0 || !yl_aligned16(A) || !yl_aligned16(wsplit) ||
      M >= (1LL << 31) - 256)
    return YOLAT_E_UNSUPPORTED;
  uint16_t *wh = wsplit, *wm = wsplit + Nout * K, *wl = wsplit + 2 * Nout * K;
  const int rc = yolat_split_bf16x3(W, ldw, Nout, K, nullptr, wh, wm, wl, stream);
  if (rc != 0) return rc;
  const int tn = (int)(Nout / 64);                     // the bias is the accumulators' initial value
  FxProb p0, p1;
  p0.A = A; p0.lda = lda; p0.N = (int)M; p0.Wh = wh; p0.Wm = wm; p0.Wl = wl; p0.tfold = bias; p0.seg = nullptr;
  p0.out = Y; p0.ldo = ldy; p0.out2 = nullptr; p0.ldo2 = 0; p0.ct2 = 1 << 30; p0.F = (int)Nout; p0.relu = 0;
  p0.key64 = nullptr; p0.sgn = nullptr;
  p0.a_scale = a_scale; p0.a_shift = a_shift; p0.a_floor = a_relu ? 0.f : -INFINITY; p0.stats = stats;
  p0.tm = yl_cdiv(M, K == 64 ? FxShape<64>::ROWS : FxShape<128>::ROWS); p0.groups = 1; p0.ng = tn;
  p1 = p0;
  p1.tm = 0; p1.groups = 1; p1.ng = 1;
  const long total = (long)p0.tm;
  if (total >= (1LL << 31)) return YOLAT_E_UNSUPPORTED;
  hipStream_t st = (hipStream_t)stream;
  if (K == 128) hipLaunchKernelGGL(k_fusion_rows_x6<128>, dim3((unsigned)total), dim3(FxShape<128>::T), 0, st, p0, p1, PoolRider{});
  else hipLaunchKernelGGL(k_fusion_rows_x6<64>, dim3((unsigned)total), dim3(FxShape<64>::T), 0, st, p0, p1, PoolRider{});
  YL_LAUNCH_CHECK();
  return 0;
}

// Node side of a factorised conv layer (eval, C = Cin = 64) on the same rows kernel: UV = f_in . Wuv'^T + uv_bias
// (two column tiles -> UV [N, 128]) and root = f_in . Wr^T + br (third column tile -> f_out) as one problem with the
// stacked, pre-split weights Wfr [192, 64] and shifts tfr [192]; the node branch relu(s_in . (sn (.) Wn)^T + tn') as the
// second problem.  Replaces yolat_node_uv_eval (three fp32-MFMA GEMMs in one launch) — the same launch count, the
// products on the bf16 matrix cores, A read once per 256 rows.
extern "C" int yolat_node_uv_eval_x6(const float* f_in, int64_t ld_f, const float* s_in, int64_t ld_s, int64_t N,
                                     const uint16_t* Wfr_h, const uint16_t* Wfr_m, const uint16_t* Wfr_l, const float* tfr,
                                     const uint16_t* Wn_h, const uint16_t* Wn_m, const uint16_t* Wn_l, const float* tn_fold,
                                     float* UV, int64_t ld_uv, float* f_out, int64_t ld_fo, float* s_out, int64_t ld_so,
                                     yolat_stream_t stream) {
  if (N <= 0 || N >= (1LL << 31) - 256 || !f_in || !s_in || !Wfr_h || !Wfr_m || !Wfr_l || !tfr || !Wn_h || !Wn_m || !Wn_l ||
      !tn_fold || !UV || !f_out || !s_out)
    return YOLAT_E_INVALID;
  if (ld_f < 64 || ld_s < 64 || ld_uv < 128 || ld_fo < 64 || ld_so < 64) return YOLAT_E_INVALID;
  if (ld_f % 4 != 0 || ld_s % 4 != 0 || !yl_aligned16(f_in) || !yl_aligned16(s_in) || !yl_aligned16(Wfr_h) ||
      !yl_aligned16(Wfr_m) || !yl_aligned16(Wfr_l) || !yl_aligned16(Wn_h) || !yl_aligned16(Wn_m) || !yl_aligned16(Wn_l))
    return YOLAT_E_UNSUPPORTED;
  FxProb p0, p1;
  p0.A = f_in; p0.lda = ld_f; p0.N = (int)N; p0.Wh = Wfr_h; p0.Wm = Wfr_m; p0.Wl = Wfr_l; p0.tfold = tfr; p0.seg = nullptr;
  p0.out = UV; p0.ldo = ld_uv; p0.out2 = f_out; p0.ldo2 = ld_fo; p0.ct2 = 2; p0.F = 192; p0.relu = 0;
  p0.key64 = nullptr; p0.sgn = nullptr; p1.key64 = nullptr; p1.sgn = nullptr;
  p0.a_scale = p0.a_shift = p1.a_scale = p1.a_shift = nullptr; p0.a_floor = p1.a_floor = 0.f; p0.stats = p1.stats = nullptr; p1.a_scale = nullptr; p1.a_shift = nullptr; p1.a_floor = 0.f; p1.stats = nullptr;
  p0.tm = yl_cdiv(N, FxShape<64>::ROWS); p0.groups = 1; p0.ng = 3;
  p1.A = s_in; p1.lda = ld_s; p1.N = (int)N; p1.Wh = Wn_h; p1.Wm = Wn_m; p1.Wl = Wn_l; p1.tfold = tn_fold; p1.seg = nullptr;
  p1.out = s_out; p1.ldo = ld_so; p1.out2 = nullptr; p1.ldo2 = 0; p1.ct2 = 1 << 30; p1.F = 64; p1.relu = 1;
  p1.tm = yl_cdiv(N, FxShape<64>::ROWS); p1.groups = 1; p1.ng = 1;
  const long total = (long)p0.tm * p0.groups + (((long)p1.tm * p1.groups + 7) & ~7L);
  if (total >= (1LL << 31)) return YOLAT_E_UNSUPPORTED;
  hipLaunchKernelGGL(k_fusion_rows_x6<64>, dim3((unsigned)total), dim3(FxShape<64>::T), 0, (hipStream_t)stream, p0, p1, PoolRider{});
  YL_LAUNCH_CHECK();
  return 0;
}

// ------------------------------------------------------------------------------------------------------------------
// Skinny Linear (+ folded BatchNorm + ReLU) with long K as a bf16x6-emulated fp32 GEMM — the per-proposal classifier
// layers (architecture3cc_rpn_gp_iter2.py:91-93,127-128: P x 2304 -> 512 -> 256 -> n_classes).  Few rows, long K:
// one 512-thread workgroup owns a 32 x 32 output tile and its 8 waves split K (wave w takes the 16-wide k steps w,
// w+8, ...), everything in registers: the wave's A values are loaded as fp32 and split on the fly (VALU, beside the
// matrix pipe); the weights are pre-split AND pre-packed in MFMA operand order (yolat_split_bf16x3_packed: one k
// step of one 32-column tile = 3 x 1 KB, each a fully coalesced wave load); four k steps in flight per wave.
// Column tiles are dealt to the XCDs (blockIdx % 8), so each XCD's L2 holds only its own eighth of the weights.
// The partial accumulators are summed through LDS by wave 0 in a fixed order (no atomics: deterministic), which
// applies shift (+ ReLU) and stores.  vs k_gemm_nt_sk (fp32 MFMA, 64 cycles per 2 k): 6 x 32 cycles per 16 k.
// ------------------------------------------------------------------------------------------------------------------
#define FX_SK_WAVES 8
// APRE: A is given pre-split and packed like the weights (yolat_split_bf16x3_packed of its rows) — no VALU in the loop;
// otherwise fp32 A [M, lda], split on the fly (36+ VALU per k step: the VALU then bounds the kernel, fine for short K).
template <bool APRE>
__global__ void __launch_bounds__(64 * FX_SK_WAVES) k_linear_x6_sk(const void* __restrict__ Av, long lda, int M, int K,
                                                                   const yl_bf16_t* __restrict__ Wp,
                                                                   const float* __restrict__ shift, int relu, int N,
                                                                   float* __restrict__ out, long ldo, int tm, int tn) {
  constexpr int NW = FX_SK_WAVES, DEPTH = 4;                  // k steps in flight per wave (the loads are latency bound)
  __shared__ float red[NW - 1][16][64];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, l31 = lane & 31, lhi = lane >> 5;
  // column tile ct = xcd + 8 * (slot / tm), row tile = slot % tm
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
  const int ct = xcd + 8 * (slot / tm), rt = slot % tm;
  if (ct >= tn) return;
  const int row0 = rt * 32, col0 = ct * 32;
  const int nks = K >> 4;
  const float* ap = APRE ? nullptr : reinterpret_cast<const float*>(Av) + (long)yl_min(row0 + l31, M - 1) * lda + 8 * lhi;
  const yl_bf16_t* app = APRE ? reinterpret_cast<const yl_bf16_t*>(Av) + (long)rt * nks * (3 * 512) + lane * 8 : nullptr;
  const yl_bf16_t* wp = Wp + (long)ct * nks * (3 * 512) + lane * 8;
  // two accumulators: the six products of a k step alternate between them (no MFMA waits on the one before it)
  f32x16 acc, acc2;
#pragma unroll
  for (int r = 0; r < 16; ++r) { acc[r] = 0.f; acc2[r] = 0.f; }
  struct Stage { float4 a0, a1; fx_bf16x8 a[3]; fx_bf16x8 w[3]; };
  // unconditional, clamped loads (a wave past its last k step re-reads that step; its products are zeroed): the
  // number of loads in flight is static
  auto load = [&](int ks, Stage& s) {
    const int kc = yl_min(ks, nks - 1);
    if (APRE) {
#pragma unroll
      for (int q = 0; q < 3; ++q) s.a[q] = *reinterpret_cast<const fx_bf16x8*>(app + (long)kc * (3 * 512) + q * 512);
    } else {
      s.a0 = *reinterpret_cast<const float4*>(ap + 16 * kc);
      s.a1 = *reinterpret_cast<const float4*>(ap + 16 * kc + 4);
    }
#pragma unroll
    for (int q = 0; q < 3; ++q) s.w[q] = *reinterpret_cast<const fx_bf16x8*>(wp + (long)kc * (3 * 512) + q * 512);
  };
  // branch-free: a step past the end contributes zeros (a branch here would make the compiler's vmcnt bookkeeping
  // give up and wait for ALL loads at the loop top)
  auto compute = [&](const Stage& s, bool valid) {
    fx_bf16x8 ah, am, al;
    if (APRE) {
      fx_u32x4 z = {0u, 0u, 0u, 0u};
      const fx_bf16x8 zero = *reinterpret_cast<fx_bf16x8*>(&z);
      ah = valid ? s.a[0] : zero; am = valid ? s.a[1] : zero; al = valid ? s.a[2] : zero;
    } else {
      const float g = valid ? 1.f : 0.f;
      const float x[8] = {s.a0.x * g, s.a0.y * g, s.a0.z * g, s.a0.w * g, s.a1.x * g, s.a1.y * g, s.a1.z * g, s.a1.w * g};
      fx_split8(x, ah, am, al);
    }
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, s.w[0], acc, 0, 0, 0);      // small terms first
    acc2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, s.w[2], acc2, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, s.w[1], acc, 0, 0, 0);
    acc2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, s.w[0], acc2, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, s.w[1], acc, 0, 0, 0);
    acc2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, s.w[0], acc2, 0, 0, 0);
  };
  Stage st[DEPTH];
#pragma unroll
  for (int d = 0; d < DEPTH; ++d) load(wave + NW * d, st[d]);
  for (int ks = wave; ks < nks; ks += NW * DEPTH) {
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) {
      compute(st[d], ks + NW * d < nks);
      __builtin_amdgcn_sched_barrier(0);               // keep the refill HERE (the scheduler would sink it to its use)
      load(ks + NW * (d + DEPTH), st[d]);
      __builtin_amdgcn_sched_barrier(0);
    }
  }
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] += acc2[r];
  if (wave > 0) {
#pragma unroll
    for (int r = 0; r < 16; ++r) red[wave - 1][r][lane] = acc[r];
  }
  __syncthreads();
  if (wave == 0) {
    const int col = col0 + l31;
    const float sh = (shift && col < N) ? shift[col] : 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      float v = acc[r];
#pragma unroll
      for (int w = 0; w < NW - 1; ++w) v += red[w][r][lane];               // fixed order
      v += sh;
      if (relu) v = fmaxf(v, 0.f);
      const int row = row0 + (r & 3) + 8 * (r >> 2) + 4 * lhi;
      if (row < M && col < N) out[(long)row * ldo + col] = v;
    }
  }
}

// W [rows = N, cols = K] fp32 (optionally scaled per row) -> its exact 3-way bfloat16 split in the operand order of
// k_linear_x6_sk: packed[ct][ks][part][lane][8] with ct = n / 32, ks = k / 16, lane = n % 32 + 32 * ((k % 16) / 8),
// element k % 8; rows beyond N are zero.  One thread per 8 consecutive k.
static __global__ void k_split_bf16x3_packed(const float* __restrict__ W, long ldw, int N, int K,
                                             const float* __restrict__ row_scale, yl_bf16_t* __restrict__ packed) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;         // (ct, ks, lane)
  const int nks = K >> 4, tn = (N + 31) >> 5;
  if (i >= (long)tn * nks * 64) return;
  const int lane = (int)(i & 63), ks = (int)((i >> 6) % nks), ct = (int)((i >> 6) / nks);
  const int n = ct * 32 + (lane & 31), k0 = ks * 16 + 8 * (lane >> 5);
  float x[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) x[e] = n < N ? W[(long)n * ldw + k0 + e] * (row_scale ? row_scale[n] : 1.f) : 0.f;
  fx_bf16x8 h, m, l;
  fx_split8(x, h, m, l);
  yl_bf16_t* o = packed + (((long)ct * nks + ks) * 3) * 512 + lane * 8;
  *reinterpret_cast<fx_bf16x8*>(o) = h;
  *reinterpret_cast<fx_bf16x8*>(o + 512) = m;
  *reinterpret_cast<fx_bf16x8*>(o + 1024) = l;
}

extern "C" size_t yolat_split_bf16x3_packed_elems(int64_t N, int64_t K) {
  return (N <= 0 || K <= 0) ? 0 : (size_t)yl_cdiv(N, 32) * (size_t)(K / 16) * 3 * 512;
}
// packed: yolat_split_bf16x3_packed_elems(N, K) bfloat16 values, 16-byte aligned.  K % 16 == 0.
extern "C" int yolat_split_bf16x3_packed(const float* W, int64_t ldw, int64_t N, int64_t K, const float* row_scale,
                                         uint16_t* packed, yolat_stream_t stream) {
  if (N <= 0 || K <= 0 || !W || !packed || ldw < K || N >= (1LL << 31) - 32 || K >= (1LL << 31)) return YOLAT_E_INVALID;
  if (K % 16 != 0 || !yl_aligned16(packed)) return YOLAT_E_UNSUPPORTED;
  const long items = (long)yl_cdiv(N, 32) * (K / 16) * 64;
  hipLaunchKernelGGL(k_split_bf16x3_packed, dim3((unsigned)yl_cdiv(items, 256)), dim3(256), 0, (hipStream_t)stream, W,
                     (long)ldw, (int)N, (int)K, row_scale, reinterpret_cast<yl_bf16_t*>(packed));
  YL_LAUNCH_CHECK();
  return 0;
}

// out [M, N] = act(A [M, K] . (s (.) W)^T + shift),  W given pre-split and packed (yolat_split_bf16x3_packed of the
// scaled rows), shift = s*b + t folded (NULL: none), act = ReLU when relu != 0.  K % 16 == 0, lda % 4 == 0, A and Wp
// 16-byte aligned.  Meant for few rows (the launch has cdiv(M,32) * cdiv(N,32) workgroups and re-reads the weights
// once per 32 rows).  yolat_linear_x6_pre: A given as yolat_split_bf16x3_packed(A, lda, M, K, NULL) — the choice for
// long K (on-the-fly splitting of A costs more VALU time than the products cost matrix-pipe time).
static int linear_x6_launch(const void* A, bool pre, int64_t lda, int64_t M, int64_t K, const uint16_t* Wp,
                            const float* shift, int relu, int64_t N, float* out, int64_t ldo, yolat_stream_t stream) {
  if (M <= 0 || N <= 0 || K <= 0 || !A || !Wp || !out) return YOLAT_E_INVALID;
  if ((!pre && lda < K) || ldo < N || M >= (1LL << 31) - 32 || N >= (1LL << 31) - 64) return YOLAT_E_INVALID;
  if (K % 16 != 0 || (!pre && lda % 4 != 0) || !yl_aligned16(A) || !yl_aligned16(Wp)) return YOLAT_E_UNSUPPORTED;
  const long tm = yl_cdiv(M, 32), tn = yl_cdiv(N, 32);
  const long total = 8 * yl_cdiv(tn, 8) * tm;
  if (total >= (1LL << 31)) return YOLAT_E_UNSUPPORTED;
  const dim3 grid((unsigned)total), block(64 * FX_SK_WAVES);
  const yl_bf16_t* wp = reinterpret_cast<const yl_bf16_t*>(Wp);
  if (pre)
    hipLaunchKernelGGL(k_linear_x6_sk<true>, grid, block, 0, (hipStream_t)stream, A, (long)lda, (int)M, (int)K, wp, shift,
                       relu, (int)N, out, (long)ldo, (int)tm, (int)tn);
  else
    hipLaunchKernelGGL(k_linear_x6_sk<false>, grid, block, 0, (hipStream_t)stream, A, (long)lda, (int)M, (int)K, wp, shift,
                       relu, (int)N, out, (long)ldo, (int)tm, (int)tn);
  YL_LAUNCH_CHECK();
  return 0;
}
extern "C" int yolat_linear_x6(const float* A, int64_t lda, int64_t M, int64_t K, const uint16_t* Wp, const float* shift,
                               int relu, int64_t N, float* out, int64_t ldo, yolat_stream_t stream) {
  return linear_x6_launch(A, false, lda, M, K, Wp, shift, relu, N, out, ldo, stream);
}
extern "C" int yolat_linear_x6_pre(const uint16_t* Ap, int64_t M, int64_t K, const uint16_t* Wp, const float* shift,
                                   int relu, int64_t N, float* out, int64_t ldo, yolat_stream_t stream) {
  return linear_x6_launch(Ap, true, K, M, K, Wp, shift, relu, N, out, ldo, stream);
}
