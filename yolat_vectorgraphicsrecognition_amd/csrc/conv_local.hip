// conv_local.hip — ALL conv layers + the pooling prologue of the bf16-storage eval forward in ONE launch, for
// proposal-local batches (round 5).
//
// Reference: Backbone.forward, cad_recognition/architecture3cc_rpn_gp_iter2.py:44-69 (head conv, n_blocks - 1 ResBlocks,
// concat of the last n_blocks_out layers, segment mean of the node branch) + the segment max of :122 over the concat
// columns; the conv itself is AttrRelativeEdgeConvGlobalPool2.forward, gcn_lib/sparse/torch_vertex.py:319-337 (eval).
//
// The property used (Datasets/graph_dict3.py:582-600,733): an edge never leaves its proposal, and a
// proposal's nodes are a contiguous row range.  So a workgroup that owns WHOLE proposals owns every row its layers will
// ever gather, and the only HBM traffic of the conv stack is x, the CSR ids, e_attr in and feats + the pooled rows out
// (~135 MB at N = 200 k / E = 1.2 M instead of the ~1.8 GB the per-layer launches move: UV [N,128] + root [N,64] + f / s
// [N,64] written and re-gathered per layer).
//
// Structure.  A workgroup (256 threads = 4 waves, 65.5 KB of LDS: two per CU) is PERSISTENT over a contiguous group of
// proposals and packs them greedily into tiles of <= 64 nodes / <= 640 edges.  Per tile, everything lives in LDS:
//   UV [64][128] bf16 (per-node products U' | V' of the factorised first edge Linear), R [64][64] fp32 (root Linear),
//   f [64][64] bf16 (layer output = next layer's input), s [64][64] bf16 (node branch), packed edge ids, e_attr as bf16
//   (hi, lo) MFMA fragments, the tile's row_ptr.
// Per layer:
//   node phase   wave g computes output group g of OUT^T[256 ch][64 nodes] = W' . f^T (g: U | V | root | node branch) on
//                v_mfma_f32_32x32x16_bf16 — transposed, so a lane holds 4 CONSECUTIVE channels of one node and writes
//                8-byte (bf16) / 16-byte (fp32) LDS rows; BatchNorm scales are folded into the packed weights, shifts are
//                the accumulators' start values.  Layer 0 (K = in_channels <= 8 raw features): fp32 FMAs.
//   edge phase   the register-chained MFMA pipeline of edge_chain.hip (layer 1 transposed through identity fragments on
//                the gathered 16-byte row chunks, ReLU + bf16 in registers = layer 2's A operand), with the gathers served
//                by LDS.  A wave owns two edge streams = the in-edges of two runs of <= 16 consecutive nodes.  NEW: the
//                mean aggregation is one more MFMA — AGG^T[ch][node slot] += M^T[ch][edge] . S[edge][slot], S the 0/1
//                incidence of the step's 2 x 16 edges on the wave's 2 x 16 node slots, built per lane from its node's CSR
//                range (a bit mask) — so the per-node sums live in 32 accumulator registers for the whole phase: no
//                per-row scalar branches, no slot tables, no staging rows.  (Messages enter that MFMA rounded to bf16:
//                2^-9 per message before a mean that is itself stored as bf16.)  At the end of the phase a lane adds the
//                root row, scales by 1 / deg and writes its node's 16 + 16 channels to the f tile.
//   outputs      layers >= n_blocks - n_blocks_out: f tile -> feats[:, 64 j ..] (16-byte stores), per-proposal max of f and
//                mean of s -> Z (fp32), and once per tile Z[p, 0:F] = 0 (what k_pool_prepare_h did in a launch of its own).
// A batch that is not proposal-local, or a proposal that does not fit a tile, raises `flag`; the caller keeps the per-layer
// launches enqueued behind this one, gated on that word (bf16_eval.hip).
#include "common.hpp"
#include <stdlib.h>

typedef unsigned short u16;
typedef unsigned cl_u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned cl_u32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 cl_bf16x8 __attribute__((ext_vector_type(8)));
typedef short cl_s16x2 __attribute__((ext_vector_type(2)));

namespace {
// tuning / debug state of yolat_conv_local_tune (0 = automatic)
int g_cl_nw = 0, g_cl_g0 = 0, g_cl_abl = 0;
long long* g_cl_stamps = nullptr;

constexpr int CL_UVB = 272;              // UV tile row stride, bytes (128 bf16 + 16 B pad)
constexpr int CL_RB = 272;               // R tile row stride, bytes (64 fp32 + 16 B pad)
constexpr int CL_FB = 144;               // f / s tile row stride, bytes (64 bf16 + 16 B pad)
constexpr int CL_GMAX = 64;              // proposals per workgroup (group arrays in LDS)
constexpr int CL_EDGE_BYTES = 12 * 1024; // packed image, per layer: W2F[8] WCA[2] TB[2] fragments
constexpr int CL_NODE_BYTES = 32 * 1024; //   node weights: 4 groups x (2 x 4) fragments (layer 0: (W_hi | W_hi), (W_lo | 0), 0, 0)
constexpr int CL_SHIFT_BYTES = 1024;     //   shift [256] fp32
constexpr int CL_LAYER_BYTES = CL_EDGE_BYTES + CL_NODE_BYTES + CL_SHIFT_BYTES;

__device__ __forceinline__ cl_bf16x8 cl_frag(cl_u32x4 v) { return __builtin_bit_cast(cl_bf16x8, v); }
__device__ __forceinline__ cl_bf16x8 cl_frag(unsigned a, unsigned b, unsigned c, unsigned d) {
  cl_u32x4 v = {a, b, c, d};
  return __builtin_bit_cast(cl_bf16x8, v);
}
// relu on a packed bfloat16 pair (v_pk_max_i16: a negative bf16 is a negative int16)
__device__ __forceinline__ unsigned cl_relu_pk(unsigned p) {
  cl_s16x2 v = __builtin_bit_cast(cl_s16x2, p);
  const cl_s16x2 z = {0, 0};
  v = __builtin_elementwise_max(v, z);
  return __builtin_bit_cast(unsigned, v);
}
__device__ __forceinline__ f32x16 cl_mfma(const cl_bf16x8& a, const cl_bf16x8& b, const f32x16& c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}
// workgroup barrier that waits for this wave's LDS traffic only: global loads (the weight-fragment prefetches) and stores
// (feats, Z) stay in flight across it — nothing inside the kernel consumes another wave's global stores
__device__ __forceinline__ void cl_lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
__device__ __forceinline__ int cl_opaque(int v) { asm volatile("" : "+v"(v)); return v; }
// bits pos, pos + 1 of mask -> packed bf16 pair of 1.0 / 0.0
__device__ __forceinline__ unsigned cl_sel2(unsigned mask, int pos) {
  const unsigned t = (mask >> pos) & 3u;
  return ((t * 0x8001u) & 0x10001u) * 0x3F80u;
}

struct ClArgs {
  const float* x; int ldx; int cin0;
  const int* row_ptr; const int* src; const int* dst; const float* attr; const int* seg_ptr;
  // COO instantiation (the tile sorts its edges by destination in LDS): the raw edge list (int64 pairs, element strides
  // se / sc as in yolat_forward_eval), e_attr in COO order through `attr`, eptr [P + 1] = first edge of each proposal
  const long long* edge; long se, sc; const int* eptr;
  int* status;           // when the caller vouched for the batch: a violation found here is an input error (YOLAT_STATUS_NOT_LOCAL)
  int N, E, P, G0;
  int L, lo;
  const unsigned char* pack;
  u16* feats; int ld_feats;
  float* Z; int ldz; int F, D;
  int* flag; int flag_val;
  int abl;
  long long* stamps;     // debug: per-workgroup phase time stamps (tools/exp/conv_local_bench.py)
};

// quad reductions through DPP (lanes 4q .. 4q + 3 hold the four row parts of one item)
__device__ __forceinline__ float cl_quad_xor1(float v) {
  return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0xB1, 0xF, 0xF, true));
}
__device__ __forceinline__ float cl_quad_xor2(float v) {
  return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0x4E, 0xF, 0xF, true));
}

// NW waves per workgroup: tiles of T = 16 NW nodes / ET = 128 NW edges.  NW = 4: 256 threads, 77 KB of LDS, two workgroups
// per CU; NW = 8: 512 threads, 141 KB, one per CU (two waves per SIMD either way).
// COO: the tile's edges come in the caller's order (grouped by proposal: eptr) and are counting-sorted by destination in
// LDS, stable — the summation order of the destination-sorted form (graph.hip) — so the global COO -> CSR build
// (count, scan, fill, rank-and-emit: 4 launches over E) disappears from the forward.
template <int NW, bool COO>
__global__ void __launch_bounds__(64 * NW, 2) k_conv_local_h(const ClArgs a) {
  constexpr int T = 16 * NW, ET = 128 * NW, NT = 64 * NW, NS = 2 * NW;
  constexpr int LOGT = (NW == 8) ? 7 : 6;          // T = 1 << LOGT
  constexpr int NE = (768 + NT - 1) / NT;          // 16-byte pieces of the edge fragments per thread
  __shared__ __attribute__((aligned(16))) unsigned char uv_s[T * CL_UVB];
  __shared__ __attribute__((aligned(16))) unsigned char r_s[T * CL_RB];
  __shared__ __attribute__((aligned(16))) unsigned char f_s[T * CL_FB];      // also: the x tile [T][8] fp32
  __shared__ __attribute__((aligned(16))) unsigned char s_s[T * CL_FB];
  __shared__ __attribute__((aligned(16))) cl_u32x4 ab_s[ET];
  __shared__ unsigned idx_s[ET];
  __shared__ int rp_s[T + 4];
  __shared__ int gseg_s[CL_GMAX + 1], grow_s[CL_GMAX + 1];
  __shared__ __attribute__((aligned(16))) float shift_s[256];                  // the coming node phase's shifts
  __shared__ int stop_s;                 // the flag word as thread 0 saw it at this tile's start (workgroup-uniform exit)
  __shared__ __attribute__((aligned(16))) cl_u32x4 ef_s[12 * 64];              // the layer's edge-phase fragments
  // COO: in-degree of every tile node inside each 64-edge chunk of the tile's edge list (one byte each; all zero between
  // tiles: the scan below clears what it reads).  The chunk bases [2 NW][T] u16 live in the (still unused) UV tile.
  __shared__ __attribute__((aligned(16))) unsigned char cnt_s[COO ? 2 * NW * T : 16];

  const int tid0 = threadIdx.x;
  const int wv = __builtin_amdgcn_readfirstlane(tid0 >> 6);
  const int grp = wv & 3, rbp = wv >> 2;           // node phase: output group, 64-row half of the tile
  const int p_lo = blockIdx.x * a.G0;
  const int np = yl_min(a.G0, a.P - p_lo);
  if (np <= 0) return;
  for (int i = tid0; i <= np; i += NT) {
    const int s = a.seg_ptr[p_lo + i];
    gseg_s[i] = s;
    if constexpr (COO) grow_s[i] = yl_min(yl_max(a.eptr[p_lo + i], 0), a.E);
    else grow_s[i] = a.row_ptr[yl_min(yl_max(s, 0), a.N)];
  }
  if constexpr (COO)
    for (int i = tid0; i < 2 * NW * T / 4; i += NT) reinterpret_cast<unsigned*>(cnt_s)[i] = 0u;
  if (tid0 < 256) shift_s[tid0] = reinterpret_cast<const float*>(a.pack + CL_EDGE_BYTES + CL_NODE_BYTES)[tid0];
  // node-phase weight fragments of this wave's output group, one phase ahead.  Layer 0 uses [.][0..1] only: while no
  // fragments are in flight (last layer -> next tile) the other four hold the NEXT tile's raw loads (pre0..3 below)
  cl_u32x4 af[2][4];
  cl_u32x4 est[NE];         // this thread's 16-byte pieces of the coming layer's edge-phase fragments
#pragma unroll
  for (int i = 0; i < NE; ++i) est[i] = reinterpret_cast<const cl_u32x4*>(a.pack)[yl_min(tid0 + NT * i, 767)];
  __syncthreads();

  // identity fragments of the transposed first layer: A[m][k] = 1 iff k == m - 16 j  (edge_chain.hip)
  cl_bf16x8 Id[2];
  {
    const int lane = tid0 & 63, l31 = lane & 31, lhi = lane >> 5;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const bool mine = ((l31 >> 4) == j) && (((l31 >> 3) & 1) == lhi);
      const int i = l31 & 7;
      unsigned d[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) d[k] = (mine && (i >> 1) == k) ? (0x3F80u << (16 * (i & 1))) : 0u;
      Id[j] = cl_frag(d[0], d[1], d[2], d[3]);
    }
  }
  const cl_bf16x8 OnesA = (tid0 & 32) ? cl_frag(0u, 0u, 0u, 0u) : cl_frag(0x3F803F80u, 0x00003F80u, 0u, 0u);

  int stamp_k = 0;
#define CL_STAMP() do { if (a.stamps != nullptr && tid0 == 0 && stamp_k < 64) a.stamps[blockIdx.x * 64 + stamp_k++] = clock64(); } while (0)

  // greedy tile: the first proposals [q0, q1) from `from` on with <= T nodes and <= ET edges; a proposal that does not
  // fit (or an unsorted segment table) raises the flag (the gated per-layer path then runs) and is skipped
  // the batch does not have the property: tell the gated per-layer launches (or, when the caller vouched for it, the status word)
  auto raise = [&]() {
    *a.flag = a.flag_val;
    if (a.status != nullptr) atomicOr(a.status, YOLAT_STATUS_NOT_LOCAL);
  };
  auto next_tile = [&](int from, int& q0, int& q1) {
    q0 = from;
    q1 = from;
    while (q0 < np) {
      const int sg = gseg_s[q0], rg = grow_s[q0];
      q1 = q0;
      while (q1 < np && gseg_s[q1 + 1] - sg <= T && grow_s[q1 + 1] - rg <= ET && gseg_s[q1 + 1] >= gseg_s[q1] &&
             grow_s[q1 + 1] >= grow_s[q1]) ++q1;
      if (q1 > q0) return;
      if (tid0 == 0) raise();
      ++q0;
    }
  };
  // the raw loads of a tile: row_ptr | x (2 floats) | dst (2) | src (2) | e_attr (2 x 4)  — 15 registers
  auto issue_tile_loads = [&](int q0, int q1, cl_u32x4& r0, cl_u32x4& r1, cl_u32x4& r2, cl_u32x4& r3) {
    const int tid = cl_opaque(tid0);
    const int n0 = gseg_s[q0], nt = gseg_s[q1] - n0, e0 = grow_s[q0], et = grow_s[q1] - e0;
    if constexpr (!COO) r0.x = (unsigned)a.row_ptr[n0 + yl_min(tid, nt)];
    else r0.x = 0u;
    unsigned xv[2], dv[2], sv[2];
    cl_u32x4 av[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const int i = tid + NT * t, n = i >> 3, k = i & 7;
      xv[t] = __float_as_uint(a.x[(long)yl_min(n0 + n, a.N - 1) * a.ldx + yl_min(k, a.cin0 - 1)]);
      const int ec = yl_min(e0 + yl_min(i, et > 0 ? et - 1 : 0), a.E > 0 ? a.E - 1 : 0);
      if constexpr (COO) {          // (ids are < 2^31 wherever they are valid: the low halves decide)
        dv[t] = (unsigned)a.edge[(long)ec * a.se + a.sc];
        sv[t] = (unsigned)a.edge[(long)ec * a.se];
      } else {
        dv[t] = (unsigned)a.dst[ec];
        sv[t] = (unsigned)a.src[ec];
      }
      av[t] = *reinterpret_cast<const cl_u32x4*>(a.attr + 4l * ec);
    }
    r0.y = xv[0]; r0.z = xv[1]; r0.w = dv[0];
    r1.x = dv[1]; r1.y = sv[0]; r1.z = sv[1]; r1.w = 0u;
    r2 = av[0]; r3 = av[1];
  };

  // a proposal of the group that cannot fit a tile is known now: raise the flag before any work is done (the other
  // workgroups stop at their next check, below)
  for (int i = tid0; i < np; i += NT)
    if (gseg_s[i + 1] - gseg_s[i] > T || grow_s[i + 1] - grow_s[i] > ET || gseg_s[i + 1] < gseg_s[i] ||
        grow_s[i + 1] < grow_s[i]) raise();
  int p0, p1;
  next_tile(0, p0, p1);
  bool first_tile = true;
  cl_u32x4& pre0 = af[0][2];
  cl_u32x4& pre1 = af[0][3];
  cl_u32x4& pre2 = af[1][2];
  cl_u32x4& pre3 = af[1][3];
  if (p0 < np && a.E > 0) issue_tile_loads(p0, p1, pre0, pre1, pre2, pre3);
  while (p0 < np) {
    CL_STAMP();      // 0: tile start
    // (lane ids re-derived from an opaque copy per section: keeps the compiler from hoisting every lane-derived address
    // out of the tile loop and spilling it — a scratch reload waits on vmcnt, i.e. on every global load / store in flight)
    const int tid = cl_opaque(tid0), lane = tid & 63, l31 = lane & 31, lhi = lane >> 5;
    const int n0 = gseg_s[p0], nt = gseg_s[p1] - n0, e0 = grow_s[p0], et = grow_s[p1] - e0;
    const int npr = p1 - p0;
    int pn0, pn1;                      // the next tile (its loads are issued during this tile's last layer)
    next_tile(p1, pn0, pn1);

    // ---- tile -> LDS: row_ptr (tile-local), x rows, packed edge ids, e_attr fragments
    int coo_clear[2] = {-1, -1};       // COO: the cnt_s entries this lane wrote (cleared behind the barrier below)
    if (a.E == 0) {                    // (no edge arrays to read from)
      if (tid <= nt) rp_s[tid] = 0;
      float* xs = reinterpret_cast<float*>(f_s);
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        const int i = tid + NT * t, n = i >> 3, k = i & 7;
        const float v = a.x[(long)yl_min(n0 + n, a.N - 1) * a.ldx + yl_min(k, a.cin0 - 1)];
        xs[i] = (n < nt && k < a.cin0) ? v : 0.f;
      }
    } else if constexpr (COO) {
      // The tile's edges arrive in the caller's order.  Stable counting sort by destination, all in LDS:
      //   rank   a lane's rank among the lanes of its 64-edge chunk with the same destination: LOGT ballots, no loop over
      //          values; the first lane of each group stores the group's size -> cnt_s[chunk][node]
      //   scan   (every wave, two nodes per lane) running sum over the chunks per node -> chunk bases, in-degrees;
      //          exclusive DPP scan over the nodes -> rp_s (the tile's row_ptr); base_s[chunk][node] = row start + edges of
      //          earlier chunks, each wave for its own two chunks
      //   emit   edge -> slot base_s[chunk][dst] + rank: packed ids and the e_attr fragment
      float* xs = reinterpret_cast<float*>(f_s);
      unsigned short* base_s = reinterpret_cast<unsigned short*>(uv_s);
      const unsigned xv[2] = {pre0.y, pre0.z}, dv[2] = {pre0.w, pre1.x}, sv[2] = {pre1.y, pre1.z};
      bool bad = false;
      int dd[2], rk[2];
      bool okk[2];
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        const int i = tid + NT * t, n = i >> 3, k = i & 7;
        xs[i] = (n < nt && k < a.cin0) ? __uint_as_float(xv[t]) : 0.f;
        const int d = (int)dv[t] - n0, sr = (int)sv[t] - n0;
        const bool ok = i < et && (unsigned)sr < (unsigned)nt && (unsigned)d < (unsigned)nt;
        bad |= (i < et) && !ok;
        unsigned long long m = __ballot(ok);
#pragma unroll
        for (int b = 0; b < LOGT; ++b) {
          const bool bit = (d >> b) & 1;
          const unsigned long long bal = __ballot(bit);
          m &= bit ? bal : ~bal;
        }
        const int r = __popcll(m & ((1ull << lane) - 1ull));
        if (ok && r == 0) cnt_s[(wv + NW * t) * T + d] = (unsigned char)__popcll(m);
        dd[t] = d; rk[t] = r; okk[t] = ok;
      }
      if (bad) raise();
      cl_lds_barrier();
      // EVERY wave runs the chunk walk and the node scan (the same LDS reads, redundantly: no second workgroup barrier, no
      // wave that works while seven wait) and keeps the bases of its OWN two chunks; lane l holds nodes 2 l, 2 l + 1
      {
        const bool act = 2 * lane < T;
        unsigned run0 = 0, run1 = 0, mine[2] = {0u, 0u};
#pragma unroll
        for (int c = 0; c < 2 * NW; ++c) {
          const unsigned v = act ? *reinterpret_cast<const unsigned short*>(cnt_s + c * T + 2 * lane) : 0u;
          if (c == wv) mine[0] = run0 | (run1 << 16);
          if (c == wv + NW) mine[1] = run0 | (run1 << 16);
          run0 += v & 0xFFu;
          run1 += v >> 8;
        }
        const int tot = (int)(run0 + run1);
        // inclusive wave scan through DPP (row shifts, then the row totals broadcast down: no LDS round trips)
        int incl = tot;
        incl += __builtin_amdgcn_update_dpp(0, incl, 0x111, 0xF, 0xF, false);
        incl += __builtin_amdgcn_update_dpp(0, incl, 0x112, 0xF, 0xF, false);
        incl += __builtin_amdgcn_update_dpp(0, incl, 0x114, 0xF, 0xF, false);
        incl += __builtin_amdgcn_update_dpp(0, incl, 0x118, 0xF, 0xF, false);
        incl += __builtin_amdgcn_update_dpp(0, incl, 0x142, 0xA, 0xF, false);
        incl += __builtin_amdgcn_update_dpp(0, incl, 0x143, 0xC, 0xF, false);
        const unsigned rp0 = (unsigned)(incl - tot), rp1 = rp0 + run0;
        if (act) {
          if (wv == 0) {
            rp_s[2 * lane] = (int)rp0;
            rp_s[2 * lane + 1] = (int)rp1;
            if (2 * lane + 2 == T) rp_s[T] = incl;
          }
#pragma unroll
          for (int t = 0; t < 2; ++t)
            *reinterpret_cast<unsigned*>(base_s + (wv + NW * t) * T + 2 * lane) =
                ((mine[t] & 0xFFFFu) + rp0) | (((mine[t] >> 16) + rp1) << 16);
        }
      }
      // (the wave reads back rows of base_s it wrote itself: LDS operations of a wave complete in order)
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        const cl_u32x4 qa = t ? pre3 : pre2;
        const float4 q = __builtin_bit_cast(float4, qa);
        if (okk[t]) {
          const int pos = (int)base_s[(wv + NW * t) * T + dd[t]] + rk[t];
          const int sr = (int)sv[t] - n0;
          idx_s[pos] = (unsigned)dd[t] | ((unsigned)sr << 8);
          const unsigned h01 = yl_pack_bf16(q.x, q.y), h23 = yl_pack_bf16(q.z, q.w);
          const unsigned l01 = yl_pack_bf16(q.x - yl_bf16_lo(h01), q.y - yl_bf16_hi(h01));
          const unsigned l23 = yl_pack_bf16(q.z - yl_bf16_lo(h23), q.w - yl_bf16_hi(h23));
          const cl_u32x4 fr = {h01, h23, l01, l23};
          ab_s[pos] = fr;
        }
      }
      coo_clear[0] = okk[0] && rk[0] == 0 ? wv * T + dd[0] : -1;
      coo_clear[1] = okk[1] && rk[1] == 0 ? (wv + NW) * T + dd[1] : -1;
    } else {
      if (tid <= nt) rp_s[tid] = (int)pre0.x - e0;
      float* xs = reinterpret_cast<float*>(f_s);
      const unsigned xv[2] = {pre0.y, pre0.z}, dv[2] = {pre0.w, pre1.x}, sv[2] = {pre1.y, pre1.z};
      bool bad = false;
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        const int i = tid + NT * t, n = i >> 3, k = i & 7;
        xs[i] = (n < nt && k < a.cin0) ? __uint_as_float(xv[t]) : 0.f;
        const int d = (int)dv[t] - n0, sr = (int)sv[t] - n0;
        const cl_u32x4 qa = t ? pre3 : pre2;
        const float4 q = __builtin_bit_cast(float4, qa);
        if (i < et) {
          const bool ok = (unsigned)sr < (unsigned)nt && (unsigned)d < (unsigned)nt;
          bad |= !ok;
          idx_s[i] = ok ? ((unsigned)d | ((unsigned)sr << 8)) : 0u;
          const unsigned h01 = yl_pack_bf16(q.x, q.y), h23 = yl_pack_bf16(q.z, q.w);
          const unsigned l01 = yl_pack_bf16(q.x - yl_bf16_lo(h01), q.y - yl_bf16_hi(h01));
          const unsigned l23 = yl_pack_bf16(q.z - yl_bf16_lo(h23), q.w - yl_bf16_hi(h23));
          const cl_u32x4 fr = {h01, h23, l01, l23};
          ab_s[i] = fr;
        }
      }
      if (bad) raise();
    }
    {   // layer 0's node-phase fragments (two k-steps): in flight under the barrier and the stream set-up
      const cl_u32x4* ap = reinterpret_cast<const cl_u32x4*>(a.pack + CL_EDGE_BYTES) + (grp * 8) * 64 + lane;
#pragma unroll
      for (int nt_ = 0; nt_ < 2; ++nt_)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) af[nt_][ks] = ap[(nt_ * 4 + ks) * 64];
    }
    // another workgroup (or an earlier tile of this one) has found the batch unfit: the gated per-layer launches will redo
    // everything, so stop after at most one tile instead of finishing ~230 us of work that is thrown away.  (An L2-served
    // load: a plain one could be answered by this CU's L1 for ever.)
    if (tid == 0) stop_s = __hip_atomic_load(a.flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    CL_STAMP();      // 1: tile loads landed + LDS written
    cl_lds_barrier();
    if (stop_s == a.flag_val) return;
    if constexpr (COO) {               // every wave has finished its chunk walk: the counters go back to zero for the next tile
#pragma unroll
      for (int t = 0; t < 2; ++t)
        if (coo_clear[t] >= 0) cnt_s[coo_clear[t]] = 0;
    }

    // ---- the wave's two edge streams: node boundaries nb[0..NS] (edge-balanced, node-aligned, <= 16 nodes each)
    int ns[3], sb[3];
    {
      int cand = 0;
      {
        const int k = lane < NS ? lane : NS;
        if (et > 0) {
          const int tgt = (et * k) / NS;
          const int d = (int)(idx_s[yl_min(tgt, et - 1)] & 0xFFu);
          cand = (k >= NS) ? nt : ((rp_s[d] == tgt) ? d : d + 1);
        } else {
          cand = (nt * k) / NS;
        }
      }
      int nb = 0;
      ns[0] = ns[1] = ns[2] = 0;
#pragma unroll
      for (int k = 1; k <= NS; ++k) {
        const int c = __builtin_amdgcn_readlane(cand, k);
        const int lob = yl_max(nb, nt - 16 * (NS - k)), hib = yl_min(nb + 16, nt);
        nb = yl_min(yl_max(c, lob), hib);
        if (k == 2 * wv) ns[0] = nb;
        if (k == 2 * wv + 1) ns[1] = nb;
        if (k == 2 * wv + 2) ns[2] = nb;
      }
#pragma unroll
      for (int i = 0; i < 3; ++i) sb[i] = __builtin_amdgcn_readfirstlane(rp_s[ns[i]]);
    }
    const int len0 = sb[1] - sb[0], len1 = sb[2] - sb[1];
    const int nsteps = (a.abl & 1) ? 0 : ((len0 > len1 ? len0 : len1) + 15) >> 4;
    // my node slot (aggregation columns / finalize): slot l31 = 16 (stream) + node offset
    const int my_sh = l31 >> 4;
    const int my_node = ns[my_sh] + (l31 & 15);
    const bool my_valid = my_node < ns[my_sh + 1];
    const int my_rp0 = my_valid ? rp_s[my_node] : 0, my_rp1 = my_valid ? rp_s[my_node + 1] : 0;
    // gather role: edge (stream gs, row gr) of the step
    const int gs = (l31 >> 2) & 1, gr = (l31 & 3) + 4 * (l31 >> 3);
    const int g_base = sb[gs] + gr, g_last = yl_max(sb[gs + 1] - 1, 0);
    const int k_base = sb[lhi], k_len = lhi ? len1 : len0;     // the stream my aggregation k-slots belong to

    CL_STAMP();      // 2: stream set-up done
    for (int l = 0; l < a.L; ++l) {
      const int tid = cl_opaque(tid0), lane = tid & 63, l31 = lane & 31, lhi = lane >> 5;
      const float* shift = shift_s;
      // the edge phase's weight fragments (12 KB, one copy per workgroup in LDS; loaded a phase ahead) and the next node
      // phase's shifts
#pragma unroll
      for (int i = 0; i < NE; ++i)
        if (NT * (i + 1) <= 768 || tid + NT * i < 768) ef_s[tid + NT * i] = est[i];   // (the previous layer's reads are behind a barrier)
      const int l_nx = (l + 1 < a.L) ? l + 1 : 0;
      const float sh_nx = reinterpret_cast<const float*>(a.pack + (long)l_nx * CL_LAYER_BYTES + CL_EDGE_BYTES + CL_NODE_BYTES)[tid & 255];
      // =========================== node phase ===========================
      // wave (grp, rbp) computes OUT^T[64 grp .. + 63][64 rbp .. + 63] = W'_grp . in^T (+ shift): U | V | root | node branch.
      // Layer 0: the raw features as bf16 (hi | lo) halves of ONE k-step against (W_hi | W_hi) and (W_lo | 0): three of the
      // four cross terms, 2^-16 relative.  Other layers: four k-steps on the f (node branch: s) tile.
      if (!((l == 0 && (a.abl & 4)) || (l > 0 && (a.abl & 2)))) {
        cl_u32x4 bf[2][4];
        if (l == 0) {
          const float* xs = reinterpret_cast<const float*>(f_s);
#pragma unroll
          for (int rb = 0; rb < 2; ++rb) {
            const int n = 64 * rbp + 32 * rb + l31;
            const float4 x0 = *reinterpret_cast<const float4*>(xs + 8 * n);
            const float4 x1 = *reinterpret_cast<const float4*>(xs + 8 * n + 4);
            const unsigned h0 = yl_pack_bf16(x0.x, x0.y), h1 = yl_pack_bf16(x0.z, x0.w), h2 = yl_pack_bf16(x1.x, x1.y),
                           h3 = yl_pack_bf16(x1.z, x1.w);
            const unsigned l0 = yl_pack_bf16(x0.x - yl_bf16_lo(h0), x0.y - yl_bf16_hi(h0)),
                           l1 = yl_pack_bf16(x0.z - yl_bf16_lo(h1), x0.w - yl_bf16_hi(h1)),
                           l2 = yl_pack_bf16(x1.x - yl_bf16_lo(h2), x1.y - yl_bf16_hi(h2)),
                           l3 = yl_pack_bf16(x1.z - yl_bf16_lo(h3), x1.w - yl_bf16_hi(h3));
            const cl_u32x4 fr = {lhi ? l0 : h0, lhi ? l1 : h1, lhi ? l2 : h2, lhi ? l3 : h3};
            bf[rb][0] = fr; bf[rb][1] = fr;
          }
        } else {
          const unsigned char* bsrc = (grp == 3) ? s_s : f_s;
#pragma unroll
          for (int rb = 0; rb < 2; ++rb)
#pragma unroll
            for (int ks = 0; ks < 4; ++ks)
              bf[rb][ks] = *reinterpret_cast<const cl_u32x4*>(bsrc + (64 * rbp + 32 * rb + l31) * CL_FB + 32 * ks + 16 * lhi);
        }
        f32x16 acc[2][2];
#pragma unroll
        for (int nt_ = 0; nt_ < 2; ++nt_)
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const float4 s4 = *reinterpret_cast<const float4*>(shift + 64 * grp + 32 * nt_ + 8 * q + 4 * lhi);
#pragma unroll
            for (int rb = 0; rb < 2; ++rb) {
              acc[rb][nt_][4 * q] = s4.x; acc[rb][nt_][4 * q + 1] = s4.y;
              acc[rb][nt_][4 * q + 2] = s4.z; acc[rb][nt_][4 * q + 3] = s4.w;
            }
          }
        if (l == 0) {
#pragma unroll
          for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int nt_ = 0; nt_ < 2; ++nt_)
#pragma unroll
              for (int rb = 0; rb < 2; ++rb)
                acc[rb][nt_] = cl_mfma(cl_frag(af[nt_][ks]), cl_frag(bf[rb][ks]), acc[rb][nt_]);
        } else {
#pragma unroll
          for (int ks = 0; ks < 4; ++ks)
#pragma unroll
            for (int nt_ = 0; nt_ < 2; ++nt_)
#pragma unroll
              for (int rb = 0; rb < 2; ++rb)
                acc[rb][nt_] = cl_mfma(cl_frag(af[nt_][ks]), cl_frag(bf[rb][ks]), acc[rb][nt_]);
        }
#pragma unroll
        for (int rb = 0; rb < 2; ++rb) {
          const int n = 64 * rbp + 32 * rb + l31;
#pragma unroll
          for (int nt_ = 0; nt_ < 2; ++nt_) {
            if (grp == 2) {
#pragma unroll
              for (int q = 0; q < 4; ++q) {
                const int ch = 32 * nt_ + 8 * q + 4 * lhi;
                *reinterpret_cast<float4*>(r_s + n * CL_RB + 4 * ch) =
                    make_float4(acc[rb][nt_][4 * q], acc[rb][nt_][4 * q + 1], acc[rb][nt_][4 * q + 2], acc[rb][nt_][4 * q + 3]);
              }
            } else {
              // bf16 rows: a lane holds channels 8 q + 4 lhi .. + 3 (8 bytes) for q = 0..3.  Written like that, the 32 rows
              // of a half-wave fall on 8 bank groups (row strides of 272 / 144 bytes step 4 banks): 4-way conflicts on every
              // store (profiles/r06_conv_local_lds_by_phase_before.txt: 1.7 conflict cycles per LDS instruction in this
              // phase).  The two lanes of a node (l31, l31 + 32) trade halves through v_permlane32_swap instead: each ends
              // up with 8 CONSECUTIVE channels twice and stores 16 bytes — 8 rows per pass, 4 banks each: conflict-free.
              unsigned pk[4][2];
#pragma unroll
              for (int q = 0; q < 4; ++q) {
                pk[q][0] = yl_pack_bf16(acc[rb][nt_][4 * q], acc[rb][nt_][4 * q + 1]);
                pk[q][1] = yl_pack_bf16(acc[rb][nt_][4 * q + 2], acc[rb][nt_][4 * q + 3]);
                if (grp == 3) { pk[q][0] = cl_relu_pk(pk[q][0]); pk[q][1] = cl_relu_pk(pk[q][1]); }
              }
              unsigned char* row = (grp == 3) ? (s_s + n * CL_FB) : (uv_s + n * CL_UVB + 128 * grp);
#pragma unroll
              for (int j = 0; j < 2; ++j) {
                const cl_u32x2 s0 = __builtin_amdgcn_permlane32_swap(pk[j][0], pk[j + 2][0], false, false);
                const cl_u32x2 s1 = __builtin_amdgcn_permlane32_swap(pk[j][1], pk[j + 2][1], false, false);
                const cl_u32x4 o = {s0.x, s1.x, s0.y, s1.y};     // channels 32 nt + 8 j + 16 lhi .. + 7
                *reinterpret_cast<cl_u32x4*>(row + 2 * (32 * nt_ + 8 * j + 16 * lhi)) = o;
              }
            }
          }
        }
      }
      if (first_tile && tid == 0) stop_s = __hip_atomic_load(a.flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      CL_STAMP();    // 3 + 5 l: node phase done (before barrier)
      cl_lds_barrier();     // UV, R, s of this layer and its edge fragments are complete; every read of f is done
      if (first_tile && stop_s == a.flag_val) return;    // (a workgroup's first tile checks at every layer: all start together)
      CL_STAMP();    // 4 + 5 l: node barrier passed
      if (tid < 256) shift_s[tid] = sh_nx;  // read after the edge phase's closing barrier
      // the next node phase's copy of the edge fragments (next layer, or layer 0 of the next tile): a phase ahead
#pragma unroll
      for (int i = 0; i < NE; ++i)
        est[i] = reinterpret_cast<const cl_u32x4*>(a.pack + (long)l_nx * CL_LAYER_BYTES)[yl_min(tid + NT * i, 767)];
      // the NEXT layer's node-phase weight fragments — or, in the last layer, the next tile's raw loads (in front of this
      // tile's output stores: vmcnt retires in order)
      if (l + 1 < a.L) {
        const cl_u32x4* ap = reinterpret_cast<const cl_u32x4*>(a.pack + (long)(l + 1) * CL_LAYER_BYTES + CL_EDGE_BYTES) + (grp * 8) * 64 + lane;
#pragma unroll
        for (int nt_ = 0; nt_ < 2; ++nt_)
#pragma unroll
          for (int ks = 0; ks < 4; ++ks) af[nt_][ks] = ap[(nt_ * 4 + ks) * 64];
      } else if (pn0 < np && a.E > 0) {
        issue_tile_loads(pn0, pn1, pre0, pre1, pre2, pre3);
      }
      // per-proposal mean of the node branch of an output layer: s is complete and nobody writes it before the next
      // node phase.  item = (proposal, 8-channel group, row part): the four lanes of a quad take every fourth row with
      // 16-byte reads and meet through DPP
      if (l >= a.lo && !(a.abl & 8)) {
        const int j = l - a.lo;
        for (int i = tid; i < npr * 32; i += NT) {
          const int part = i & 3, cg = (i >> 2) & 7, pp = i >> 5;
          const int r0 = gseg_s[p0 + pp] - n0, r1 = gseg_s[p0 + pp + 1] - n0;
          float v[8];
#pragma unroll
          for (int k = 0; k < 8; ++k) v[k] = 0.f;
#pragma unroll 2
          for (int r = r0 + part; r < r1; r += 4) {
            const cl_u32x4 q = *reinterpret_cast<const cl_u32x4*>(s_s + r * CL_FB + 16 * cg);
            v[0] += yl_bf16_lo(q.x); v[1] += yl_bf16_hi(q.x); v[2] += yl_bf16_lo(q.y); v[3] += yl_bf16_hi(q.y);
            v[4] += yl_bf16_lo(q.z); v[5] += yl_bf16_hi(q.z); v[6] += yl_bf16_lo(q.w); v[7] += yl_bf16_hi(q.w);
          }
#pragma unroll
          for (int k = 0; k < 8; ++k) { v[k] += cl_quad_xor1(v[k]); v[k] += cl_quad_xor2(v[k]); }
          if (part == 0) {
            const int cnt = r1 - r0;
            const float sc = 1.f / (float)(cnt > 1 ? cnt : 1);
            float* z = a.Z + (long)(p_lo + p0 + pp) * a.ldz + 2 * a.F + a.D + 64 * j + 8 * cg;
            *reinterpret_cast<float4*>(z) = make_float4(v[0] * sc, v[1] * sc, v[2] * sc, v[3] * sc);
            *reinterpret_cast<float4*>(z + 4) = make_float4(v[4] * sc, v[5] * sc, v[6] * sc, v[7] * sc);
          }
        }
      }

      // =========================== edge phase ===========================
      {
        f32x16 agg0, agg1;
#pragma unroll
        for (int i = 0; i < 16; ++i) { agg0[i] = 0.f; agg1[i] = 0.f; }
        // One step = 16 edges of each of the wave's two streams, in two stages that are software-pipelined across steps
        // (stage B of step t - 1 and stage A of step t are independent: one stage's ReLU / convert / mask work runs under
        // the other's MFMAs — a single step is a chain MFMA -> VALU -> MFMA -> VALU -> MFMA):
        //   A(t): gather U'[dst] | V'[src] chunks + e_attr fragment, layer 1 transposed (10 MFMAs), ReLU + bf16 -> hp[16]
        //   B(t): layer 2 on hp (10 MFMAs), ReLU + bf16, incidence fragments, aggregation (4 MFMAs)
        auto stage_a = [&](int t, unsigned (&hp)[16]) {
          const int e = yl_min(yl_min(g_base + 16 * t, g_last), et - 1);
          const unsigned iw = idx_s[e];
          // (a step-dependent zero keeps the fragment reads inside the loop — hoisted, the layer's 12 fragments are 48
          // registers — without an asm statement, which would split the scheduling region)
          const cl_u32x4* efl = ef_s + lane + (t >> 24);
          const unsigned uo = (iw & 0xFFu) * CL_UVB + 16u * lhi, vo = ((iw >> 8) & 0xFFu) * CL_UVB + 128u + 16u * lhi;
          const cl_u32x4 aq = ab_s[e];
          const cl_bf16x8 ab = cl_frag(aq.x, aq.y, lhi ? 0u : aq.z, lhi ? 0u : aq.w);
          // layer 1, transposed: z[r] = pre-activation of channel 32 b + (r & 3) + 8 (r >> 2) + 4 lhi of MY edge
#pragma unroll
          for (int b = 0; b < 2; ++b) {
            const cl_u32x4 u0 = *reinterpret_cast<const cl_u32x4*>(uv_s + uo + 64 * b);
            const cl_u32x4 u1 = *reinterpret_cast<const cl_u32x4*>(uv_s + uo + 64 * b + 32);
            const cl_u32x4 v0 = *reinterpret_cast<const cl_u32x4*>(uv_s + vo + 64 * b);
            const cl_u32x4 v1 = *reinterpret_cast<const cl_u32x4*>(uv_s + vo + 64 * b + 32);
            f32x16 z;
#pragma unroll
            for (int i = 0; i < 16; ++i) z[i] = 0.f;
            z = cl_mfma(Id[0], cl_frag(u0), z);
            z = cl_mfma(Id[1], cl_frag(u1), z);
            z = cl_mfma(Id[0], cl_frag(v0), z);
            z = cl_mfma(Id[1], cl_frag(v1), z);
            z = cl_mfma(cl_frag(efl[(8 + b) * 64]), ab, z);
#pragma unroll
            for (int i = 0; i < 8; ++i) hp[8 * b + i] = cl_relu_pk(yl_pack_bf16(z[2 * i], z[2 * i + 1]));
          }
        };
        auto stage_b = [&](int t, const unsigned (&hp)[16]) {
          const cl_u32x4* efl = ef_s + lane + (t >> 24);
          // mean aggregation as an MFMA: incidence of the step's edges (k-slots: stream lhi, rows 8 j ..) on MY node slot
          unsigned mask;
          {
            const int base = k_base + 16 * t;
            const int lim = yl_min(16, k_len - 16 * t);
            const int lo = yl_max(my_rp0 - base, 0), hi = yl_min(my_rp1 - base, lim);
            mask = (hi > lo) ? ((0xFFFFu >> (16 - hi)) & (0xFFFFu << lo)) : 0u;
          }
          const cl_bf16x8 S0 = cl_frag(cl_sel2(mask, 0), cl_sel2(mask, 2), cl_sel2(mask, 4), cl_sel2(mask, 6));
          const cl_bf16x8 S1 = cl_frag(cl_sel2(mask, 8), cl_sel2(mask, 10), cl_sel2(mask, 12), cl_sel2(mask, 14));
          // layer 2, one 32-channel block at a time: m[r] = pre-ReLU message of edge (stream lhi, row r), channel
          // 32 nb + l31; ReLU + bf16 -> the A operand of the aggregation
#pragma unroll
          for (int nb = 0; nb < 2; ++nb) {
            f32x16 m;
#pragma unroll
            for (int i = 0; i < 16; ++i) m[i] = 0.f;
            m = cl_mfma(OnesA, cl_frag(efl[(10 + nb) * 64]), m);
#pragma unroll
            for (int sk = 0; sk < 4; ++sk)
              m = cl_mfma(cl_frag(hp[4 * sk], hp[4 * sk + 1], hp[4 * sk + 2], hp[4 * sk + 3]), cl_frag(efl[(4 * nb + sk) * 64]), m);
            const cl_bf16x8 a0 = cl_frag(cl_relu_pk(yl_pack_bf16(m[0], m[1])), cl_relu_pk(yl_pack_bf16(m[2], m[3])),
                                         cl_relu_pk(yl_pack_bf16(m[4], m[5])), cl_relu_pk(yl_pack_bf16(m[6], m[7])));
            const cl_bf16x8 a1 = cl_frag(cl_relu_pk(yl_pack_bf16(m[8], m[9])), cl_relu_pk(yl_pack_bf16(m[10], m[11])),
                                         cl_relu_pk(yl_pack_bf16(m[12], m[13])), cl_relu_pk(yl_pack_bf16(m[14], m[15])));
            if (nb == 0) { agg0 = cl_mfma(a0, S0, agg0); agg0 = cl_mfma(a1, S1, agg0); }
            else { agg1 = cl_mfma(a0, S0, agg1); agg1 = cl_mfma(a1, S1, agg1); }
          }
        };
        // A(t) and B(t - 1) as ONE hand-ordered instruction stream.  A wave issues in order and the compiler emits every
        // MFMA chain back to back with the ReLU / convert / mask blocks in between (the matrix pipe idles through each ALU
        // block, the ALU through each chain: ~1500 cycles per step for 768 cycles of MFMA).  Here every MFMA is followed by
        // <= 8 ALU instructions of the OTHER stage; CL_FENCE pins MFMA and ALU order (LDS reads and scalar code may move).
#define CL_FENCE() __builtin_amdgcn_sched_barrier(0x0014)
#define CL_PK(v, i) cl_relu_pk(yl_pack_bf16((v)[2 * (i)], (v)[2 * (i) + 1]))
        auto pipe = [&](int t, const unsigned (&hb)[16], unsigned (&ha)[16]) {
          const cl_u32x4* efl = ef_s + lane + (t >> 24);
          // ---- A(t): ids; B(t - 1): the first half's weight fragments (TB0, W2F 0..3) and A's attr weights
          const int e = yl_min(yl_min(g_base + 16 * t, g_last), et - 1);
          const unsigned iw = idx_s[e];
          cl_u32x4 fr0 = efl[10 * 64], fr1 = efl[0 * 64], fr2 = efl[1 * 64], fr3 = efl[2 * 64], fr4 = efl[3 * 64];
          const cl_u32x4 aq = ab_s[e];
          cl_u32x4 wca = efl[8 * 64];
          const unsigned uo = (iw & 0xFFu) * CL_UVB + 16u * lhi, vo = ((iw >> 8) & 0xFFu) * CL_UVB + 128u + 16u * lhi;
          cl_u32x4 gu0 = *reinterpret_cast<const cl_u32x4*>(uv_s + uo), gu1 = *reinterpret_cast<const cl_u32x4*>(uv_s + uo + 32);
          cl_u32x4 gv0 = *reinterpret_cast<const cl_u32x4*>(uv_s + vo), gv1 = *reinterpret_cast<const cl_u32x4*>(uv_s + vo + 32);
          const cl_bf16x8 ab = cl_frag(aq.x, aq.y, lhi ? 0u : aq.z, lhi ? 0u : aq.w);
          // ---- B(t - 1): incidence mask of step t - 1
          unsigned mask;
          {
            const int base = k_base + 16 * (t - 1);
            const int lim = yl_min(16, k_len - 16 * (t - 1));
            const int lo = yl_max(my_rp0 - base, 0), hi = yl_min(my_rp1 - base, lim);
            mask = (hi > lo) ? ((0xFFFFu >> (16 - hi)) & (0xFFFFu << lo)) : 0u;
          }
          unsigned sd[8], pa[8];
          f32x16 m, z;
#pragma unroll
          for (int i = 0; i < 16; ++i) { m[i] = 0.f; z[i] = 0.f; }
          CL_FENCE();
          // B: layer 2, channels 0..31  |  ALU: incidence fragments
          m = cl_mfma(OnesA, cl_frag(fr0), m);                                            CL_FENCE();
          sd[0] = cl_sel2(mask, 0); sd[1] = cl_sel2(mask, 2);                             CL_FENCE();
          m = cl_mfma(cl_frag(hb[0], hb[1], hb[2], hb[3]), cl_frag(fr1), m);              CL_FENCE();
          sd[2] = cl_sel2(mask, 4); sd[3] = cl_sel2(mask, 6);                             CL_FENCE();
          m = cl_mfma(cl_frag(hb[4], hb[5], hb[6], hb[7]), cl_frag(fr2), m);              CL_FENCE();
          sd[4] = cl_sel2(mask, 8); sd[5] = cl_sel2(mask, 10);                            CL_FENCE();
          m = cl_mfma(cl_frag(hb[8], hb[9], hb[10], hb[11]), cl_frag(fr3), m);            CL_FENCE();
          sd[6] = cl_sel2(mask, 12); sd[7] = cl_sel2(mask, 14);                           CL_FENCE();
          m = cl_mfma(cl_frag(hb[12], hb[13], hb[14], hb[15]), cl_frag(fr4), m);          CL_FENCE();
          // the second half's fragments (TB1, W2F 4..7) into the same registers: in flight under A's first chain
          fr0 = efl[11 * 64]; fr1 = efl[4 * 64]; fr2 = efl[5 * 64]; fr3 = efl[6 * 64]; fr4 = efl[7 * 64];   CL_FENCE();
          // A: layer 1, channels 0..31  |  ALU: ReLU + bf16 of B's messages
          z = cl_mfma(Id[0], cl_frag(gu0), z);                                            CL_FENCE();
          pa[0] = CL_PK(m, 0); pa[1] = CL_PK(m, 1);                                       CL_FENCE();
          z = cl_mfma(Id[1], cl_frag(gu1), z);                                            CL_FENCE();
          pa[2] = CL_PK(m, 2); pa[3] = CL_PK(m, 3);                                       CL_FENCE();
          z = cl_mfma(Id[0], cl_frag(gv0), z);                                            CL_FENCE();
          pa[4] = CL_PK(m, 4); pa[5] = CL_PK(m, 5);                                       CL_FENCE();
          z = cl_mfma(Id[1], cl_frag(gv1), z);                                            CL_FENCE();
          pa[6] = CL_PK(m, 6); pa[7] = CL_PK(m, 7);                                       CL_FENCE();
          z = cl_mfma(cl_frag(wca), ab, z);                                               CL_FENCE();
          // second half of A's gathers and attr weights (the first half's registers are free)
          gu0 = *reinterpret_cast<const cl_u32x4*>(uv_s + uo + 64); gu1 = *reinterpret_cast<const cl_u32x4*>(uv_s + uo + 96);
          gv0 = *reinterpret_cast<const cl_u32x4*>(uv_s + vo + 64); gv1 = *reinterpret_cast<const cl_u32x4*>(uv_s + vo + 96);
          wca = efl[9 * 64];                                                              CL_FENCE();
          // B: aggregation of channels 0..31, layer 2 of channels 32..63  |  ALU: ReLU + bf16 of A's hidden activations
          agg0 = cl_mfma(cl_frag(pa[0], pa[1], pa[2], pa[3]), cl_frag(sd[0], sd[1], sd[2], sd[3]), agg0);   CL_FENCE();
          ha[0] = CL_PK(z, 0); ha[1] = CL_PK(z, 1);                                       CL_FENCE();
          agg0 = cl_mfma(cl_frag(pa[4], pa[5], pa[6], pa[7]), cl_frag(sd[4], sd[5], sd[6], sd[7]), agg0);   CL_FENCE();
          ha[2] = CL_PK(z, 2); ha[3] = CL_PK(z, 3);                                       CL_FENCE();
#pragma unroll
          for (int i = 0; i < 16; ++i) m[i] = 0.f;
          m = cl_mfma(OnesA, cl_frag(fr0), m);                                            CL_FENCE();
          ha[4] = CL_PK(z, 4); ha[5] = CL_PK(z, 5);                                       CL_FENCE();
          m = cl_mfma(cl_frag(hb[0], hb[1], hb[2], hb[3]), cl_frag(fr1), m);              CL_FENCE();
          ha[6] = CL_PK(z, 6); ha[7] = CL_PK(z, 7);                                       CL_FENCE();
          m = cl_mfma(cl_frag(hb[4], hb[5], hb[6], hb[7]), cl_frag(fr2), m);              CL_FENCE();
          m = cl_mfma(cl_frag(hb[8], hb[9], hb[10], hb[11]), cl_frag(fr3), m);            CL_FENCE();
          m = cl_mfma(cl_frag(hb[12], hb[13], hb[14], hb[15]), cl_frag(fr4), m);          CL_FENCE();
          // A: layer 1, channels 32..63  |  ALU: ReLU + bf16 of B's messages
#pragma unroll
          for (int i = 0; i < 16; ++i) z[i] = 0.f;
          z = cl_mfma(Id[0], cl_frag(gu0), z);                                            CL_FENCE();
          pa[0] = CL_PK(m, 0); pa[1] = CL_PK(m, 1);                                       CL_FENCE();
          z = cl_mfma(Id[1], cl_frag(gu1), z);                                            CL_FENCE();
          pa[2] = CL_PK(m, 2); pa[3] = CL_PK(m, 3);                                       CL_FENCE();
          z = cl_mfma(Id[0], cl_frag(gv0), z);                                            CL_FENCE();
          pa[4] = CL_PK(m, 4); pa[5] = CL_PK(m, 5);                                       CL_FENCE();
          z = cl_mfma(Id[1], cl_frag(gv1), z);                                            CL_FENCE();
          pa[6] = CL_PK(m, 6); pa[7] = CL_PK(m, 7);                                       CL_FENCE();
          z = cl_mfma(cl_frag(wca), ab, z);                                               CL_FENCE();
          // B: aggregation of channels 32..63  |  ALU: ReLU + bf16 of A's hidden activations
          agg1 = cl_mfma(cl_frag(pa[0], pa[1], pa[2], pa[3]), cl_frag(sd[0], sd[1], sd[2], sd[3]), agg1);   CL_FENCE();
          ha[8] = CL_PK(z, 0); ha[9] = CL_PK(z, 1); ha[10] = CL_PK(z, 2); ha[11] = CL_PK(z, 3);             CL_FENCE();
          agg1 = cl_mfma(cl_frag(pa[4], pa[5], pa[6], pa[7]), cl_frag(sd[4], sd[5], sd[6], sd[7]), agg1);   CL_FENCE();
          ha[12] = CL_PK(z, 4); ha[13] = CL_PK(z, 5); ha[14] = CL_PK(z, 6); ha[15] = CL_PK(z, 7);           CL_FENCE();
        };
#undef CL_PK
        if (nsteps > 0) {
          unsigned h0[16], h1[16];
          stage_a(0, h0);
          int t = 1;
          for (; t + 1 < nsteps; t += 2) {
            pipe(t, h0, h1);
            pipe(t + 1, h1, h0);
          }
          if (t < nsteps) {
            pipe(t, h0, h1);
            stage_b(t, h1);
          } else {
            stage_b(t - 1, h0);
          }
        }
        CL_STAMP();  // 5 + 5 l: steps done
        // ---- finalize: f[node] = bf16(root + sum / deg); a lane holds channels 32 nb + 8 q + 4 lhi .. + 3 of its node
        if (my_valid) {
          const int deg = my_rp1 - my_rp0;
          const float inv = 1.f / (float)(deg > 1 ? deg : 1);
#pragma unroll
          for (int nb = 0; nb < 2; ++nb) {
            const f32x16& ag = nb ? agg1 : agg0;
            unsigned pk[4][2];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
              const int ch = 32 * nb + 8 * q + 4 * lhi;
              const float4 rt = *reinterpret_cast<const float4*>(r_s + my_node * CL_RB + 4 * ch);
              pk[q][0] = yl_pack_bf16(fmaf(ag[4 * q], inv, rt.x), fmaf(ag[4 * q + 1], inv, rt.y));
              pk[q][1] = yl_pack_bf16(fmaf(ag[4 * q + 2], inv, rt.z), fmaf(ag[4 * q + 3], inv, rt.w));
            }
            // (the lanes l31 / l31 + 32 of a node trade halves: 16-byte stores, see the node phase)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
              const cl_u32x2 s0 = __builtin_amdgcn_permlane32_swap(pk[j][0], pk[j + 2][0], false, false);
              const cl_u32x2 s1 = __builtin_amdgcn_permlane32_swap(pk[j][1], pk[j + 2][1], false, false);
              const cl_u32x4 o = {s0.x, s1.x, s0.y, s1.y};
              *reinterpret_cast<cl_u32x4*>(f_s + my_node * CL_FB + 2 * (32 * nb + 8 * j + 16 * lhi)) = o;
            }
          }
        }
      }
      cl_lds_barrier();     // f of this layer is complete; every gather of UV / R is done
      CL_STAMP();    // 6 + 5 l: edge barrier passed

      // =========================== outputs of this layer ===========================
      // f rows -> feats, per-proposal max of f -> Z (nobody writes f before the next edge phase's finalize)
      if (l >= a.lo && !(a.abl & 8)) {
        const int tid = cl_opaque(tid0);
        const int j = l - a.lo;
        for (int i = tid; i < nt * 8; i += NT) {
          const int n = i >> 3, c = i & 7;
          const cl_u32x4 v = *reinterpret_cast<const cl_u32x4*>(f_s + n * CL_FB + 16 * c);
          *reinterpret_cast<cl_u32x4*>(a.feats + (long)(n0 + n) * a.ld_feats + 64 * j + 8 * c) = v;
        }
        for (int i = tid; i < npr * 32; i += NT) {
          const int part = i & 3, cg = (i >> 2) & 7, pp = i >> 5;
          const int r0 = gseg_s[p0 + pp] - n0, r1 = gseg_s[p0 + pp + 1] - n0;
          float v[8];
#pragma unroll
          for (int k = 0; k < 8; ++k) v[k] = -INFINITY;
#pragma unroll 2
          for (int r = r0 + part; r < r1; r += 4) {
            const cl_u32x4 q = *reinterpret_cast<const cl_u32x4*>(f_s + r * CL_FB + 16 * cg);
            v[0] = fmaxf(v[0], yl_bf16_lo(q.x)); v[1] = fmaxf(v[1], yl_bf16_hi(q.x));
            v[2] = fmaxf(v[2], yl_bf16_lo(q.y)); v[3] = fmaxf(v[3], yl_bf16_hi(q.y));
            v[4] = fmaxf(v[4], yl_bf16_lo(q.z)); v[5] = fmaxf(v[5], yl_bf16_hi(q.z));
            v[6] = fmaxf(v[6], yl_bf16_lo(q.w)); v[7] = fmaxf(v[7], yl_bf16_hi(q.w));
          }
#pragma unroll
          for (int k = 0; k < 8; ++k) { v[k] = fmaxf(v[k], cl_quad_xor1(v[k])); v[k] = fmaxf(v[k], cl_quad_xor2(v[k])); }
          if (part == 0) {
            const bool any = r1 > r0;
            float* z = a.Z + (long)(p_lo + p0 + pp) * a.ldz + a.F + 64 * j + 8 * cg;
            *reinterpret_cast<float4*>(z) = make_float4(any ? v[0] : 0.f, any ? v[1] : 0.f, any ? v[2] : 0.f, any ? v[3] : 0.f);
            *reinterpret_cast<float4*>(z + 4) = make_float4(any ? v[4] : 0.f, any ? v[5] : 0.f, any ? v[6] : 0.f, any ? v[7] : 0.f);
          }
        }
      }
      CL_STAMP();    // 7 + 5 l: outputs of the layer done
    }
    // Z[p, 0:F] = 0 for the tile's proposals (the fusion launch max-accumulates into it)
    if (!(a.abl & 8)) {
      const int f4 = a.F >> 2;
      const float zf = __int_as_float(cl_opaque(0));        // (a hoisted zero vector was the kernel's last spill)
      const float4 z4 = make_float4(zf, zf, zf, zf);
      for (int i = tid; i < npr * f4; i += NT) {
        const int pp = i / f4, c = i - pp * f4;
        *reinterpret_cast<float4*>(a.Z + (long)(p_lo + p0 + pp) * a.ldz + 4 * c) = z4;
      }
    }
    cl_lds_barrier();       // the tiles are dead: the next tile may be written
    p0 = pn0;
    p1 = pn1;
    first_tile = false;
  }
#undef CL_STAMP
}

// ------------------------------------------------------------------------------------------------
// packed weight image (once per weight version)
// ------------------------------------------------------------------------------------------------
struct ClPackLayer {
  const float *Wuv, *Wr, *Wn, *Wc4, *s1, *uv_scale, *uv_shift, *br, *bn, *sn, *tn, *t2f;
  const u16* W2f;
  int Cin;
};

__global__ void k_conv_local_pack(ClPackLayer p, int layer, unsigned char* dst) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;     // one 16-byte piece each
  const int lane = i & 63, l31 = lane & 31, lhi = lane >> 5;
  cl_u32x4* out = reinterpret_cast<cl_u32x4*>(dst);
  auto bf = [](float v) { return yl_pack_bf16(v, 0.f) & 0xFFFFu; };
  if (i < 12 * 64) {
    const int fi = i >> 6;
    cl_u32x4 o = {0u, 0u, 0u, 0u};
    if (fi < 8) {                     // W2F (nb, sk): elements e -> W2f[32 nb + l31][16 sk + 4 lhi + (e & 3) + 8 (e >> 2)]
      const int nb = fi >> 2, sk = fi & 3;
      const u16* q = p.W2f + (32 * nb + l31) * 64 + 16 * sk + 4 * lhi;
      o.x = q[0] | ((unsigned)q[1] << 16); o.y = q[2] | ((unsigned)q[3] << 16);
      o.z = q[8] | ((unsigned)q[9] << 16); o.w = q[10] | ((unsigned)q[11] << 16);
    } else if (fi < 10) {             // WCA[b]: s1 * Wc4 as bf16 (hi, lo)
      const int c = 32 * (fi - 8) + l31;
      const float sc = p.s1 ? p.s1[c] : 1.f;
      const float w0 = p.Wc4[4 * c] * sc, w1 = p.Wc4[4 * c + 1] * sc, w2 = p.Wc4[4 * c + 2] * sc, w3 = p.Wc4[4 * c + 3] * sc;
      const unsigned h01 = yl_pack_bf16(w0, w1), h23 = yl_pack_bf16(w2, w3);
      const unsigned l01 = yl_pack_bf16(w0 - yl_bf16_lo(h01), w1 - yl_bf16_hi(h01));
      const unsigned l23 = yl_pack_bf16(w2 - yl_bf16_lo(h23), w3 - yl_bf16_hi(h23));
      if (lhi) { o.x = l01; o.y = l23; } else { o.x = h01; o.y = h23; o.z = h01; o.w = h23; }
    } else {                          // TB[nb]: t2f as three bf16 terms
      const float t = p.t2f[32 * (fi - 10) + l31];
      const unsigned th = __float_as_uint(t) & 0xFFFF0000u;
      const float r1 = t - __uint_as_float(th);
      const unsigned tm = __float_as_uint(r1) & 0xFFFF0000u;
      const float r2 = r1 - __uint_as_float(tm);
      const unsigned tl = yl_pack_bf16(r2, 0.f) & 0xFFFFu;
      if (!lhi) { o.x = (th >> 16) | tm; o.y = tl; }
    }
    out[i] = o;
    return;
  }
  const int j = i - 12 * 64;
  cl_u32x4* nout = reinterpret_cast<cl_u32x4*>(dst + CL_EDGE_BYTES);
  // row c of the stacked, scale-folded node weight [256][Cin]: U | V | root | node branch
  auto wrow = [&](int c, int k) -> float {
    if (c < 128) return p.Wuv[(long)c * p.Cin + k] * (p.uv_scale ? p.uv_scale[c] : 1.f);
    if (c < 192) return p.Wr[(long)(c - 128) * p.Cin + k];
    return p.Wn[(long)(c - 192) * p.Cin + k] * (p.sn ? p.sn[c - 192] : 1.f);
  };
  if (j < 32 * 64) {
    cl_u32x4 o = {0u, 0u, 0u, 0u};
    if (layer == 0) {                 // fragments (g, nt, ks): ks 0 = W_hi[ch][0..7] in both lane halves, ks 1 = W_lo | 0
      const int fi = j >> 6, g = fi >> 3, nt_ = (fi >> 2) & 1, ks = fi & 3;
      const int c = 64 * g + 32 * nt_ + l31;
      if (ks < 2 && !(ks == 1 && lhi)) {
        unsigned v[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          const float w = k < p.Cin ? wrow(c, k) : 0.f;
          const unsigned h = bf(w);
          v[k] = ks == 0 ? h : bf(w - __uint_as_float(h << 16));
        }
        o.x = v[0] | (v[1] << 16); o.y = v[2] | (v[3] << 16); o.z = v[4] | (v[5] << 16); o.w = v[6] | (v[7] << 16);
      }
    } else {                          // A fragments (g, nt, ks): elements e -> W'[64 g + 32 nt + l31][16 ks + 8 lhi + e]
      const int fi = j >> 6, g = fi >> 3, nt_ = (fi >> 2) & 1, ks = fi & 3;
      const int c = 64 * g + 32 * nt_ + l31, k0 = 16 * ks + 8 * lhi;
      o.x = bf(wrow(c, k0)) | (bf(wrow(c, k0 + 1)) << 16); o.y = bf(wrow(c, k0 + 2)) | (bf(wrow(c, k0 + 3)) << 16);
      o.z = bf(wrow(c, k0 + 4)) | (bf(wrow(c, k0 + 5)) << 16); o.w = bf(wrow(c, k0 + 6)) | (bf(wrow(c, k0 + 7)) << 16);
    }
    nout[j] = o;
    return;
  }
  const int c = j - 32 * 64;
  if (c < 256) {
    float* sh = reinterpret_cast<float*>(dst + CL_EDGE_BYTES + CL_NODE_BYTES);
    float v;
    if (c < 128) v = p.uv_shift ? p.uv_shift[c] : 0.f;
    else if (c < 192) v = p.br ? p.br[c - 128] : 0.f;
    else v = (p.bn ? p.bn[c - 192] : 0.f) * (p.sn ? p.sn[c - 192] : 1.f) + (p.tn ? p.tn[c - 192] : 0.f);
    sh[c] = v;
  }
}

}  // namespace

extern "C" size_t yolat_conv_local_pack_bytes(int64_t n_layers) {
  return n_layers > 0 ? (size_t)n_layers * CL_LAYER_BYTES : 0;
}

// can the one-launch conv stack run this model at all?  (shapes only; the per-batch property is checked on the device)
bool yl_conv_local_model_ok(const yolat_model_eval_bf16* mh) {
  if (!mh || !mh->base) return false;
  const yolat_model_eval* m = mh->base;
  if (m->C != 64 || m->n_blocks < 1 || m->n_blocks > YOLAT_MAX_LAYERS || m->n_blocks_out < 1 || m->n_blocks_out > m->n_blocks)
    return false;
  if (m->F % 4 != 0) return false;
  for (int l = 0; l < m->n_blocks; ++l) {
    const yolat_conv_eval& cv = m->conv[l];
    if (!cv.Wuv || !cv.Wc4 || !cv.Wr || !cv.Wn || !mh->W2[l] || !mh->t2f[l]) return false;
    if (l == 0 ? (cv.Cin < 1 || cv.Cin > 8) : (cv.Cin != 64)) return false;
  }
  return true;
}

extern "C" int yolat_conv_local_pack(const yolat_model_eval_bf16* mh, void* dst, size_t dst_bytes, yolat_stream_t stream) {
  if (!mh || !mh->base || !dst) return YOLAT_E_INVALID;
  if (!yl_conv_local_model_ok(mh)) return YOLAT_E_UNSUPPORTED;
  const yolat_model_eval* m = mh->base;
  if (dst_bytes < yolat_conv_local_pack_bytes(m->n_blocks) || (((uintptr_t)dst) & 15) != 0) return YOLAT_E_INVALID;
  for (int l = 0; l < m->n_blocks; ++l) {
    const yolat_conv_eval& cv = m->conv[l];
    ClPackLayer p;
    p.Wuv = cv.Wuv; p.Wr = cv.Wr; p.Wn = cv.Wn; p.Wc4 = cv.Wc4; p.s1 = cv.s1;
    p.uv_scale = mh->uv_scale[l]; p.uv_shift = mh->uv_shift[l]; p.br = cv.br; p.bn = cv.bn; p.sn = cv.sn; p.tn = cv.tn;
    p.t2f = mh->t2f[l]; p.W2f = mh->W2[l]; p.Cin = (int)cv.Cin;
    const int pieces = 12 * 64 + 32 * 64 + 256;
    hipLaunchKernelGGL(k_conv_local_pack, dim3(yl_cdiv(pieces, 256)), dim3(256), 0, (hipStream_t)stream, p, l,
                       reinterpret_cast<unsigned char*>(dst) + (size_t)l * CL_LAYER_BYTES);
    YL_LAUNCH_CHECK();
  }
  return 0;
}

// ------------------------------------------------------------------------------------------------
// Proposal structure of a COO batch, without the global destination sort (what the COO instantiation needs instead of
// yolat_graph_prepare): seg_ptr / node_seg from bbox_idx, eptr [P + 1] = first edge of every proposal from the edge
// list — valid when the list is GROUPED by proposal (bbox_idx[dst] non-decreasing along it: Datasets/graph_dict3.py:725
// appends a proposal's edges as one block, :752-764 keeps per-proposal edge ranges) — and the violations of the
// locality property as flag bits (YOLAT_LOC_*).  One thread per node and per edge.
// ------------------------------------------------------------------------------------------------
namespace {
__global__ void __launch_bounds__(256) k_local_prep(const long long* __restrict__ edge, long se, long sc,
                                                    const long long* __restrict__ bbox_idx, int N, int E, int P,
                                                    int* __restrict__ seg_ptr, int* __restrict__ node_seg,
                                                    int* __restrict__ eptr, int* status, int* info, int vouched) {
  const int t = blockIdx.x * 256 + threadIdx.x;
  auto seg_of = [&](long n, bool flag) -> int {
    long b = bbox_idx[n];
    if (b < 0 || b >= P) {
      if (flag) atomicOr(status, YOLAT_STATUS_SEG_RANGE);
      b = b < 0 ? 0 : P - 1;
    }
    return (int)b;
  };
  if (t < N) {
    const int cur = seg_of(t, true);
    const int prev = t > 0 ? seg_of(t - 1, false) : -1;
    node_seg[t] = cur;
    if (cur < prev) atomicOr(status, YOLAT_STATUS_SEG_UNSORTED);
    else for (int p = prev + 1; p <= cur; ++p) seg_ptr[p] = t;
    if (t == N - 1) for (int p = cur + 1; p <= P; ++p) seg_ptr[p] = N;
  }
  if (t < E) {
    auto node_of = [&](long e, long col, bool flag) -> long {
      long v = edge[e * se + col * sc];
      if (v < 0 || v >= N) {
        if (flag) atomicOr(status, YOLAT_STATUS_EDGE_RANGE);
        v = v < 0 ? 0 : N - 1;
      }
      return v;
    };
    const int pd = seg_of(node_of(t, 1, true), false), ps = seg_of(node_of(t, 0, true), false);
    const int prev = t > 0 ? seg_of(node_of(t - 1, 1, false), false) : -1;
    int bad = 0;
    if (ps != pd) bad |= YOLAT_LOC_CROSSING;
    if (pd < prev) bad |= YOLAT_LOC_UNGROUPED;
    else for (int p = prev + 1; p <= pd; ++p) eptr[p] = t;
    if (t == E - 1) for (int p = pd + 1; p <= P; ++p) eptr[p] = E;
    if (bad) {
      if (info) atomicOr(info, bad);
      if (vouched) atomicOr(status, YOLAT_STATUS_NOT_LOCAL);
    }
  }
  if (E == 0) for (int p = t; p <= P; p += gridDim.x * 256) eptr[p] = 0;
}

// info[1] = nodes of the largest proposal, info[2] = edges of the largest proposal (info zeroed by the caller)
__global__ void __launch_bounds__(256) k_local_sizes(const int* __restrict__ seg_ptr, const int* __restrict__ eptr, int P,
                                                     int* info) {
  const int p = blockIdx.x * 256 + threadIdx.x;
  int n = 0, e = 0;
  if (p < P) { n = seg_ptr[p + 1] - seg_ptr[p]; e = eptr[p + 1] - eptr[p]; }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) { n = yl_max(n, __shfl_xor(n, off)); e = yl_max(e, __shfl_xor(e, off)); }
  if ((threadIdx.x & 63) == 0) { atomicMax(info + 1, n); atomicMax(info + 2, e); }
}
__global__ void k_local_info_zero(int* info) { if (threadIdx.x < 4) info[threadIdx.x] = 0; }
__global__ void k_local_info_fold(const int* status, int* info) { if (threadIdx.x == 0) info[3] = status[0]; }
}  // namespace

int yl_local_prep(const int64_t* edge, int64_t se, int64_t sc, const int64_t* bbox_idx, int64_t N, int64_t E, int64_t P,
                  int32_t* seg_ptr, int32_t* node_seg, int32_t* eptr, int32_t* status, int32_t* info, bool vouched,
                  hipStream_t st) {
  const long n = N > E ? N : E;
  hipLaunchKernelGGL(k_local_prep, dim3(yl_cdiv(n > 0 ? n : 1, 256)), dim3(256), 0, st,
                     reinterpret_cast<const long long*>(edge), (long)se, (long)sc, reinterpret_cast<const long long*>(bbox_idx),
                     (int)N, (int)E, (int)P, seg_ptr, node_seg, eptr, status, info, vouched ? 1 : 0);
  YL_LAUNCH_CHECK();
  return 0;
}

// tile shape the launcher will pick for a batch of P proposals: nodes / edges a proposal may have
static void yl_conv_local_tile(int64_t P, int* nw, int* t_nodes, int* t_edges) {
  const int w = (g_cl_nw == 4 || g_cl_nw == 8) ? g_cl_nw : (P >= 2048 ? 8 : 4);
  *nw = w; *t_nodes = 16 * w; *t_edges = 128 * w;
}

extern "C" size_t yolat_batch_locality_workspace_bytes(int64_t N, int64_t E, int64_t P) {
  if (N <= 0 || E < 0 || P <= 0) return 0;
  return (size_t)(2 * (P + 1) + N + 64) * sizeof(int32_t) + 1024;
}

extern "C" int yolat_batch_locality(const int64_t* edge, int64_t stride_e, int64_t stride_c, const int64_t* bbox_idx,
                                    int64_t N, int64_t E, int64_t P, int32_t* info, void* workspace,
                                    size_t workspace_bytes, yolat_stream_t stream) {
  if (!bbox_idx || !info || !workspace || N <= 0 || E < 0 || P <= 0 || (E > 0 && !edge)) return YOLAT_E_INVALID;
  if (N >= (1LL << 30) || E >= (1LL << 30) || P >= (1LL << 30)) return YOLAT_E_UNSUPPORTED;
  if (workspace_bytes < yolat_batch_locality_workspace_bytes(N, E, P)) return YOLAT_E_INVALID;
  hipStream_t st = (hipStream_t)stream;
  char* base = reinterpret_cast<char*>(((uintptr_t)workspace + 255) & ~(uintptr_t)255);
  int32_t* seg_ptr = reinterpret_cast<int32_t*>(base);
  int32_t* eptr = seg_ptr + (P + 1);
  int32_t* node_seg = eptr + (P + 1);
  int32_t* scratch_status = node_seg + N;
  hipLaunchKernelGGL(k_local_info_zero, dim3(1), dim3(64), 0, st, info);
  hipLaunchKernelGGL(k_local_info_zero, dim3(1), dim3(64), 0, st, scratch_status);
  YL_LAUNCH_CHECK();
  int rc = yl_local_prep(edge, stride_e, stride_c, bbox_idx, N, E, P, seg_ptr, node_seg, eptr, scratch_status, info, false, st);
  if (rc != 0) return rc;
  hipLaunchKernelGGL(k_local_sizes, dim3(yl_cdiv(P, 256)), dim3(256), 0, st, seg_ptr, eptr, (int)P, info);
  // malformed ids (status bits of the scratch word) make the batch unfit as well: info[3] carries them
  hipLaunchKernelGGL(k_local_info_fold, dim3(1), dim3(64), 0, st, scratch_status, info);
  YL_LAUNCH_CHECK();
  return 0;
}

extern "C" int yolat_conv_local_fits(const yolat_locality* loc, int64_t P) {
  if (!loc || !loc->known || loc->flags != 0) return 0;
  int nw, tn, te;
  yl_conv_local_tile(P, &nw, &tn, &te);
  return (loc->max_nodes <= tn && loc->max_edges <= te) ? 1 : 0;
}

// launch on a prepared (destination-sorted) graph, or — in.edge != nullptr — on the raw COO list with per-proposal edge
// ranges (yl_local_prep); *flag = flag_val when the batch turns out not to have the property
int yl_conv_local_bf16(const yolat_model_eval_bf16* mh, const void* pack, const float* x, int64_t ldx, const YlLocalIn& in,
                       int64_t N, int64_t E, int64_t P, uint16_t* feats, int64_t ld_feats, float* Z, int64_t ldz, int32_t* flag,
                       int32_t flag_val, hipStream_t st) {
  const yolat_model_eval* m = mh->base;
  ClArgs a;
  a.x = x; a.ldx = (int)ldx; a.cin0 = (int)m->conv[0].Cin;
  a.row_ptr = in.row_ptr; a.src = in.src; a.dst = in.dst; a.attr = in.attr; a.seg_ptr = in.seg_ptr;
  a.edge = reinterpret_cast<const long long*>(in.edge); a.se = (long)in.se; a.sc = (long)in.sc; a.eptr = in.eptr;
  a.status = in.status;
  a.N = (int)N; a.E = (int)E; a.P = (int)P;
  // 8-wave workgroups (128-node tiles, one per CU) once there are enough proposals to give each of 256 workgroups a few
  // tiles; 4-wave workgroups (64-node tiles, two per CU) below that
  int nw, tn, te;
  yl_conv_local_tile(P, &nw, &tn, &te);
  const long slots = nw == 8 ? 256 : 512;          // one round: one 8-wave / two 4-wave workgroups per CU
  long g0 = (P + slots - 1) / slots;
  if (g0 < 4) g0 = 4;
  if (g0 > CL_GMAX) g0 = CL_GMAX;
  if (g_cl_g0 > 0 && g_cl_g0 <= CL_GMAX) g0 = g_cl_g0;
  a.G0 = (int)g0;
  a.L = m->n_blocks; a.lo = m->n_blocks - m->n_blocks_out;
  a.pack = reinterpret_cast<const unsigned char*>(pack);
  a.feats = feats; a.ld_feats = (int)ld_feats;
  a.Z = Z; a.ldz = (int)ldz; a.F = (int)m->F; a.D = (int)(m->C * m->n_blocks_out);
  a.flag = flag; a.flag_val = flag_val;
  a.abl = g_cl_abl;
  a.stamps = g_cl_stamps;
  const dim3 grid((unsigned)((P + g0 - 1) / g0));
  const bool coo = in.edge != nullptr && E > 0;
  if (coo) {
    if (nw == 8) hipLaunchKernelGGL((k_conv_local_h<8, true>), grid, dim3(512), 0, st, a);
    else hipLaunchKernelGGL((k_conv_local_h<4, true>), grid, dim3(256), 0, st, a);
  } else {
    if (nw == 8) hipLaunchKernelGGL((k_conv_local_h<8, false>), grid, dim3(512), 0, st, a);
    else hipLaunchKernelGGL((k_conv_local_h<4, false>), grid, dim3(256), 0, st, a);
  }
  YL_LAUNCH_CHECK();
  return 0;
}

extern "C" int yolat_conv_stack_local_bf16(const yolat_model_eval_bf16* mh, const void* pack, const float* x, int64_t ldx,
                                           const yolat_graph_csr* g, int64_t N, int64_t E, int64_t P, uint16_t* feats,
                                           int64_t ld_feats, float* Z, int64_t ldz, int32_t* flag, yolat_stream_t stream) {
  if (!mh || !mh->base || !pack || !x || !g || !feats || !Z || !flag || N <= 0 || E < 0 || P <= 0) return YOLAT_E_INVALID;
  if (!g->row_ptr || !g->seg_ptr || (E > 0 && (!g->src || !g->dst || !g->attr))) return YOLAT_E_INVALID;
  if (!yl_conv_local_model_ok(mh)) return YOLAT_E_UNSUPPORTED;
  const yolat_model_eval* m = mh->base;
  const long D = m->C * m->n_blocks_out;
  if (ld_feats < D || ld_feats % 8 != 0 || (((uintptr_t)feats) & 15) != 0 || ldz < 2 * (m->F + D) || ldz % 4 != 0 ||
      !yl_aligned16(Z) || (E > 0 && !yl_aligned16(g->attr)) || N >= (1LL << 30) || E >= (1LL << 30))
    return YOLAT_E_UNSUPPORTED;
  YlLocalIn in{};
  in.row_ptr = g->row_ptr; in.src = g->src; in.dst = g->dst; in.attr = g->attr; in.seg_ptr = g->seg_ptr;
  return yl_conv_local_bf16(mh, pack, x, ldx, in, N, E, P, feats, ld_feats, Z, ldz, flag, 1, (hipStream_t)stream);
}

// The same launch on the RAW edge list (COO order, grouped by proposal): the proposal structure comes from
// yl_local_prep into `workspace` ((2 (P + 1) + N) int32), the tiles sort their edges in LDS.
extern "C" int yolat_conv_stack_local_bf16_coo(const yolat_model_eval_bf16* mh, const void* pack, const float* x, int64_t ldx,
                                               const int64_t* edge, int64_t stride_e, int64_t stride_c, const float* e_attr,
                                               const int64_t* bbox_idx, int64_t N, int64_t E, int64_t P, uint16_t* feats,
                                               int64_t ld_feats, float* Z, int64_t ldz, int32_t* flag, int32_t* status,
                                               void* workspace, size_t workspace_bytes, yolat_stream_t stream) {
  if (!mh || !mh->base || !pack || !x || !bbox_idx || !feats || !Z || !flag || !status || !workspace || N <= 0 || E < 0 ||
      P <= 0 || (E > 0 && (!edge || !e_attr)))
    return YOLAT_E_INVALID;
  if (!yl_conv_local_model_ok(mh)) return YOLAT_E_UNSUPPORTED;
  const yolat_model_eval* m = mh->base;
  const long D = m->C * m->n_blocks_out;
  if (ld_feats < D || ld_feats % 8 != 0 || (((uintptr_t)feats) & 15) != 0 || ldz < 2 * (m->F + D) || ldz % 4 != 0 ||
      !yl_aligned16(Z) || (E > 0 && !yl_aligned16(e_attr)) || N >= (1LL << 30) || E >= (1LL << 30))
    return YOLAT_E_UNSUPPORTED;
  if (workspace_bytes < yolat_batch_locality_workspace_bytes(N, E, P)) return YOLAT_E_INVALID;
  char* base = reinterpret_cast<char*>(((uintptr_t)workspace + 255) & ~(uintptr_t)255);
  int32_t* seg_ptr = reinterpret_cast<int32_t*>(base);
  int32_t* eptr = seg_ptr + (P + 1);
  int32_t* node_seg = eptr + (P + 1);
  hipStream_t st = (hipStream_t)stream;
  YlLocalIn in{};
  in.attr = e_attr; in.seg_ptr = seg_ptr; in.edge = edge; in.se = stride_e; in.sc = stride_c; in.eptr = eptr; in.status = status;
  int rc = yl_local_prep(edge, stride_e, stride_c, bbox_idx, N, E, P, seg_ptr, node_seg, eptr, status, nullptr, true, st);
  if (rc != 0) return rc;
  return yl_conv_local_bf16(mh, pack, x, ldx, in, N, E, P, feats, ld_feats, Z, ldz, flag, 1, st);
}

// tuning / debug hook (tests, tools/exp/conv_local_bench.py): waves per workgroup (4 | 8, 0 = automatic), proposals per
// workgroup (0 = automatic), phase-ablation bits (results are then WRONG on purpose: 1 no edge steps, 2 no node phase of
// layers >= 1, 4 no node phase of layer 0, 8 no outputs), device buffer of 64 int64 phase stamps per workgroup or NULL
extern "C" void yolat_conv_local_tune(int nw, int g0, int abl, long long* stamps) {
  g_cl_nw = nw; g_cl_g0 = g0; g_cl_abl = abl; g_cl_stamps = stamps;
}
