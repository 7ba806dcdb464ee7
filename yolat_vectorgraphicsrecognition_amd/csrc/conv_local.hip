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

constexpr int CL_T = 64;                 // nodes per tile
constexpr int CL_ET = 640;               // edges per tile
constexpr int CL_UVB = 272;              // UV tile row stride, bytes (128 bf16 + 16 B pad)
constexpr int CL_RB = 272;               // R tile row stride, bytes (64 fp32 + 16 B pad)
constexpr int CL_FB = 144;               // f / s tile row stride, bytes (64 bf16 + 16 B pad)
constexpr int CL_GMAX = 64;              // proposals per workgroup (group arrays in LDS)
constexpr int CL_EDGE_BYTES = 12 * 1024; // packed image, per layer: W2F[8] WCA[2] TB[2] fragments
constexpr int CL_NODE_BYTES = 32 * 1024; //   node weights: 4 groups x (2 x 4) fragments (layer 0: W0 [256][8] fp32)
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
// bits pos, pos + 1 of mask -> packed bf16 pair of 1.0 / 0.0
__device__ __forceinline__ unsigned cl_sel2(unsigned mask, int pos) {
  const unsigned t = (mask >> pos) & 3u;
  return ((t * 0x8001u) & 0x10001u) * 0x3F80u;
}

struct ClArgs {
  const float* x; int ldx; int cin0;
  const int* row_ptr; const int* src; const int* dst; const float* attr; const int* seg_ptr;
  int N, E, P, G0;
  int L, lo;
  const unsigned char* pack;
  u16* feats; int ld_feats;
  float* Z; int ldz; int F, D;
  int* flag; int flag_val;
  int abl;
};

__global__ void __launch_bounds__(256, 2) k_conv_local_h(const ClArgs a) {
  __shared__ __attribute__((aligned(16))) unsigned char uv_s[CL_T * CL_UVB];
  __shared__ __attribute__((aligned(16))) unsigned char r_s[CL_T * CL_RB];
  __shared__ __attribute__((aligned(16))) unsigned char f_s[CL_T * CL_FB];      // also: the x tile [64][8] fp32
  __shared__ __attribute__((aligned(16))) unsigned char s_s[CL_T * CL_FB];
  __shared__ __attribute__((aligned(16))) cl_u32x4 ab_s[CL_ET];
  __shared__ unsigned idx_s[CL_ET];
  __shared__ int rp_s[CL_T + 4];
  __shared__ int gseg_s[CL_GMAX + 1], grow_s[CL_GMAX + 1];

  const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, lhi = lane >> 5;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int p_lo = blockIdx.x * a.G0;
  const int np = yl_min(a.G0, a.P - p_lo);
  if (np <= 0) return;
  for (int i = tid; i <= np; i += 256) {
    const int s = a.seg_ptr[p_lo + i];
    gseg_s[i] = s;
    grow_s[i] = a.row_ptr[yl_min(yl_max(s, 0), a.N)];
  }
  __syncthreads();

  // identity fragments of the transposed first layer: A[m][k] = 1 iff k == m - 16 j  (edge_chain.hip)
  cl_bf16x8 Id[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const bool mine = ((l31 >> 4) == j) && (((l31 >> 3) & 1) == lhi);
    const int i = l31 & 7;
    unsigned d[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) d[k] = (mine && (i >> 1) == k) ? (0x3F80u << (16 * (i & 1))) : 0u;
    Id[j] = cl_frag(d[0], d[1], d[2], d[3]);
  }
  const cl_bf16x8 OnesA = lhi ? cl_frag(0u, 0u, 0u, 0u) : cl_frag(0x3F803F80u, 0x00003F80u, 0u, 0u);

  int p0 = 0;
  while (p0 < np) {
    // ---- greedy tile: proposals [p0, p1) with <= CL_T nodes and <= CL_ET edges
    int p1 = p0;
    const int sg0 = gseg_s[p0], rg0 = grow_s[p0];
    while (p1 < np && gseg_s[p1 + 1] - sg0 <= CL_T && grow_s[p1 + 1] - rg0 <= CL_ET && gseg_s[p1 + 1] >= gseg_s[p1]) ++p1;
    if (p1 == p0) {                    // a proposal that does not fit (or an unsorted segment table): the gated path runs
      if (tid == 0) *a.flag = a.flag_val;
      ++p0;
      continue;
    }
    const int n0 = sg0, nt = gseg_s[p1] - sg0, e0 = rg0, et = grow_s[p1] - rg0;
    const int npr = p1 - p0;

    // ---- tile -> LDS: row_ptr (tile-local), x rows, packed edge ids, e_attr fragments
    if (tid <= nt) rp_s[tid] = a.row_ptr[n0 + tid] - e0;
    {
      float* xs = reinterpret_cast<float*>(f_s);
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        const int i = tid + 256 * t, n = i >> 3, k = i & 7;
        const float v = a.x[(long)yl_min(n0 + n, a.N - 1) * a.ldx + yl_min(k, a.cin0 - 1)];
        xs[i] = (n < nt && k < a.cin0) ? v : 0.f;
      }
    }
    {
      bool bad = false;
#pragma unroll
      for (int t = 0; t < (CL_ET + 255) / 256; ++t) {
        const int e = tid + 256 * t;
        const int ec = yl_min(e0 + yl_min(e, et > 0 ? et - 1 : 0), a.E > 0 ? a.E - 1 : 0);
        const int d = a.dst[ec] - n0, s = a.src[ec] - n0;
        const float4 q = *reinterpret_cast<const float4*>(a.attr + 4l * ec);
        if (e < et) {
          const bool ok = (unsigned)s < (unsigned)nt && (unsigned)d < (unsigned)nt;
          bad |= !ok;
          idx_s[e] = ok ? ((unsigned)d | ((unsigned)s << 8)) : 0u;
          const unsigned h01 = yl_pack_bf16(q.x, q.y), h23 = yl_pack_bf16(q.z, q.w);
          const unsigned l01 = yl_pack_bf16(q.x - yl_bf16_lo(h01), q.y - yl_bf16_hi(h01));
          const unsigned l23 = yl_pack_bf16(q.z - yl_bf16_lo(h23), q.w - yl_bf16_hi(h23));
          const cl_u32x4 fr = {h01, h23, l01, l23};
          ab_s[e] = fr;
        }
      }
      if (bad) *a.flag = a.flag_val;
    }
    __syncthreads();

    // ---- the wave's two edge streams: node boundaries nb[0..8] (edge-balanced, node-aligned, <= 16 nodes each)
    int ns[3], sb[3];
    {
      int cand = 0;
      {
        const int k = lane < 8 ? lane : 8;
        if (et > 0) {
          const int tgt = (et * k) >> 3;
          const int d = (int)(idx_s[yl_min(tgt, et - 1)] & 0xFFu);
          cand = (k >= 8) ? nt : ((rp_s[d] == tgt) ? d : d + 1);
        } else {
          cand = (nt * k) >> 3;
        }
      }
      int nb = 0;
      ns[0] = ns[1] = ns[2] = 0;
#pragma unroll
      for (int k = 1; k <= 8; ++k) {
        const int c = __builtin_amdgcn_readlane(cand, k);
        const int lob = yl_max(nb, nt - 16 * (8 - k)), hib = yl_min(nb + 16, nt);
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

    for (int l = 0; l < a.L; ++l) {
      const unsigned char* pk = a.pack + (long)l * CL_LAYER_BYTES;
      const float* shift = reinterpret_cast<const float*>(pk + CL_EDGE_BYTES + CL_NODE_BYTES);
      // =========================== node phase ===========================
      if (l == 0 && (a.abl & 4)) {
      } else if (l > 0 && (a.abl & 2)) {
      } else if (l == 0) {
        // K = in_channels raw features: thread = column pair (2 cp, 2 cp + 1) of [U | V | R | S], half of the rows
        const int cp = tid & 127, rh = tid >> 7;
        const float* w = reinterpret_cast<const float*>(pk + CL_EDGE_BYTES) + 16 * cp;
        const float4 w00 = *reinterpret_cast<const float4*>(w), w01 = *reinterpret_cast<const float4*>(w + 4);
        const float4 w10 = *reinterpret_cast<const float4*>(w + 8), w11 = *reinterpret_cast<const float4*>(w + 12);
        const float2 sh = *reinterpret_cast<const float2*>(shift + 2 * cp);
        const float* xs = reinterpret_cast<const float*>(f_s);
#pragma unroll 4
        for (int i = 0; i < 32; ++i) {
          const int n = 32 * rh + i;
          const float4 x0 = *reinterpret_cast<const float4*>(xs + 8 * n), x1 = *reinterpret_cast<const float4*>(xs + 8 * n + 4);
          float v0 = sh.x, v1 = sh.y;
          v0 = fmaf(w00.x, x0.x, v0); v0 = fmaf(w00.y, x0.y, v0); v0 = fmaf(w00.z, x0.z, v0); v0 = fmaf(w00.w, x0.w, v0);
          v0 = fmaf(w01.x, x1.x, v0); v0 = fmaf(w01.y, x1.y, v0); v0 = fmaf(w01.z, x1.z, v0); v0 = fmaf(w01.w, x1.w, v0);
          v1 = fmaf(w10.x, x0.x, v1); v1 = fmaf(w10.y, x0.y, v1); v1 = fmaf(w10.z, x0.z, v1); v1 = fmaf(w10.w, x0.w, v1);
          v1 = fmaf(w11.x, x1.x, v1); v1 = fmaf(w11.y, x1.y, v1); v1 = fmaf(w11.z, x1.z, v1); v1 = fmaf(w11.w, x1.w, v1);
          if (cp < 64) {
            *reinterpret_cast<unsigned*>(uv_s + n * CL_UVB + 4 * cp) = yl_pack_bf16(v0, v1);
          } else if (cp < 96) {
            *reinterpret_cast<float2*>(r_s + n * CL_RB + 8 * (cp - 64)) = make_float2(v0, v1);
          } else {
            *reinterpret_cast<unsigned*>(s_s + n * CL_FB + 4 * (cp - 96)) = yl_pack_bf16(fmaxf(v0, 0.f), fmaxf(v1, 0.f));
          }
        }
      } else {
        // outputs of layer l - 1 that leave the chip: waves 0-2 copy f rows / take the per-proposal max, wave 3 the mean of s
        // (done below, after this layer's operands are in registers)
        const unsigned char* bsrc = (wv == 3) ? s_s : f_s;
        cl_u32x4 bf[2][4];
#pragma unroll
        for (int rb = 0; rb < 2; ++rb)
#pragma unroll
          for (int ks = 0; ks < 4; ++ks)
            bf[rb][ks] = *reinterpret_cast<const cl_u32x4*>(bsrc + (32 * rb + l31) * CL_FB + 32 * ks + 16 * lhi);
        const cl_u32x4* ap = reinterpret_cast<const cl_u32x4*>(pk + CL_EDGE_BYTES) + (wv * 8) * 64 + lane;
        cl_u32x4 af[2][4];
#pragma unroll
        for (int nt_ = 0; nt_ < 2; ++nt_)
#pragma unroll
          for (int ks = 0; ks < 4; ++ks) af[nt_][ks] = ap[(nt_ * 4 + ks) * 64];
        f32x16 acc[2][2];
#pragma unroll
        for (int nt_ = 0; nt_ < 2; ++nt_)
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const float4 s4 = *reinterpret_cast<const float4*>(shift + 64 * wv + 32 * nt_ + 8 * q + 4 * lhi);
#pragma unroll
            for (int rb = 0; rb < 2; ++rb) {
              acc[rb][nt_][4 * q] = s4.x; acc[rb][nt_][4 * q + 1] = s4.y;
              acc[rb][nt_][4 * q + 2] = s4.z; acc[rb][nt_][4 * q + 3] = s4.w;
            }
          }
#pragma unroll
        for (int ks = 0; ks < 4; ++ks)
#pragma unroll
          for (int nt_ = 0; nt_ < 2; ++nt_)
#pragma unroll
            for (int rb = 0; rb < 2; ++rb)
              acc[rb][nt_] = cl_mfma(cl_frag(af[nt_][ks]), cl_frag(bf[rb][ks]), acc[rb][nt_]);
#pragma unroll
        for (int rb = 0; rb < 2; ++rb) {
          const int n = 32 * rb + l31;
#pragma unroll
          for (int nt_ = 0; nt_ < 2; ++nt_)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
              const int ch = 32 * nt_ + 8 * q + 4 * lhi;
              const float v0 = acc[rb][nt_][4 * q], v1 = acc[rb][nt_][4 * q + 1], v2 = acc[rb][nt_][4 * q + 2],
                          v3 = acc[rb][nt_][4 * q + 3];
              if (wv < 2) {
                cl_u32x2 o = {yl_pack_bf16(v0, v1), yl_pack_bf16(v2, v3)};
                *reinterpret_cast<cl_u32x2*>(uv_s + n * CL_UVB + 128 * wv + 2 * ch) = o;
              } else if (wv == 2) {
                *reinterpret_cast<float4*>(r_s + n * CL_RB + 4 * ch) = make_float4(v0, v1, v2, v3);
              } else {
                cl_u32x2 o = {cl_relu_pk(yl_pack_bf16(v0, v1)), cl_relu_pk(yl_pack_bf16(v2, v3))};
                *reinterpret_cast<cl_u32x2*>(s_s + n * CL_FB + 2 * ch) = o;
              }
            }
        }
      }
      __syncthreads();      // UV, R (and s) of this layer are complete; every read of f is done

      // =========================== edge phase ===========================
      {
        const cl_u32x4* ep = reinterpret_cast<const cl_u32x4*>(pk) + lane;
        cl_bf16x8 w2f[8], WcA[2], TB[2];
        if (a.abl & 16) {
#pragma unroll
          for (int i = 0; i < 8; ++i) w2f[i] = Id[i & 1];
          WcA[0] = WcA[1] = TB[0] = TB[1] = Id[0];
        } else {
#pragma unroll
        for (int i = 0; i < 8; ++i) w2f[i] = cl_frag(ep[i * 64]);
#pragma unroll
        for (int i = 0; i < 2; ++i) { WcA[i] = cl_frag(ep[(8 + i) * 64]); TB[i] = cl_frag(ep[(10 + i) * 64]); }
        }
        f32x16 agg0, agg1;
#pragma unroll
        for (int i = 0; i < 16; ++i) { agg0[i] = 0.f; agg1[i] = 0.f; }
        for (int t = 0; t < nsteps; ++t) {
          const int e = yl_min(yl_min(g_base + 16 * t, g_last), et - 1);
          const unsigned iw = idx_s[e];
          const unsigned uo = (iw & 0xFFu) * CL_UVB + 16u * lhi, vo = ((iw >> 8) & 0xFFu) * CL_UVB + 128u + 16u * lhi;
          cl_u32x4 ru[4], rv[4];
#pragma unroll
          for (int ks = 0; ks < 4; ++ks) {
            ru[ks] = *reinterpret_cast<const cl_u32x4*>(uv_s + uo + 32 * ks);
            rv[ks] = *reinterpret_cast<const cl_u32x4*>(uv_s + vo + 32 * ks);
          }
          const cl_u32x4 aq = ab_s[e];
          const cl_bf16x8 ab = cl_frag(aq.x, aq.y, lhi ? 0u : aq.z, lhi ? 0u : aq.w);
          // ---- layer 1, transposed: z[r] = pre-activation of channel 32 b + (r & 3) + 8 (r >> 2) + 4 lhi of MY edge
          unsigned hp[16];
#pragma unroll
          for (int b = 0; b < 2; ++b) {
            f32x16 z;
#pragma unroll
            for (int i = 0; i < 16; ++i) z[i] = 0.f;
            z = cl_mfma(Id[0], cl_frag(ru[2 * b]), z);
            z = cl_mfma(Id[1], cl_frag(ru[2 * b + 1]), z);
            z = cl_mfma(Id[0], cl_frag(rv[2 * b]), z);
            z = cl_mfma(Id[1], cl_frag(rv[2 * b + 1]), z);
            z = cl_mfma(WcA[b], ab, z);
#pragma unroll
            for (int i = 0; i < 8; ++i) hp[8 * b + i] = cl_relu_pk(yl_pack_bf16(z[2 * i], z[2 * i + 1]));
          }
          // ---- layer 2: m_nb[r] = pre-ReLU message of edge (stream lhi, row r), channel 32 nb + l31
          f32x16 m0, m1;
#pragma unroll
          for (int i = 0; i < 16; ++i) { m0[i] = 0.f; m1[i] = 0.f; }
          m0 = cl_mfma(OnesA, TB[0], m0);
          m1 = cl_mfma(OnesA, TB[1], m1);
#pragma unroll
          for (int sk = 0; sk < 4; ++sk) {
            const cl_bf16x8 hA = cl_frag(hp[4 * sk], hp[4 * sk + 1], hp[4 * sk + 2], hp[4 * sk + 3]);
            m0 = cl_mfma(hA, w2f[sk], m0);
            m1 = cl_mfma(hA, w2f[4 + sk], m1);
          }
          // ---- mean aggregation as an MFMA: incidence of the step's edges (k-slots: stream lhi, rows 8 j ..) on MY node slot
          unsigned mask;
          {
            const int base = k_base + 16 * t;
            const int lim = yl_min(16, k_len - 16 * t);
            const int lo = yl_max(my_rp0 - base, 0), hi = yl_min(my_rp1 - base, lim);
            mask = (hi > lo) ? ((0xFFFFu >> (16 - hi)) & (0xFFFFu << lo)) : 0u;
          }
#pragma unroll
          for (int j = 0; j < 2; ++j) {
            const cl_bf16x8 S = cl_frag(cl_sel2(mask, 8 * j), cl_sel2(mask, 8 * j + 2), cl_sel2(mask, 8 * j + 4),
                                        cl_sel2(mask, 8 * j + 6));
            const cl_bf16x8 a0 = cl_frag(cl_relu_pk(yl_pack_bf16(m0[8 * j], m0[8 * j + 1])),
                                         cl_relu_pk(yl_pack_bf16(m0[8 * j + 2], m0[8 * j + 3])),
                                         cl_relu_pk(yl_pack_bf16(m0[8 * j + 4], m0[8 * j + 5])),
                                         cl_relu_pk(yl_pack_bf16(m0[8 * j + 6], m0[8 * j + 7])));
            const cl_bf16x8 a1 = cl_frag(cl_relu_pk(yl_pack_bf16(m1[8 * j], m1[8 * j + 1])),
                                         cl_relu_pk(yl_pack_bf16(m1[8 * j + 2], m1[8 * j + 3])),
                                         cl_relu_pk(yl_pack_bf16(m1[8 * j + 4], m1[8 * j + 5])),
                                         cl_relu_pk(yl_pack_bf16(m1[8 * j + 6], m1[8 * j + 7])));
            agg0 = cl_mfma(a0, S, agg0);
            agg1 = cl_mfma(a1, S, agg1);
          }
        }
        // ---- finalize: f[node] = bf16(root + sum / deg); a lane holds channels 32 nb + 8 q + 4 lhi .. + 3 of its node
        if (my_valid) {
          const int deg = my_rp1 - my_rp0;
          const float inv = 1.f / (float)(deg > 1 ? deg : 1);
#pragma unroll
          for (int nb = 0; nb < 2; ++nb)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
              const int ch = 32 * nb + 8 * q + 4 * lhi;
              const float4 rt = *reinterpret_cast<const float4*>(r_s + my_node * CL_RB + 4 * ch);
              const f32x16& ag = nb ? agg1 : agg0;
              cl_u32x2 o = {yl_pack_bf16(fmaf(ag[4 * q], inv, rt.x), fmaf(ag[4 * q + 1], inv, rt.y)),
                            yl_pack_bf16(fmaf(ag[4 * q + 2], inv, rt.z), fmaf(ag[4 * q + 3], inv, rt.w))};
              *reinterpret_cast<cl_u32x2*>(f_s + my_node * CL_FB + 2 * ch) = o;
            }
        }
      }
      __syncthreads();      // f of this layer is complete; every gather of UV / R is done

      // =========================== outputs of this layer ===========================
      if (l >= a.lo && !(a.abl & 8)) {
        const int j = l - a.lo;
        if (wv < 3) {
          for (int i = tid; i < nt * 8; i += 192) {
            const int n = i >> 3, c = i & 7;
            const cl_u32x4 v = *reinterpret_cast<const cl_u32x4*>(f_s + n * CL_FB + 16 * c);
            *reinterpret_cast<cl_u32x4*>(a.feats + (long)(n0 + n) * a.ld_feats + 64 * j + 8 * c) = v;
          }
          for (int i = tid; i < npr * 64; i += 192) {
            const int pp = i >> 6, c = i & 63;
            const int r0 = gseg_s[p0 + pp] - n0, r1 = gseg_s[p0 + pp + 1] - n0;
            float best = 0.f;
            bool any = false;
            for (int r = r0; r < r1; ++r) {
              const float v = __uint_as_float((unsigned)*reinterpret_cast<const u16*>(f_s + r * CL_FB + 2 * c) << 16);
              if (!any || v > best) { best = v; any = true; }
            }
            a.Z[(long)(p_lo + p0 + pp) * a.ldz + a.F + 64 * j + c] = best;
          }
        } else {
          for (int pp = 0; pp < npr; ++pp) {
            const int r0 = gseg_s[p0 + pp] - n0, r1 = gseg_s[p0 + pp + 1] - n0;
            float sm = 0.f;
            for (int r = r0; r < r1; ++r)
              sm += __uint_as_float((unsigned)*reinterpret_cast<const u16*>(s_s + r * CL_FB + 2 * lane) << 16);
            const int cnt = r1 - r0;
            a.Z[(long)(p_lo + p0 + pp) * a.ldz + 2 * a.F + a.D + 64 * j + lane] = sm / (float)(cnt > 1 ? cnt : 1);
          }
        }
      }
    }
    // Z[p, 0:F] = 0 for the tile's proposals (the fusion launch max-accumulates into it)
    if (!(a.abl & 8)) {
      const int f4 = a.F >> 2;
      const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
      for (int i = tid; i < npr * f4; i += 256) {
        const int pp = i / f4, c = i - pp * f4;
        *reinterpret_cast<float4*>(a.Z + (long)(p_lo + p0 + pp) * a.ldz + 4 * c) = z4;
      }
    }
    __syncthreads();        // the tiles are dead: the next tile may be loaded
    p0 = p1;
  }
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
    if (layer == 0) {                 // W0 [256][8] fp32, columns >= Cin zero
      if (j < 512) {
        const int c = j >> 1, k0 = 4 * (j & 1);
        float v[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) v[k] = (k0 + k < p.Cin) ? wrow(c, k0 + k) : 0.f;
        o.x = __float_as_uint(v[0]); o.y = __float_as_uint(v[1]); o.z = __float_as_uint(v[2]); o.w = __float_as_uint(v[3]);
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

// launch on a prepared (destination-sorted) graph; *flag = flag_val when the gated per-layer path must run
int yl_conv_local_bf16(const yolat_model_eval_bf16* mh, const void* pack, const float* x, int64_t ldx, const int32_t* row_ptr,
                       const int32_t* src, const int32_t* dst, const float* attr, const int32_t* seg_ptr, int64_t N,
                       int64_t E, int64_t P, uint16_t* feats, int64_t ld_feats, float* Z, int64_t ldz, int32_t* flag,
                       int32_t flag_val, hipStream_t st) {
  const yolat_model_eval* m = mh->base;
  ClArgs a;
  a.x = x; a.ldx = (int)ldx; a.cin0 = (int)m->conv[0].Cin;
  a.row_ptr = row_ptr; a.src = src; a.dst = dst; a.attr = attr; a.seg_ptr = seg_ptr;
  a.N = (int)N; a.E = (int)E; a.P = (int)P;
  long g0 = (P + 511) / 512;          // one round of two workgroups per CU
  if (g0 < 4) g0 = 4;
  if (g0 > CL_GMAX) g0 = CL_GMAX;
  {
    const char* e = getenv("YOLAT_CONV_LOCAL_G0");
    if (e && atoi(e) > 0 && atoi(e) <= CL_GMAX) g0 = atoi(e);
  }
  a.G0 = (int)g0;
  a.L = m->n_blocks; a.lo = m->n_blocks - m->n_blocks_out;
  a.pack = reinterpret_cast<const unsigned char*>(pack);
  a.feats = feats; a.ld_feats = (int)ld_feats;
  a.Z = Z; a.ldz = (int)ldz; a.F = (int)m->F; a.D = (int)(m->C * m->n_blocks_out);
  a.flag = flag; a.flag_val = flag_val;
  { const char* e = getenv("YOLAT_CONV_LOCAL_ABL"); a.abl = e ? atoi(e) : 0; }
  hipLaunchKernelGGL(k_conv_local_h, dim3((unsigned)((P + g0 - 1) / g0)), dim3(256), 0, st, a);
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
  return yl_conv_local_bf16(mh, pack, x, ldx, g->row_ptr, g->src, g->dst, g->attr, g->seg_ptr, N, E, P, feats, ld_feats, Z,
                            ldz, flag, 1, (hipStream_t)stream);
}
