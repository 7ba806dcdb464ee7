/*
 * yolat_hip.h — C ABI of libyolat_hip.so: the MI355X (gfx950) kernels of the YOLaT GNN
 * message-passing hot path (SparseCADGCN forward / backward / Adam).
 *
 * The reference (microsoft/YOLaT-VectorGraphicsRecognition) is pure Python and has no FFI layer
 * of its own: its "operators" for this path are torch / torch_geometric / torch_scatter calls
 * made from
 *     cad_recognition/architecture3cc_rpn_gp_iter2.py   (model wiring, :44-71, :106-137, :358-379)
 *     gcn_lib/sparse/torch_vertex.py                    (AttrRelativeEdgeConvGlobalPool2, :288-341)
 *     gcn_lib/sparse/torch_nn.py                        (MLP = Linear/BatchNorm1d/ReLU, :50-71)
 *     cad_recognition/train.py                          (Adam step, :212,283-284)
 * Each entry point below names the reference call it replaces.  A ctypes binding is in
 * yolat_vectorgraphicsrecognition_amd/_lib.py; INTEGRATION.md shows the stub a reference
 * maintainer would add.
 *
 * Conventions
 *  - every pointer is a DEVICE pointer (HBM) unless marked "host"; the library never allocates,
 *    never frees, never synchronises; all work is enqueued on `stream` (a hipStream_t);
 *  - float tensors are fp32 row-major with an explicit leading dimension `ld*` (in elements), so
 *    a column slice of a concat buffer can be passed without a copy;
 *  - device index arrays are int32 (after yolat_coo_to_csr); raw inputs are int64 as in the
 *    reference (Datasets/graph_dict3.py:1049-1066);
 *  - return value: 0 = enqueued; <0 = invalid argument (YOLAT_E_*); >0 = hipError_t from launch.
 *  - `status` words are device int32 flags OR-ed by kernels that validate data (bit meanings
 *    YOLAT_STATUS_*); the caller reads them back when it next synchronises.
 *  - optional pointers may be NULL where the comment says "nullable".
 */
#ifndef YOLAT_HIP_H
#define YOLAT_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* yolat_stream_t; /* hipStream_t */

#define YOLAT_E_INVALID   (-1) /* bad size / NULL pointer                    */
#define YOLAT_E_UNSUPPORTED (-2) /* shape outside what the kernels implement */

#define YOLAT_STATUS_EDGE_RANGE  1 /* an edge endpoint is outside [0,N)      */
#define YOLAT_STATUS_SEG_UNSORTED 2 /* bbox_idx is not non-decreasing        */
#define YOLAT_STATUS_SEG_RANGE   4 /* bbox_idx value outside [0,P)           */
#define YOLAT_STATUS_NOT_LOCAL   8 /* a batch the caller vouched for as proposal-local (yolat_locality) is not */

/* A-operand "prologue": the consumer applies  a' = a*scale[k] + shift[k]; if (relu) a' = max(a',0)
 * while loading, so BatchNorm1d+ReLU outputs (gcn_lib/sparse/torch_nn.py:58-66) need not be
 * materialised.  scale == NULL means identity. */

int yolat_abi_version(void);
const char* yolat_strerror(int code);

/* ------------------------------------------------------------------------------------------
 * Graph pre-processing (integer, bit-exact).  Replaces what PyG's MessagePassing.propagate does
 * implicitly with edge_index (gather by edge_index[0]/[1], scatter by edge_index[1];
 * gcn_lib/sparse/torch_vertex.py:324) by an explicit destination-sorted CSR.
 * ------------------------------------------------------------------------------------------ */

/* Stable counting sort of E edges by destination.
 *   edge: int64, element (e, c) at edge[e*stride_e + c*stride_c]; c=0 source j, c=1 target i
 *         (data.edge is [E,2]: stride_e=2, stride_c=1; the model's .T view is the same memory:
 *         architecture3cc_rpn_gp_iter2.py:110)
 *   row_ptr[N+1], perm[E] (CSR slot -> original edge id, ascending inside a row),
 *   src_csr[E], dst_csr[E]
 *   work: int32 scratch of yolat_csr_work_elems(N,E) elements.                                  */
size_t yolat_csr_work_elems(int64_t N, int64_t E);
int yolat_coo_to_csr(const int64_t* edge, int64_t stride_e, int64_t stride_c, int64_t E, int64_t N,
                     int32_t* row_ptr, int32_t* perm, int32_t* src_csr, int32_t* dst_csr,
                     int32_t* work, int32_t* status, yolat_stream_t stream);

/* CSC by source over the CSR slots (needed only by the backward scatter to x[src]):
 *   col_ptr[N+1], slots[E] = CSR slots of the edges leaving each node, ascending.
 *   work: int32 scratch of yolat_csc_work_elems(N) elements.                                    */
size_t yolat_csc_work_elems(int64_t N);
int yolat_csc_by_source(const int32_t* src_csr, int64_t E, int64_t N, int32_t* col_ptr,
                        int32_t* slots, int32_t* work, yolat_stream_t stream);

/* Sub-batch extraction for SparseCADGCN.predict's two passes (architecture3cc_rpn_gp_iter2.py:153-242):
 * the Python list / dict / per-edge loops become integer kernels, bit-exact with them.
 *   yolat_expand_ranges:    out[i] = start[j] + (i - prefix[j]) for the range j containing slot i;
 *                           start[S], prefix[S+1] exclusive prefix of the range lengths, total = prefix[S]
 *   yolat_subgraph_reindex: node_ids[n_sub] old node ids of the subset (new id = position; on duplicates
 *                           the later position wins, like the reference's dict), edge_ids[m_sub] rows of the
 *                           batch edge list -> edge_out[m_sub,2] int64 in new ids (an endpoint outside the
 *                           subset raises YOLAT_STATUS_EDGE_RANGE: the reference raises KeyError);
 *                           bbox_idx_out[n_sub] int64 = run-length renumbering of bbox_idx[node_ids]
 *                           work: int32 scratch of yolat_subgraph_work_elems(N, n_sub)
 *   yolat_gather_rows_bytes: dst[r] = src[idx[r]] for rows of row_bytes (multiple of 4) bytes            */
int yolat_expand_ranges(const int32_t* start, const int32_t* prefix, int64_t S, int64_t total, int32_t* out,
                        yolat_stream_t stream);
size_t yolat_subgraph_work_elems(int64_t N, int64_t n_sub);
int yolat_subgraph_reindex(const int32_t* node_ids, int64_t n_sub, int64_t N, const int64_t* edge,
                           int64_t stride_e, int64_t stride_c, const int32_t* edge_ids, int64_t m_sub,
                           const int64_t* bbox_idx, int64_t* edge_out, int64_t* bbox_idx_out, int32_t* work,
                           int32_t* status, yolat_stream_t stream);
int yolat_gather_rows_bytes(const void* src, int64_t src_row_bytes, const int32_t* idx, int64_t rows,
                            int64_t row_bytes, void* dst, int64_t dst_row_bytes, yolat_stream_t stream);

/* Offset fix-up of a collated batch of B images (cad_recognition/train.py:238-258), one launch:
 *   edge[e,:]  += node_off[b]  for edge_ptr[b] <= e < edge_ptr[b+1]     (edge [E,2] int64 contiguous)
 *   bbox_idx[n] += prop_off[b] for node_ptr[b] <= n < node_ptr[b+1]
 * edge_ptr / node_ptr [B+1], node_off / prop_off [B], all int64 on the device.                          */
int yolat_fixup_offsets(int64_t* edge, int64_t E, const int64_t* edge_ptr, int64_t* bbox_idx, int64_t N,
                        const int64_t* node_ptr, const int64_t* node_off, const int64_t* prop_off, int64_t B,
                        yolat_stream_t stream);

/* seg_ptr[P+1] from a non-decreasing int64 bbox_idx[N] (Datasets/graph_dict3.py:732):
 * seg_ptr[p] = first row r with bbox_idx[r] >= p.  Also writes node_seg[N] (int32 copy).
 * Replaces the implicit segmentation done by torch_scatter.scatter(index=bbox_idx)
 * (architecture3cc_rpn_gp_iter2.py:67,122).                                                     */
int yolat_segment_ptr(const int64_t* bbox_idx, int64_t N, int64_t P, int32_t* seg_ptr,
                      int32_t* node_seg, int32_t* status, yolat_stream_t stream);

/* All of the above in one call (4 kernel launches + 1 memset): CSR by destination (stable), e_attr
 * permuted to CSR order, and — when bbox_idx is not NULL — the proposal segment pointers.
 *   e_attr [E,4] contiguous, 16-byte aligned;  work: int32 scratch of yolat_graph_work_elems(N,E).   */
size_t yolat_graph_work_elems(int64_t N, int64_t E);
int yolat_graph_prepare(const int64_t* edge, int64_t stride_e, int64_t stride_c, const float* e_attr,
                        const int64_t* bbox_idx, int64_t E, int64_t N, int64_t P, int32_t* row_ptr,
                        int32_t* perm, int32_t* src_csr, int32_t* dst_csr, float* attr_csr,
                        int32_t* seg_ptr, int32_t* node_seg, int32_t* work, int32_t* status,
                        yolat_stream_t stream);

/* dst[r, 0:width] = src[idx[r], 0:width]  (e_attr -> CSR order; fp32)                           */
int yolat_gather_rows(const float* src, int64_t ld_src, const int32_t* idx, int64_t rows,
                      int64_t width, float* dst, int64_t ld_dst, yolat_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Dense layers: nn.Linear (+ BatchNorm1d + ReLU) of gcn_lib/sparse/torch_nn.py:50-71.
 * fp32-input MFMA (v_mfma_f32_32x32x2_f32), fp32 accumulate.
 * ------------------------------------------------------------------------------------------ */

/* Y[M,Nout] = epi( pro(A)[M,K] . W[Nout,K]^T + bias )
 *   pro: a_scale/a_shift [K] nullable, a_relu
 *   epi: o_scale/o_shift [Nout] nullable (eval-mode BatchNorm folded), o_relu
 *   accumulate != 0: Y += (result)   (used for  out += lin_r(x), torch_vertex.py:325)
 *   stats (nullable): fp32 scratch of yolat_bn_stats_elems(M,Nout) elements; receives per
 *        32-row-group (sum, M2) partials of the pre-epilogue values (incl. bias) for training-mode
 *        BatchNorm (batch statistics over all M rows, torch_nn.py:27); reduce with
 *        yolat_bn_finalize (which also uses the tail of the buffer as fp64 scratch).              */
#define YOLAT_STATS_ROWS 32
size_t yolat_bn_stats_elems(int64_t M, int64_t C);
int yolat_linear_fwd(const float* A, int64_t lda, int64_t M, int64_t K,
                     const float* a_scale, const float* a_shift, int a_relu,
                     const float* W, int64_t ldw, const float* bias, int64_t Nout,
                     const float* o_scale, const float* o_shift, int o_relu,
                     float* Y, int64_t ldy, int accumulate, float* stats, yolat_stream_t stream);

/* Eval-mode fusion_block + per-proposal max pooling fused (arch:61-63 + arch:122):
 *   pool[p, 0:Nout] = max_{rows r of proposal p} relu((A[r].W^T + bias)*o_scale + o_shift)
 * node_seg[M] = proposal id of each row (non-decreasing).  `pool` MUST be zero-filled first
 * (yolat_pool_prepare does it); the [M,Nout] activation is never materialised.                    */
int yolat_linear_segmax_fwd(const float* A, int64_t lda, int64_t M, int64_t K, const float* W,
                            int64_t ldw, const float* bias, int64_t Nout, const float* o_scale,
                            const float* o_shift, const int32_t* node_seg, float* pool, int64_t ldpool,
                            yolat_stream_t stream);

/* Same, with W used transposed: Y[M,Nout] = pro(A)[M,K] . Wt[K,Nout]   (dX = dY . W)            */
int yolat_linear_fwd_wt(const float* A, int64_t lda, int64_t M, int64_t K,
                        const float* Wt, int64_t ldw, int64_t Nout,
                        float* Y, int64_t ldy, int accumulate, yolat_stream_t stream);

/* dW[Nout,K] (+)= dY[M,Nout]^T . pro(A)[M,K];  db[Nout] (+)= colsum(dY)  (db nullable)
 *   partial: fp32 scratch of yolat_linear_bwd_w_work_elems(M,Nout,K) elements; the split-row
 *   partial products are reduced in a fixed order (deterministic, no atomics).                  */
size_t yolat_linear_bwd_w_work_elems(int64_t M, int64_t Nout, int64_t K);
int yolat_linear_bwd_w(const float* dY, int64_t lddy, int64_t M, int64_t Nout,
                       const float* A, int64_t lda, int64_t K,
                       const float* a_scale, const float* a_shift, int a_relu,
                       float* dW, int64_t lddw, float* db, int accumulate,
                       float* partial, yolat_stream_t stream);

/* Training-mode BatchNorm1d statistics from the row-block partials of yolat_linear_fwd /
 * yolat_edge_lin1_fwd:  mean, biased var -> invstd; scale = gamma*invstd; shift = beta-mean*scale;
 * running_mean/var updated with `momentum` and the UNBIASED variance (torch defaults,
 * torch_nn.py:27).  fp64 Chan merge in row-block order (deterministic).
 * save_mean/save_invstd [C] are kept for the backward.                                           */
int yolat_bn_finalize(const float* stats, int64_t M, int64_t C, const float* gamma,
                      const float* beta, float* running_mean, float* running_var,
                      float momentum, float eps, float* save_mean, float* save_invstd,
                      float* scale, float* shift, yolat_stream_t stream);

/* Eval-mode coefficients from running stats: scale = gamma/sqrt(var+eps), shift = beta-mean*scale */
int yolat_bn_eval_coeffs(const float* gamma, const float* beta, const float* running_mean,
                         const float* running_var, float eps, int64_t C, float* scale,
                         float* shift, yolat_stream_t stream);

/* Z[M,C] = max(Y*scale + shift, 0)  (relu optional)                                             */
int yolat_scale_shift_relu(const float* Y, int64_t ldy, int64_t M, int64_t C, const float* scale,
                           const float* shift, int relu, float* Z, int64_t ldz,
                           yolat_stream_t stream);

/* Backward of Z = relu(BN_train(Y)) given dZ:  (torch autograd of torch_nn.py:58-66)
 *   step 1 (reduce): per-column s1 = sum dyh, s2 = sum dyh*xhat with dyh = dZ*[Z>0],
 *           xhat = (Y-mean)*invstd; writes dgamma(+)=s2, dbeta(+)=s1 and coef[2*C] = (s1/M, s2/M)
 *   step 2 (apply):  dY = scale*(dyh - coef1 - xhat*coef2)
 *   work: fp32 scratch of yolat_bn_bwd_work_elems(M,C).
 *   If relu == 0 the mask is all-ones.                                                           */
size_t yolat_bn_bwd_work_elems(int64_t M, int64_t C);
int yolat_bn_relu_bwd(const float* dZ, int64_t lddz, const float* Y, int64_t ldy, int64_t M,
                      int64_t C, const float* gamma, const float* save_mean,
                      const float* save_invstd, const float* scale, const float* shift, int relu,
                      float* dgamma, float* dbeta, int accumulate, float* dY, int64_t lddy,
                      float* work, yolat_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Edge convolution AttrRelativeEdgeConvGlobalPool2 (gcn_lib/sparse/torch_vertex.py:288-341):
 *   m_e = nn([x_i, x_j - x_i, a_e]);  out_i = mean_{e->i} m_e + lin_r(x_i)
 * All [E,*] tensors are in CSR (destination-sorted) order.
 * ------------------------------------------------------------------------------------------ */

/* H1[E,C] = epi( [x[dst], x[src]-x[dst], attr] . W1[C,2Cin+4]^T + b1 )
 * (gather = PyG __lift__, cat+Linear = torch_vertex.py:331,335 / nn.0).  epi/stats as in
 * yolat_linear_fwd.  The 2Cin+4 wide edge-feature matrix is never materialised.                 */
int yolat_edge_lin1_fwd(const float* x, int64_t ldx, int64_t N, int64_t Cin,
                        const int32_t* src_csr, const int32_t* dst_csr, const float* attr_csr,
                        int64_t E, const float* W1, int64_t ldw, const float* b1, int64_t C,
                        const float* o_scale, const float* o_shift, int o_relu,
                        float* H1, int64_t ldh, float* stats, yolat_stream_t stream);

/* dW1[C,2Cin+4] (+)= dH1^T . [x[dst], x[src]-x[dst], attr];  db1 (+)= colsum(dH1)               */
int yolat_edge_lin1_bwd_w(const float* dH1, int64_t lddh, int64_t E, int64_t C,
                          const float* x, int64_t ldx, int64_t N, int64_t Cin,
                          const int32_t* src_csr, const int32_t* dst_csr, const float* attr_csr,
                          float* dW1, int64_t lddw, float* db1, int accumulate, float* partial,
                          yolat_stream_t stream);

/* dG[E,2Cin] = dH1[E,C] . Wc,  Wc[:,0:Cin] = W1[:,0:Cin]-W1[:,Cin:2Cin], Wc[:,Cin:2Cin] = W1[:,Cin:2Cin]
 * i.e. dG[:,0:Cin] is the gradient that flows to x[dst_e], dG[:,Cin:2Cin] to x[src_e].          */
int yolat_edge_lin1_bwd_x(const float* dH1, int64_t lddh, int64_t E, int64_t C, const float* W1,
                          int64_t ldw, int64_t Cin, float* dG, int64_t lddg,
                          yolat_stream_t stream);

/* dX[n,0:Cin] (+)= sum_{q in CSR row n} dG[q,0:Cin] + sum_{q in CSC col n} dG[q,Cin:2Cin]
 * (autograd of the two index_selects = two scatter-adds; here an atomic-free gather-reduce).     */
int yolat_edge_scatter_bwd(const float* dG, int64_t lddg, int64_t Cin, const int32_t* row_ptr,
                           const int32_t* col_ptr, const int32_t* slots, int64_t N, float* dX,
                           int64_t lddx, int accumulate, yolat_stream_t stream);

/* Eval-mode edge MLP `self.nn` of AttrRelativeEdgeConvGlobalPool2 (torch_vertex.py:311,331-335), both
 * Linear+BN+ReLU layers in one kernel, BatchNorm folded into (s, t):
 *   H2[q] = relu(s2*(W2.relu(s1*(W1.[x[dst] | x[src]-x[dst] | attr](q) + b1) + t1) + b2) + t2)
 * W1 [C, 2Cin+4] and W2 [C, C] row-major contiguous, C must be 64.  Bit-identical to
 * yolat_edge_lin1_fwd followed by yolat_linear_fwd; the [E,C] hidden activation stays in LDS.        */
int yolat_edge_mlp2_eval(const float* x, int64_t ldx, int64_t N, int64_t Cin, const int32_t* src_csr,
                         const int32_t* dst_csr, const float* attr_csr, int64_t E, const float* W1,
                         const float* b1, const float* s1, const float* t1, const float* W2,
                         const float* b2, const float* s2, const float* t2, int64_t C, float* H2,
                         int64_t ldh, yolat_stream_t stream);

/* yolat_graph_prepare + yolat_node_uv_eval of the FIRST conv layer in the same launches: the node side of
 * layer 0 reads only the raw node features x (both of its inputs, arch:45), so its GEMM tiles ride in the last,
 * latency-bound pre-processing launch.  Same outputs as the two calls made separately.                     */
int yolat_graph_prepare_node_uv(const int64_t* edge, int64_t stride_e, int64_t stride_c, const float* e_attr,
                                const int64_t* bbox_idx, int64_t E, int64_t N, int64_t P, int32_t* row_ptr,
                                int32_t* perm, int32_t* src_csr, int32_t* dst_csr, float* attr_csr,
                                int32_t* seg_ptr, int32_t* node_seg, int32_t* work, int32_t* status, const float* x,
                                int64_t ldx, int64_t Cin, const float* Wuv, const float* uv_bias, const float* Wr,
                                const float* br,
                                const float* Wn, const float* bn, const float* sn, const float* tn, int64_t C,
                                float* UV, int64_t ld_uv, float* f_out, int64_t ld_fo, float* s_out, int64_t ld_so,
                                yolat_stream_t stream);

/* Factorised edge MLP (eval).  The first edge Linear acts on [x_i | x_j - x_i | attr], so
 *   W1.[x_i | x_j-x_i | attr] = (W1a - W1b).x_i + W1b.x_j + W1c.attr      (torch_vertex.py:331 + torch_nn.py:58)
 * and the two node terms can be computed ONCE PER NODE (N rows) instead of once per edge (E = 4..6 N rows):
 *   yolat_conv_split_w1:   Wuv[0:C] = W1a - W1b, Wuv[C:2C] = W1b   ([2C,Cin]);  Wc4 = W1c  ([C,4])
 *   yolat_node_uv_eval:    UV[N,2C] = f_in . Wuv^T  |  f_out = lin_r(f_in)  |  s_out = relu(BN(mlp_node(s_in)))
 *                          as one launch (the node side of the layer; csr_mean is accumulated into f_out later)
 *   yolat_edge_uv_mlp2_eval: H2[q] = relu(s2*(W2.relu(s1*(U[dst_q] + V[src_q] + Wc4.attr_q + b1) + t1) + b2) + t2)
 * Exact algebra, different summation order than yolat_edge_mlp2_eval (agrees to ~1e-6 relative).  C = 64.   */
int yolat_conv_split_w1(const float* W1, int64_t Cin, int64_t C, float* Wuv, float* Wc4, yolat_stream_t stream);
/* uv_bias (nullable, [2C]): added to UV by the GEMM epilogue — the folded form of the layer puts s1*b1 + t1 there
 * (see yolat_conv_eval.uvb).                                                                                 */
int yolat_node_uv_eval(const float* f_in, int64_t ld_f, const float* s_in, int64_t ld_s, int64_t N, int64_t Cin,
                       const float* Wuv, const float* uv_bias, const float* Wr, const float* br, const float* Wn,
                       const float* bn, const float* sn, const float* tn, int64_t C, float* UV, int64_t ld_uv,
                       float* f_out, int64_t ld_fo, float* s_out, int64_t ld_so, yolat_stream_t stream);
/* Training-forward form of the factorised first edge Linear (no BatchNorm applied):
 *   H1[q] = U[dst_q] + V[src_q] + W1c.attr_q + b1,   UV [N,2C] = x.[W1a-W1b | W1b]^T from a dense Linear
 * (yolat_conv_split_w1 + yolat_linear_fwd).  stats (nullable): BatchNorm partial statistics in the format of
 * yolat_linear_fwd's `stats` (yolat_bn_stats_elems(E, C) floats) for yolat_bn_finalize.  Replaces
 * yolat_edge_lin1_fwd (torch_vertex.py:324,331,335 + nn.0) when E >> N.  C must be 64.                */
int yolat_edge_uv_lin1_fwd(const float* UV, int64_t ld_uv, const int32_t* src_csr, const int32_t* dst_csr,
                           const float* attr_csr, int64_t E, const float* Wc4, const float* b1, int64_t C,
                           float* H1, int64_t ldh, float* stats, yolat_stream_t stream);

/* Backward of the factorised first edge Linear (training; replaces yolat_edge_lin1_bwd_w / _bwd_x /
 * yolat_edge_scatter_bwd when E >> N):  dUV[n, 0:C] = sum of dH1 over the CSR row of n (edges INTO n),
 * dUV[n, C:2C] = sum of dH1 over the CSC column of n (edges OUT OF n; col_ptr / slots of yolat_csc_by_source).
 * Then dWuv = dUV^T.x and dx += dUV.Wuv are N-row dense calls, dWc4 = dH1^T.attr, db1 = column sums of dH1
 * (yolat_linear_bwd_w), and yolat_conv_merge_dw1 maps (dWuv, dWc4) back onto dW1 (inverse of yolat_conv_split_w1).
 * C = 64; deterministic (ascending slot order).                                                               */
int yolat_edge_uv_sums(const float* dH1, int64_t ldh, const int32_t* row_ptr, const int32_t* col_ptr,
                       const int32_t* slots, int64_t N, int64_t C, float* dUV, int64_t ld_uv, yolat_stream_t stream);
int yolat_conv_merge_dw1(const float* dWuv, const float* dWc4, int64_t Cin, int64_t C, float* dW1, int64_t lddw,
                         int accumulate, yolat_stream_t stream);

/* yolat_edge_uv_mlp2_mean_eval: the same edge MLP with the mean aggregation fused in:
 *   f_out[n] += mean_{q in CSR row n} H2[q]       (H2 is never written; f_out already holds lin_r(f_in))
 * b1, (s1, t1), b2, (s2, t2) are each nullable (0 / identity); b1 = s1 = t1 = b2 = NULL is the folded form of
 * yolat_conv_eval.{Wc4f, t2f} and takes the shorter per-edge arithmetic.
 * `variant` selects the kernel (yolat_edge_uv_mlp2_mean_eval = YOLAT_EDGE_AUTO):
 *   YOLAT_EDGE_TILES   one workgroup per tile of destination nodes, layer 2 on fp32-input MFMAs; per-node summation
 *                      in CSR order: bit-identical to yolat_edge_uv_mlp2_eval + yolat_csr_mean_fwd(accumulate)
 *   YOLAT_EDGE_WS_F32  persistent wave-specialised workgroups over edge ranges, same arithmetic in the same order
 *                      (bit-identical to YOLAT_EDGE_TILES)
 *   YOLAT_EDGE_WS_X6   the same structure with layer 2 as an fp32 GEMM emulated on the bf16 matrix cores (three-term
 *                      exact bfloat16 splits of both operands, six products, fp32 accumulation): differs from the
 *                      fp32-MFMA variants by the summation order only (~3e-7 of scale), deterministic
 *   YOLAT_EDGE_AUTO    WS_X6 for E >= 131072 (where the persistent workgroups are amortised), else TILES         */
#define YOLAT_EDGE_AUTO 0
#define YOLAT_EDGE_TILES 1
#define YOLAT_EDGE_WS_F32 2
#define YOLAT_EDGE_WS_X6 3
int yolat_edge_uv_mlp2_mean_eval(const float* UV, int64_t ld_uv, const int32_t* src_csr, const int32_t* dst_csr,
                                 const float* attr_csr, const int32_t* row_ptr, int64_t N, int64_t E,
                                 const float* Wc4, const float* b1, const float* s1, const float* t1,
                                 const float* W2, const float* b2, const float* s2, const float* t2, int64_t C,
                                 float* f_out, int64_t ld_fo, yolat_stream_t stream);
int yolat_edge_uv_mlp2_mean_eval_variant(const float* UV, int64_t ld_uv, const int32_t* src_csr,
                                         const int32_t* dst_csr, const float* attr_csr, const int32_t* row_ptr,
                                         int64_t N, int64_t E, const float* Wc4, const float* b1, const float* s1,
                                         const float* t1, const float* W2, const float* b2, const float* s2,
                                         const float* t2, int64_t C, float* f_out, int64_t ld_fo, int variant,
                                         yolat_stream_t stream);
int yolat_edge_uv_mlp2_eval(const float* UV, int64_t ld_uv, const int32_t* src_csr, const int32_t* dst_csr,
                            const float* attr_csr, int64_t E, const float* Wc4, const float* b1, const float* s1,
                            const float* t1, const float* W2, const float* b2, const float* s2, const float* t2,
                            int64_t C, float* H2, int64_t ldh, yolat_stream_t stream);

/* Node side of an eval-mode conv layer in one launch (torch_vertex.py:324-327), BatchNorm folded:
 *   f_out[n] = mean_{q in [row_ptr[n], row_ptr[n+1])} H2[q]  +  Wr.f_in[n] + br
 *   s_out[n] = relu(sn*(Wn.s_in[n] + bn) + tn)
 * Wr, Wn: [C, Cin] row-major contiguous, C <= 64.  H2 NULL (or E == 0) drops the aggregation term.
 * Same values as yolat_linear_fwd + yolat_csr_mean_fwd(accumulate) + yolat_linear_fwd.                  */
int yolat_node_side_eval(const float* f_in, int64_t ld_f, const float* s_in, int64_t ld_s, int64_t N,
                         int64_t Cin, const float* Wr, const float* br, const float* Wn, const float* bn,
                         const float* sn, const float* tn, const float* H2, int64_t ldh,
                         const int32_t* row_ptr, int64_t E, int64_t C, float* f_out, int64_t ld_fo,
                         float* s_out, int64_t ld_so, yolat_stream_t stream);

/* out[n,0:C] (+)= (1/max(deg,1)) * sum_{q in CSR row n} pro(H)[q,0:C]
 * (= torch_scatter.scatter(reduce='mean', dim_size=N) called by propagate, aggr='mean'
 * torch_vertex.py:308; summation in ascending edge order like the CPU scatter_add).             */
int yolat_csr_mean_fwd(const float* H, int64_t ldh, int64_t C, const float* h_scale,
                       const float* h_shift, int h_relu, const int32_t* row_ptr, int64_t N,
                       float* out, int64_t ldo, int accumulate, yolat_stream_t stream);

/* dM[q,0:C] = dOut[dst_csr[q],0:C] / max(deg(dst),1)    (backward of the mean)                  */
int yolat_csr_mean_bwd(const float* dOut, int64_t lddo, int64_t C, const int32_t* row_ptr,
                       const int32_t* dst_csr, int64_t E, float* dM, int64_t lddm,
                       yolat_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Per-proposal pooling: torch_scatter.scatter(src, bbox_idx, dim=0, reduce='mean'|'max')
 * (architecture3cc_rpn_gp_iter2.py:67,122).  Rows of a proposal are contiguous (seg_ptr).
 * Empty segments give 0 (both modes); max ties: lowest row wins; arg = N for empty.
 * ------------------------------------------------------------------------------------------ */
int yolat_segment_mean_fwd(const float* X, int64_t ldx, int64_t D, const float* x_scale,
                           const float* x_shift, int x_relu, const int32_t* seg_ptr, int64_t P,
                           float* Y, int64_t ldy, yolat_stream_t stream);
int yolat_segment_max_fwd(const float* X, int64_t ldx, int64_t D, const float* x_scale,
                          const float* x_shift, int x_relu, const int32_t* seg_ptr, int64_t P,
                          int64_t N, float* Y, int64_t ldy, int32_t* arg /* nullable [P,D] */,
                          yolat_stream_t stream);
/* Eval-mode fusion stage in one launch (arch:61-69,122): yolat_linear_segmax_fwd over the N nodes together
 * with fusion_block_super (Linear + folded BatchNorm + ReLU) over the P per-proposal means — two independent
 * GEMMs, one flattened grid.  Wf, Wfs: [F, D] contiguous.  `pool` must be zero-filled (yolat_pool_prepare).  */
int yolat_fusion_pair_eval(const float* A, int64_t lda, int64_t N, int64_t D, const float* Wf, const float* bf,
                           const float* sf, const float* tf, int64_t F, const int32_t* node_seg, float* pool,
                           int64_t ldpool, const float* S, int64_t lds, int64_t P, const float* Wfs,
                           const float* bfs, const float* sfs, const float* tfs, float* Ys, int64_t ldys,
                           yolat_stream_t stream);

/* Pooling prologue of the eval forward in one launch, Z = [P, 2(F+D)] (arch:127 layout):
 *   Z[:,0:F] = 0;  Z[:,F:F+D] = segment-max of feats[N,D];  Z[:,2F+D:2F+2D] = segment-mean of fsup[N,D]
 * (fsup may be NULL: the mean part is skipped, e.g. when it is computed on another stream).              */
int yolat_pool_prepare(const float* feats, const float* fsup, int64_t ld, int64_t D, int64_t F,
                       const int32_t* seg_ptr, int64_t P, float* Z, int64_t ldz, yolat_stream_t stream);

/* dX[r,:] = dY[seg(r),:] / max(len(seg),1) */
int yolat_segment_mean_bwd(const float* dY, int64_t lddy, int64_t D, const int32_t* seg_ptr,
                           const int32_t* node_seg, int64_t N, float* dX, int64_t lddx,
                           yolat_stream_t stream);
/* dX[r,c] = (arg[seg(r),c] == r) ? dY[seg(r),c] : 0 */
int yolat_segment_max_bwd(const float* dY, int64_t lddy, int64_t D, const int32_t* arg,
                          const int32_t* node_seg, int64_t N, float* dX, int64_t lddx,
                          yolat_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Training-mode fusion block + per-proposal max pooling without the [N,F] activation
 * (architecture3cc_rpn_gp_iter2.py:61-63,122: MLP([fusion_dims,1024]) -> scatter(reduce='max')) and its
 * autograd.  BatchNorm batch statistics come from the K x K centered Gram matrix of the input, the GEMM
 * epilogue keeps only the per-(proposal, column) extreme pre-activation and its row, the backward is
 * sparse over those P*F entries (derivation: csrc/fusion_train.hip).
 *   A [N,K] (lda % 4 == 0, K % 4 == 0), W [F,K] contiguous, node_seg[N] non-decreasing proposal ids,
 *   Z [P,F] <- max over rows of relu(BN(A W^T + bias)); coef [4,F] <- scale, shift, batch mean, invstd;
 *   saved: fp32 scratch of yolat_fusion_pool_train_saved_elems(K,F,P) kept until the backward;
 *   work: fp32 scratch of yolat_fusion_pool_train_work_elems(N,K,F,P).
 * bwd: gZ [P,F] = dL/dZ; writes dW [F,K], dbias [F] (= 0), dgamma, dbeta [F]; dA [N,K] += dL/dA.
 * ------------------------------------------------------------------------------------------ */
size_t yolat_fusion_pool_train_saved_elems(int64_t K, int64_t F, int64_t P);
size_t yolat_fusion_pool_train_work_elems(int64_t N, int64_t K, int64_t F, int64_t P);
int yolat_fusion_pool_train_fwd(const float* A, int64_t lda, int64_t N, int64_t K, const float* W,
                                const float* bias, int64_t F, const float* gamma, const float* beta,
                                float* running_mean, float* running_var, float momentum, float eps,
                                const int32_t* node_seg, int64_t P, float* Z, int64_t ldz, float* coef,
                                float* saved, float* work, yolat_stream_t stream);
int yolat_fusion_pool_train_bwd(const float* A, int64_t lda, int64_t N, int64_t K, const float* W,
                                const float* gamma, int64_t F, const float* coef, const float* saved,
                                const int32_t* node_seg, const int32_t* seg_ptr, int64_t P, const float* gZ,
                                int64_t ldg, float* dW, float* dbias, float* dgamma, float* dbeta, float* dA,
                                int64_t ldda, float* work, yolat_stream_t stream);
/* the same backward in parts (bit mask) for a caller that overlaps the weight gradient with the input gradient on a
 * second stream: _COLS (dgamma, dbeta and the coefficient vectors in `work` that both other parts read) must be complete
 * before _DW (dW) and _DA (dA +=), which touch disjoint regions of `work`; yolat_fusion_pool_train_bwd = all three      */
#define YOLAT_FUS_BWD_COLS 1
#define YOLAT_FUS_BWD_DW 2
#define YOLAT_FUS_BWD_DA 4
#define YOLAT_FUS_BWD_ALL 7
int yolat_fusion_pool_train_bwd_parts(const float* A, int64_t lda, int64_t N, int64_t K, const float* W,
                                      const float* gamma, int64_t F, const float* coef, const float* saved,
                                      const int32_t* node_seg, const int32_t* seg_ptr, int64_t P, const float* gZ,
                                      int64_t ldg, float* dW, float* dbias, float* dgamma, float* dbeta, float* dA,
                                      int64_t ldda, float* work, int parts, yolat_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Loss and optimiser
 * ------------------------------------------------------------------------------------------ */

/* nn.CrossEntropyLoss() (mean over P) of architecture3cc_rpn_gp_iter2.py:363,376.
 * loss[0] = mean_p( logsumexp(z_p) - z_p[label_p] );  dlogits[P,K] = (softmax - onehot)/P
 * (dlogits nullable).  work: fp32 scratch of yolat_softmax_ce_work_elems(P) elements (per-workgroup
 * partial sums, added in a fixed order -> deterministic); NULL selects a single-workgroup kernel.   */
size_t yolat_softmax_ce_work_elems(int64_t P);
int yolat_softmax_ce(const float* logits, int64_t ld, const int64_t* labels, int64_t P, int64_t K,
                     float* loss, float* dlogits, int64_t lddl, float* work, yolat_stream_t stream);

/* torch.optim.Adam step (train.py:212: lr, weight_decay as L2-in-grad, betas (0.9,0.999),
 * eps 1e-8, no amsgrad) over one flat fp32 buffer of n elements; `step` is the 1-based step.     */
int yolat_adam_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t n,
                    float lr, float beta1, float beta2, float eps, float weight_decay,
                    int64_t step, float grad_scale, yolat_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Whole-model eval forward: SparseCADGCN.forward in eval mode
 * (architecture3cc_rpn_gp_iter2.py:44-71,106-137) as ONE host call that enqueues the complete kernel
 * sequence (graph pre-processing included) on `stream`.  BatchNorm1d is folded: every (s*, t*) pair is
 * the per-channel scale/shift of yolat_bn_eval_coeffs.  All pointers are device pointers; the structs
 * themselves are host memory.
 * ------------------------------------------------------------------------------------------ */
#define YOLAT_MAX_LAYERS 8

typedef struct {
  int64_t Cin;                                /* input channels of this layer                    */
  const float *W1, *b1, *s1, *t1;             /* gconv.nn.0 [C,2Cin+4], gconv.nn.1 folded        */
  const float *W2, *b2, *s2, *t2;             /* gconv.nn.3 [C,C],     gconv.nn.4 folded        */
  const float *Wr, *br;                       /* gconv.lin_r [C,Cin]                             */
  const float *Wn, *bn, *sn, *tn;             /* gconv.mlp_node.0 [C,Cin], mlp_node.1 folded     */
  const float *Wuv, *Wc4;                     /* nullable: yolat_conv_split_w1(W1) outputs, [2C,Cin], [C,4] */
  /* nullable (all four or none): the same layer with nn.1 (BatchNorm, folded s1/t1) and the biases b1, b2 moved
   * out of the per-edge arithmetic — Wuvf = [s1 | s1] (rows) * Wuv, uvb = [s1*b1 + t1 | 0] ([2C], added to UV by the
   * node-side GEMM's epilogue), Wc4f = s1 (rows) * Wc4, t2f = s2*b2 + t2:
   *   h1 = relu(U'[dst] + V'[src] + Wc4f.attr),  message = relu(s2 * (W2.h1) + t2f)                        */
  const float *Wuvf, *uvb, *Wc4f, *t2f;
  /* nullable (all eight or none; Cin == C == 64): the node side prepared for yolat_node_uv_eval_x6 — Wfr_x6 = hi / mid
   * / lo bfloat16 split (yolat_split_bf16x3) of the stacked [Wuvf ; Wr] [192, 64], tfr = [uvb ; br] [192]; Wn_x6 = split
   * of sn (rows) * Wn [64, 64], tn_fold = sn*bn + tn.  Used when N >= YOLAT_NODE_X6_MIN_ROWS.                  */
  const uint16_t *Wfr_x6[3];
  const float *tfr;
  const uint16_t *Wn_x6[3];
  const float *tn_fold;
  /* nullable (both or none; Cin == C == 64, folded form present): the node side of THIS layer prepared for the edge
   * kernel of the PREVIOUS layer (small graphs: one launch per conv layer, csrc/common.hpp EdgeNext) — the stacked
   * W' = [Wuvf ; Wr] [192, 64] in v_mfma_f32_16x16x4_f32 B-fragment order,
   *   Wnx[(ct * 16 + ks) * 64 + l] = W'[ct * 16 + (l & 15)][4 * ks + (l >> 4)],   ct < 12, ks < 16, l < 64,
   * and tnx = [uvb ; br] [192].                                                                                    */
  const float *Wnx, *tnx;
} yolat_conv_eval;

#define YOLAT_CLS_X6_MAX_ROWS 2048
#define YOLAT_CLS1_X6_MIN_ROWS 1024
#define YOLAT_NODE_X6_MIN_ROWS 65536
typedef struct {
  int32_t n_blocks, n_blocks_out, n_classes, reserved;
  int64_t C;                                  /* n_filters (64)                                  */
  int64_t F;                                  /* fusion width (1024)                             */
  int64_t H1, H2;                             /* classifier hidden widths (512, 256)             */
  yolat_conv_eval conv[YOLAT_MAX_LAYERS];     /* head, then backbone[i].body                     */
  const float *Wf, *bf, *sf, *tf;             /* cls_net.fusion_block                            */
  /* nullable (all eight or none): both fusion blocks prepared for the bf16x6-emulated kernel (yolat_fusion_pair_eval_x6) —
   * Wf_hi + Wf_mid + Wf_lo == fl(sf (rows) * Wf) exactly, three bfloat16 [F, D] matrices (yolat_split_bf16x3);
   * tf_fold = sf*bf + tf                                                                                     */
  const uint16_t *Wf_hi, *Wf_mid, *Wf_lo;
  const float *tf_fold;
  const uint16_t *Wfs_hi, *Wfs_mid, *Wfs_lo;  /* the same for fusion_block_super                              */
  const float *tfs_fold;
  const float *Wfs, *bfs, *sfs, *tfs;         /* cls_net.fusion_block_super                      */
  const float *Wc1, *bc1, *sc1, *tc1;         /* prediction_cls.0                                */
  const float *Wc2, *bc2, *sc2, *tc2;         /* prediction_cls.1                                */
  const float *Wc3, *bc3;                     /* prediction_cls.2 (bare Linear)                  */
  /* nullable (all six or none): the three classifier layers prepared for yolat_linear_x6 — Wc_x6[i] =
   * yolat_split_bf16x3_packed of (scale (rows) * Wc{i+1}) (scale = 1 for the bare last layer), tc_fold[i] =
   * scale*b + shift (= b for the last layer).  Used while P <= YOLAT_CLS_X6_MAX_ROWS.                          */
  const uint16_t *Wc_x6[3];
  const float *tc_fold[3];
  /* nullable (both or none): prediction_cls.0 prepared for yolat_gemm_x6 — Wc1_gx = yolat_gemm_x6_pack of
   * (sc1 (rows) * Wc1), tc1_gx = sc1*bc1 + tc1.  Used when P >= YOLAT_CLS1_X6_MIN_ROWS.                          */
  const uint16_t *Wc1_gx;
  const float *tc1_gx;
} yolat_model_eval;

/* workspace (bytes) needed by yolat_forward_eval for a batch of N nodes / E edges / P proposals */
size_t yolat_forward_eval_workspace_bytes(const yolat_model_eval* m, int64_t N, int64_t E, int64_t P);

/* x [N,Cin0] fp32 (ld = ldx); edge int64 with element strides like yolat_coo_to_csr; e_attr [E,4];
 * bbox_idx int64 [N] non-decreasing; logits [P,n_classes] (ld = ld_logits).  `status` as above.   */
int yolat_forward_eval(const yolat_model_eval* m, const float* x, int64_t ldx, const int64_t* edge,
                       int64_t stride_e, int64_t stride_c, const float* e_attr,
                       const int64_t* bbox_idx, int64_t N, int64_t E, int64_t P, float* logits,
                       int64_t ld_logits, void* workspace, size_t workspace_bytes, int32_t* status,
                       yolat_stream_t stream);
/* The same forward for a caller that keeps `workspace` to itself (a serving loop, plan.EvalPlan): the PREVIOUS use of
 * this workspace was yolat_forward_eval / yolat_forward_eval_primed with the same m layout, N, E and P, enqueued on
 * the same stream or already complete, and nothing else has written into it since.  Every forward leaves the
 * CSR-build counters zero, so this call skips their memset launch (3.8 us of the ~120 us cfg-2 forward).  A broken
 * promise gives a wrong CSR (undefined results, memory-safe).                                                       */
int yolat_forward_eval_primed(const yolat_model_eval* m, const float* x, int64_t ldx, const int64_t* edge,
                              int64_t stride_e, int64_t stride_c, const float* e_attr,
                              const int64_t* bbox_idx, int64_t N, int64_t E, int64_t P, float* logits,
                              int64_t ld_logits, void* workspace, size_t workspace_bytes, int32_t* status,
                              yolat_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Batch hand-over as native host code (collate.hip).  Reference: collate (cad_recognition/train.py:
 * 123-171), the edge / bbox_idx offset fix-up loops (train.py:238-258) and the six synchronous .cuda() copies of
 * architecture3cc_rpn_gp_iter2.py:107-115.  All three are HOST functions (no stream, no device pointers).
 * ------------------------------------------------------------------------------------------ */
typedef struct { const void* ptr; int64_t bytes; } yolat_span;
/* dst + field_off[f] receives spans[f*n_items + 0 .. n_items) back to back (torch.cat of one key over the items). */
int yolat_collate_pack(void* dst, const int64_t* field_off, const yolat_span* spans, int64_t n_fields, int64_t n_items);
/* The destination-sorted (CSR) form of ONE dataset item, the host twin of yolat_graph_prepare: same outputs, same
 * clamping of out-of-range ids, same YOLAT_STATUS_* flags (written to the HOST word *status).  A dataset item's graph
 * never changes: computed once per item and cached with it.  bbox_idx == NULL skips the segment outputs.             */
int yolat_item_csr_host(const int64_t* edge, int64_t stride_e, int64_t stride_c, const float* e_attr,
                        const int64_t* bbox_idx, int64_t E, int64_t N, int64_t P, int32_t* row_ptr, int32_t* perm,
                        int32_t* src, int32_t* dst, float* attr, int32_t* seg_ptr, int32_t* node_seg, int32_t* status);
typedef struct {
  int64_t N, E, P;
  const int32_t *row_ptr, *src, *dst;   /* [N+1], [E], [E]  item-local ids  */
  const float* attr;                    /* [E,4] in CSR order               */
  const int32_t *seg_ptr, *node_seg;    /* [P+1], [N]                       */
} yolat_item_csr;
/* The batch's CSR = concatenation of the items' CSRs with the node / edge / proposal offsets added (the adjacency of a
 * collated batch is block diagonal: Datasets/graph_dict3.py:594-600) — bit-identical to yolat_graph_prepare on the
 * collated, fixed-up COO list.  Outputs sized for the batch totals: row_ptr [N+1], src/dst [E], attr [E,4],
 * seg_ptr [P+1], node_seg [N]; typically slices of the pinned staging buffer.                                          */
int yolat_collate_csr_pack(const yolat_item_csr* items, int64_t B, int32_t* row_ptr, int32_t* src, int32_t* dst,
                           float* attr, int32_t* seg_ptr, int32_t* node_seg);

/* The whole csr-mode hand-over in ONE call.  A caller caches one yolat_item_desc per dataset item (pointers to the item's
 * dense arrays — x, pos, bbox, labels ... in a fixed key order — their row counts, and its yolat_item_csr).  The call lays
 * the batch out in `dst` (256-byte aligned fields: key 0 .. n_keys-1, then row_ptr, src, dst, attr, seg_ptr, node_seg;
 * byte offsets in off[n_keys + 6]), writes the collate slices (slices[k*(B+1) + b] = rows of key k before item b,
 * train.py:141-147) and totals = {N, E, P}, copies every key of every item and merges the CSRs.  dst == NULL or
 * cap < *total: only off / total / slices / totals are produced.                                                        */
#define YOLAT_MAX_KEYS 8
typedef struct {
  int64_t n_keys;
  yolat_span key[YOLAT_MAX_KEYS];
  int64_t rows[YOLAT_MAX_KEYS];
  yolat_item_csr csr;                   /* all zero: the batch carries no prepared graph (COO mode, below)               */
  /* COO mode (round 5): keys shipped as raw int64 index tensors get the reference's offset fix-up (train.py:238-258)
   * while they are copied: fix[k] = 1: every element += the rows of key `node_key` in the items before this one (edge:
   * += slices['pos'][i]), 2: += the rows of key `prop_key` before it (bbox_idx: += slices['labels'][i]), 0: copied as is. */
  int32_t fix[YOLAT_MAX_KEYS];
  int32_t node_key, prop_key;
} yolat_item_desc;
int yolat_collate_batch(const yolat_item_desc* const* items, int64_t B, void* dst, int64_t cap, int64_t* off,
                        int64_t* total, int64_t* slices, int64_t* totals);

/* The same hand-over OFF the consumer's thread (loader.hip; the reference hides collate behind DataLoader(num_workers=8)
 * worker processes, cad_recognition/train.py:178-189, and copies six tensors synchronously inside forward(),
 * architecture3cc_rpn_gp_iter2.py:107-115).  One native worker thread per loader runs yolat_collate_batch into a ring of
 * `slots` (2..16) pinned staging buffers and enqueues ONE asynchronous H2D copy per batch on the loader's own stream while the
 * consumer still works on the previous batch.
 *   submit   queue a batch (the pointer list is copied; the descriptors and the arrays they point to must stay alive
 *            until yolat_loader_next has returned this batch).  Batches come out in submission order.
 *   next     blocks (host) until the oldest submitted batch's copy has been ENQUEUED, makes `consumer_stream` wait for
 *            that copy (stream-side) and describes the batch: `device` holds, at byte offsets off[0 .. n_keys + 5], key
 *            0 .. n_keys - 1, then row_ptr, src, dst, attr, seg_ptr, node_seg (yolat_collate_batch's layout); slices as
 *            yolat_collate_batch writes them (valid until the slot is released).  Returns the batch's error code.
 *   release  the consumer has enqueued everything that reads the slot's device buffer on `consumer_stream`: the slot is
 *            rewritten once that work has completed.  Every batch obtained from next must be released.
 * create uses the calling thread's current HIP device.  submit / next / release are meant for ONE consumer thread.   */
typedef struct yolat_loader yolat_loader;
typedef struct {
  void* device;
  int64_t total, n_keys, B, N, E, P;
  int64_t off[YOLAT_MAX_KEYS + 6];
  const int64_t* slices;
  int32_t slot, rc;
} yolat_loader_batch;
yolat_loader* yolat_loader_create(int slots);
int yolat_loader_submit(yolat_loader* loader, const yolat_item_desc* const* items, int64_t B);
int yolat_loader_next(yolat_loader* loader, yolat_stream_t consumer_stream, yolat_loader_batch* out);
int yolat_loader_release(yolat_loader* loader, int slot, yolat_stream_t consumer_stream);
void yolat_loader_destroy(yolat_loader* loader);

/* A prepared graph on the DEVICE (the outputs of yolat_graph_prepare, or the device copy of yolat_collate_csr_pack's
 * staging) for the forwards below: the forward then skips the COO -> CSR conversion (4 launches) and runs the first
 * layer's node side as a launch of its own.  The caller has validated the ids (host status word).                      */
typedef struct {
  const int32_t *row_ptr, *src, *dst;
  const float* attr;
  const int32_t *seg_ptr, *node_seg;
} yolat_graph_csr;
int yolat_forward_eval_csr(const yolat_model_eval* m, const float* x, int64_t ldx, const yolat_graph_csr* g, int64_t N,
                           int64_t E, int64_t P, float* logits, int64_t ld_logits, void* workspace,
                           size_t workspace_bytes, yolat_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * bf16-storage eval forward (bf16_eval.hip): the precision mode of BASELINE.json's large-graph
 * configuration ("N=200k / E=1.2M, n_blocks=4, bf16").  Same contract and kernel sequence as
 * yolat_forward_eval; node activations cross HBM as bfloat16, every Linear with K >= 64 runs on
 * v_mfma_f32_32x32x16_bf16 with fp32 accumulation, bias / folded BatchNorm / ReLU / mean / max stay
 * fp32.  The first layer's K = Cin0 products, the e_attr term, the pooled matrix and the logits are fp32.
 * `base` carries the fp32 parameters (biases, folded BN, first-layer weights, Wuv/Wc4 of every layer are
 * REQUIRED); the bf16 members are yolat_f32_to_bf16 copies of the named fp32 weights (row-major, same
 * shapes).  Wuv/Wr/Wn[0] are unused (layer 0 multiplies the raw fp32 features).
 * Supported shapes: C = 64, Cin0 <= 16, F, C*n_blocks_out, H1, H2 multiples of 64, N <= 2^23, E <= 2^29;
 * else YOLAT_E_UNSUPPORTED.
 * Accuracy: <= 1e-2 of the logits' scale against the fp32 path (tests/test_gpu_bf16.py).
 * ------------------------------------------------------------------------------------------ */
typedef struct {
  const yolat_model_eval* base;
  const uint16_t* Wuv[YOLAT_MAX_LAYERS];      /* bf16 of conv[l].Wuv [2C,Cin]                   */
  const uint16_t* Wr[YOLAT_MAX_LAYERS];       /* bf16 of conv[l].Wr  [C,Cin]                    */
  const uint16_t* Wn[YOLAT_MAX_LAYERS];       /* bf16 of conv[l].Wn  [C,Cin]                    */
  const uint16_t* W2[YOLAT_MAX_LAYERS];       /* bf16 of diag(s2) . conv[l].W2  [C,C]: the scale of the edge MLP's second
                                               * BatchNorm folded into the weight rows BEFORE rounding (ABI 4)   */
  /* layer 1's folded BatchNorm, applied by the node-side GEMM epilogue instead of per edge:
   * uv_scale[l] = [s1 | s1], uv_shift[l] = [s1*b1 + t1 | 0]   (fp32 [2C] each; s1 = 1, t1 = 0 without a norm) */
  const float* uv_scale[YOLAT_MAX_LAYERS];
  const float* uv_shift[YOLAT_MAX_LAYERS];
  const uint16_t *Wf, *Wfs, *Wc1, *Wc2, *Wc3; /* bf16 of the fusion / classifier weights        */
  const float* t2f[YOLAT_MAX_LAYERS];         /* fp32 [C]: s2*b2 + t2, the shift that goes with the folded W2 (ABI 4) */
  /* fusion_block / fusion_block_super with their BatchNorm folded (ABI 4, optional: NULL keeps the unfolded kernels):
   * Wf_fold = bf16(diag(sf) Wf) [F, D], tf_fold = sf*bf + tf [F] fp32; same for the super block */
  const uint16_t *Wf_fold, *Wfs_fold;
  const float *tf_fold, *tfs_fold;
  /* ABI 5, optional (NULL keeps the per-layer launches): the conv stack's weights in the register-fragment order of the
   * one-launch proposal-local kernel (conv_local.hip), yolat_conv_local_pack_bytes(n_blocks) bytes filled by
   * yolat_conv_local_pack from the members above */
  const void* conv_local;
} yolat_model_eval_bf16;

/* All conv layers + the pooling prologue of the bf16-storage forward in ONE launch (conv_local.hip): Backbone.forward,
 * architecture3cc_rpn_gp_iter2.py:44-69 + the segment max / mean of :67,122 over the concat columns, for batches whose
 * edges stay inside their proposal (Datasets/graph_dict3.py:582-600,733) and whose proposals fit a 64-node / 640-edge tile.
 *   yolat_conv_local_pack_bytes / yolat_conv_local_pack: the packed weight image (once per weight version; `m` must carry
 *     every member the bf16 forward needs; C = 64, Cin0 <= 8, else YOLAT_E_UNSUPPORTED).
 *   yolat_conv_stack_local_bf16: on a prepared graph g (yolat_graph_prepare outputs) writes
 *     feats [N, ld_feats] bf16 = the concat of the last n_blocks_out layer outputs, and per proposal p of Z [P, ldz] fp32:
 *     Z[p, 0:F] = 0, Z[p, F:F+D] = max over the proposal's rows of feats, Z[p, 2F+D:2F+2D] = mean of the node branch
 *     (D = C * n_blocks_out) — what the per-layer launches + k_pool_prepare leave behind.
 *     *flag (device int32, zero on entry) is raised to 1 when the batch does not have the property (an edge leaving its
 *     tile, a proposal that does not fit): the outputs are then incomplete and the caller must run the per-layer path;
 *     yolat_forward_eval_bf16 does that by itself (its per-layer launches stay enqueued, gated on the flag).            */
size_t yolat_conv_local_pack_bytes(int64_t n_layers);
int yolat_conv_local_pack(const yolat_model_eval_bf16* m, void* dst, size_t dst_bytes, yolat_stream_t stream);
/* tuning / debug hook of the kernel above (process-wide; tests and tools/exp/conv_local_bench.py): nw = waves per workgroup
 * (4: 64-node tiles, 8: 128-node tiles; 0 = automatic), g0 = proposals per workgroup (0 = automatic), abl = phase-ablation
 * bits (non-zero: results are wrong on purpose), stamps = device buffer of 64 int64 s_memtime stamps per workgroup or NULL */
void yolat_conv_local_tune(int nw, int g0, int abl, long long* stamps);
int yolat_conv_stack_local_bf16(const yolat_model_eval_bf16* m, const void* pack, const float* x, int64_t ldx,
                                const yolat_graph_csr* g, int64_t N, int64_t E, int64_t P, uint16_t* feats,
                                int64_t ld_feats, float* Z, int64_t ldz, int32_t* flag, yolat_stream_t stream);

/* ABI 6.  The locality property of a batch, decided BEFORE the forward is enqueued (integer structure: once per batch
 * version).  The reference stores a proposal's edges as one contiguous block (Datasets/graph_dict3.py:725 `new_edge.append`
 * per proposal, :752-764 per-proposal edge ranges, :732 sorted bbox_idx) and never lets an edge leave its proposal
 * (:582-600,733); a batch like that, whose proposals fit a tile, needs neither the global COO -> CSR build nor the gated
 * per-layer fall-back launches: each tile of the conv kernel sorts its <= 1024 edges by destination in LDS (stable).
 *   yolat_batch_locality: info[4] (device int32) = { YOLAT_LOC_* violations, nodes of the largest proposal, edges of the
 *     largest proposal (by edge RANGES: meaningful when the list is grouped), YOLAT_STATUS_* bits of malformed ids };
 *     workspace of yolat_batch_locality_workspace_bytes.
 *   yolat_locality: the same on the host (known = 1 once examined), handed to yolat_forward_eval_bf16_loc:
 *     known and fitting (yolat_conv_local_fits) -> local prep + ONE conv launch, no gated launches; a violation the device
 *       still finds (the caller's information was stale) raises YOLAT_STATUS_NOT_LOCAL in *status, the logits are invalid;
 *     known and unfit -> the per-layer path directly;   NULL / known = 0 -> found out on the device (gated fall-back).
 *   yolat_conv_stack_local_bf16_coo: yolat_conv_stack_local_bf16 on the raw edge list of a vouched batch (op tests).  */
#define YOLAT_LOC_UNGROUPED 1 /* the edge list is not grouped by proposal (bbox_idx[dst] decreases along it) */
#define YOLAT_LOC_CROSSING  2 /* an edge joins nodes of two proposals                                        */
#define YOLAT_LOC_MALFORMED 4 /* ids out of range / bbox_idx unsorted (the YOLAT_STATUS_* conditions)         */
typedef struct yolat_locality {
  int32_t known, flags, max_nodes, max_edges;
} yolat_locality;
size_t yolat_batch_locality_workspace_bytes(int64_t N, int64_t E, int64_t P);
int yolat_batch_locality(const int64_t* edge, int64_t stride_e, int64_t stride_c, const int64_t* bbox_idx, int64_t N,
                         int64_t E, int64_t P, int32_t* info, void* workspace, size_t workspace_bytes,
                         yolat_stream_t stream);
int yolat_conv_local_fits(const yolat_locality* loc, int64_t P);
/* the record of ONE dataset item, on the host (collate.hip); a batch of items has the OR of the flags and the maxima of the
 * sizes (collate keeps the order and adds per-image offsets, train.py:238-258) */
int yolat_item_locality_host(const int64_t* edge, int64_t stride_e, int64_t stride_c, const int64_t* bbox_idx, int64_t E,
                             int64_t N, int64_t P, yolat_locality* out);
int yolat_conv_stack_local_bf16_coo(const yolat_model_eval_bf16* m, const void* pack, const float* x, int64_t ldx,
                                    const int64_t* edge, int64_t stride_e, int64_t stride_c, const float* e_attr,
                                    const int64_t* bbox_idx, int64_t N, int64_t E, int64_t P, uint16_t* feats,
                                    int64_t ld_feats, float* Z, int64_t ldz, int32_t* flag, int32_t* status,
                                    void* workspace, size_t workspace_bytes, yolat_stream_t stream);

/* The edge stage of the bf16-storage forward on its own (op tests, benchmarks): factorised edge MLP + mean
 * aggregation of one conv layer, gcn_lib/sparse/torch_vertex.py:319-337 in eval mode.
 *   UV [N, ld_uv] bf16: per-node products U' | V' with layer 1's folded BatchNorm applied (node-side epilogue);
 *   src/dst/attr/row_ptr: the CSR (destination-sorted) edge arrays of yolat_graph_prepare;
 *   Wc4 [C,4] fp32 attr columns of the first Linear, s1 [C] their scale (NULL = 1);
 *   W2f [C,C] bf16 = bf16(diag(s2) W2), t2f [C] fp32 = s2*b2 + t2;  root [N, ld_r] fp32 = lin_r(x);
 *   f_out [N, ld_fo] bf16 = bf16(root + mean over in-edges of relu(W2f . relu(U'[dst] + V'[src] + s1*Wc4 . attr) + t2f)).
 * variant: 0 = automatic, 1 = node tiles (k_edge_uv_mlp2_mean_h), 2 = register-chained MFMA waves (edge_chain.hip).
 * C = 64 only. */
int yolat_edge_uv_mlp2_mean_eval_bf16(const uint16_t* UV, int64_t ld_uv, const int32_t* src_csr, const int32_t* dst_csr,
                                      const float* attr_csr, const int32_t* row_ptr, int64_t N, int64_t E,
                                      const float* Wc4, const float* s1, const uint16_t* W2f, const float* t2f,
                                      const float* root, int64_t ld_r, uint16_t* f_out, int64_t ld_fo, int variant,
                                      yolat_stream_t stream);

/* dst[i] = bfloat16(src[i]), round-to-nearest-even (what torch's .to(torch.bfloat16) does); dst 4-byte aligned */
int yolat_f32_to_bf16(const float* src, int64_t n, uint16_t* dst, yolat_stream_t stream);
size_t yolat_forward_eval_bf16_workspace_bytes(const yolat_model_eval_bf16* m, int64_t N, int64_t E, int64_t P);
int yolat_forward_eval_bf16(const yolat_model_eval_bf16* m, const float* x, int64_t ldx, const int64_t* edge,
                            int64_t stride_e, int64_t stride_c, const float* e_attr, const int64_t* bbox_idx,
                            int64_t N, int64_t E, int64_t P, float* logits, int64_t ld_logits, void* workspace,
                            size_t workspace_bytes, int32_t* status, yolat_stream_t stream);
/* yolat_forward_eval_bf16 under the contract of yolat_forward_eval_primed (workspace last used by the same call shape on
 * the same stream: the counter memset is skipped) */
int yolat_forward_eval_bf16_primed(const yolat_model_eval_bf16* m, const float* x, int64_t ldx, const int64_t* edge,
                                   int64_t stride_e, int64_t stride_c, const float* e_attr, const int64_t* bbox_idx,
                                   int64_t N, int64_t E, int64_t P, float* logits, int64_t ld_logits, void* workspace,
                                   size_t workspace_bytes, int32_t* status, yolat_stream_t stream);
/* yolat_forward_eval_bf16 with the batch's locality decided by the caller (yolat_locality above; NULL = unknown) and the
 * `primed` promise as a flag; g != NULL: the prepared-graph form (edge / e_attr / bbox_idx unused) */
int yolat_forward_eval_bf16_loc(const yolat_model_eval_bf16* m, const float* x, int64_t ldx, const int64_t* edge,
                                int64_t stride_e, int64_t stride_c, const float* e_attr, const int64_t* bbox_idx,
                                const yolat_graph_csr* g, int64_t N, int64_t E, int64_t P, float* logits,
                                int64_t ld_logits, void* workspace, size_t workspace_bytes, int32_t* status,
                                const yolat_locality* loc, int primed, yolat_stream_t stream);
/* the same forward on a prepared device graph (yolat_graph_csr above) */
int yolat_forward_eval_bf16_csr(const yolat_model_eval_bf16* m, const float* x, int64_t ldx, const yolat_graph_csr* g,
                                int64_t N, int64_t E, int64_t P, float* logits, int64_t ld_logits, void* workspace,
                                size_t workspace_bytes, yolat_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * ABI 6.  The training step as ONE native call (train_plan.hip): the loop body of cad_recognition/train.py:263-284 —
 * forward (architecture3cc_rpn_gp_iter2.py:106-137), CrossEntropyLoss (arch:363), backward, Adam (train.py:212) — enqueued
 * from C on `stream` (+ the weight gradients and node branches on `side_stream`, NULL = one stream): the schedule engine.py
 * issues from Python, the same kernels on the same operands in the same order per stream, so the results are bit-identical.
 *   yolat_train_model: pointers into ONE flat parameter buffer (param_base) and its gradient twin (grad_base, same
 *     offsets: trainer.FlatParams); BatchNorm running statistics / counters are updated in place.
 *   Shapes: n_filters C = 64, n_blocks_out = 2, biases and BatchNorm on every layer, no dropout, E >= N; `half` != 0:
 *     bfloat16 storage of the per-edge tensors where E >= 2 N.  Anything else: YOLAT_E_UNSUPPORTED (the caller keeps its
 *     own schedule).
 *   Batch: the collated COO arrays (edge / e_attr / bbox_idx; the destination-sorted form is built inside the call), or a
 *     prepared graph g.  labels [P] int64.  logits [P, ld_logits] and loss [1] are written; *status as yolat_graph_prepare.
 *   phases (bit mask): 1 = graph + forward + loss + backward of the classifier and the fusion blocks — on return every
 *     gradient of the parameters BEHIND the conv layers (the "head" bucket of the data-parallel exchange) is final on
 *     `stream`; 2 = backward of the conv layers (all gradients final on `stream`); 4 = Adam (yolat_adam_args).  A caller that
 *     exchanges gradients issues its collectives between the calls; 7 = the whole step.  The workspace
 *     (yolat_train_step_workspace_bytes, 256-byte aligned) carries the step's state between the calls.                  */
typedef struct yolat_train_lin { const float* W; const float* b; } yolat_train_lin;
typedef struct yolat_train_bn {
  const float* gamma; const float* beta; float* running_mean; float* running_var; int64_t* num_batches_tracked;
  float momentum, eps;
} yolat_train_bn;
typedef struct yolat_train_conv {
  int64_t Cin;
  yolat_train_lin nn0; yolat_train_bn bn1; yolat_train_lin nn3; yolat_train_bn bn4;   /* gconv.nn.{0,1,3,4}   */
  yolat_train_lin lin_r; yolat_train_lin node; yolat_train_bn bn_node;                /* lin_r, mlp_node.{0,1} */
} yolat_train_conv;
typedef struct yolat_train_model {
  int32_t n_blocks, n_blocks_out, n_classes, half;
  int64_t C, F, H1, H2;
  yolat_train_conv conv[YOLAT_MAX_LAYERS];
  yolat_train_lin fus; yolat_train_bn fus_bn; yolat_train_lin fus_s; yolat_train_bn fus_s_bn;
  yolat_train_lin c1; yolat_train_bn c1_bn; yolat_train_lin c2; yolat_train_bn c2_bn; yolat_train_lin c3;
  const float* param_base; float* grad_base;
} yolat_train_model;
typedef struct yolat_adam_args {
  float* exp_avg; float* exp_avg_sq; int64_t n; int64_t step;
  float lr, beta1, beta2, eps, weight_decay, grad_scale;
} yolat_adam_args;
size_t yolat_train_step_workspace_bytes(const yolat_train_model* m, int64_t N, int64_t E, int64_t P);
int yolat_train_step(const yolat_train_model* m, const float* x, int64_t ldx, const int64_t* edge, int64_t stride_e,
                     int64_t stride_c, const float* e_attr, const int64_t* bbox_idx, const yolat_graph_csr* g,
                     const int64_t* labels, int64_t N, int64_t E, int64_t P, float* logits, int64_t ld_logits, float* loss,
                     void* workspace, size_t workspace_bytes, int32_t* status, const yolat_adam_args* adam, int phases,
                     yolat_stream_t stream, yolat_stream_t side_stream);

/* ------------------------------------------------------------------------------------------
 * ABI 6.  predict() in one submission (subgraph.hip): SparseCADGCN.predict, architecture3cc_rpn_gp_iter2.py:139-356.
 * In eval mode every proposal's logits are a row of ONE forward over the whole batch (a proposal sees only its own nodes
 * and edges), so the root pass / has_object / child pass of the reference (:259-281) reduce to integer selection on the
 * device: no host round trip between the passes, one read at the end.
 *   yolat_predict_tree: the proposal tree of the batch, flattened on the host once (device int32 arrays): per root its
 *     global proposal row and (idx_pos start, end, idx_edge start, end) as global ranges; child_ptr [R + 1] into the same
 *     for the children; image_root_ptr [B + 1] = first root of every image.
 *   yolat_predict_select: out[0] = rows of the result, out[1] = 0 when the tree's ranges ARE its proposals' node / edge
 *     ranges and no edge leaves its proposal (else != 0: the one-forward shortcut does not apply — duplicated or foreign
 *     ranges, the reference's KeyError cases — and the caller runs the two-pass extraction), out[4 .. 4 + B] =
 *     slice_image_bbox, out[4 + B + 1 ..] = the proposal row of every output row (slice_bbox): per image the roots, then
 *     the children of the roots whose arg-max (first maximum) is class K - 1.  out: 4 + B + 1 + R + Ctot int32.
 *   yolat_predict_gather: out_cls [total, K] = logits[rows], out_bbox [total, 4] = the rows' boxes enlarged by 5 % about
 *     their centre with the reference's fp32 steps (:341-346).                                                           */
typedef struct yolat_predict_tree {
  int64_t R, Ctot, B;
  const int32_t *root_row, *root_range;      /* [R], [R, 4]       */
  const int32_t *child_ptr;                  /* [R + 1]           */
  const int32_t *child_row, *child_range;    /* [Ctot], [Ctot, 4] */
  const int32_t *image_root_ptr;             /* [B + 1]           */
} yolat_predict_tree;
size_t yolat_predict_select_workspace_bytes(int64_t N, int64_t P, int64_t R);
int yolat_predict_select(const float* logits, int64_t ld_logits, int64_t P, int64_t K, const int64_t* edge, int64_t stride_e,
                         int64_t stride_c, const int64_t* bbox_idx, int64_t N, int64_t E, const yolat_predict_tree* tree,
                         int32_t* out, void* workspace, size_t workspace_bytes, yolat_stream_t stream);
int yolat_predict_gather(const float* logits, int64_t ld_logits, int64_t P, int64_t K, const float* bbox, const int32_t* rows,
                         int64_t total, float* out_cls, float* out_bbox, yolat_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Post-processing: torchvision.ops.nms(boxes, scores, iou_threshold) as called by the
 * reference's non_max_suppression (cad_recognition/train.py:34-121 at :105; detect.py:118).
 * boxes [n,4] fp32 (x1,y1,x2,y2; 16-byte aligned), scores [n] fp32.  keep[0 .. *num_keep) = indices of the kept
 * boxes in descending score order (ties: ascending index); greedy: a box is dropped when its IoU with a kept,
 * higher-scored box is > iou_threshold (fp32 arithmetic as in torchvision's kernels).  keep has room for n
 * entries; num_keep is a DEVICE int32.  work: yolat_nms_work_bytes(n) bytes, 256-byte aligned.
 * n <= 524288 (the reference caps at max_nms = 30000, train.py:47), else YOLAT_E_UNSUPPORTED.
 * ------------------------------------------------------------------------------------------ */
size_t yolat_nms_work_bytes(int64_t n);
int yolat_nms(const float* boxes, const float* scores, int64_t n, float iou_threshold, int64_t* keep,
              int32_t* num_keep, void* work, size_t work_bytes, yolat_stream_t stream);

/* Stage profiler of yolat_forward_eval: when enabled, a hipEvent pair is recorded on `stream` around
 * every stage (each stage = the launch(es) of one kernel family); totals accumulate across calls until
 * reset.  yolat_profile_get must be called after the stream has been synchronised.  `flops` / `bytes`
 * are the ALGORITHMIC work of one call of the stage.                                */
int yolat_profile_enable(int on);
int yolat_profile_enabled(void);
int yolat_profile_reset(void);
int yolat_profile_count(void);
int yolat_profile_get(int index, char* name, int name_capacity, float* total_ms, int* calls,
                      double* flops, double* bytes);

/* Eval fusion block + per-proposal max with the [N,D] x [D,F] GEMM emulated on the bf16 matrix cores (fusion_x6.hip):
 * both operands split exactly into three bfloat16 terms, six v_mfma_f32_32x32x16_bf16 products per 16 k, fp32
 * accumulation — the fp32 result up to the summation order (~1e-7 relative) at 2.7x the fp32-MFMA rate.
 *   yolat_split_bf16x3         hi + mid + lo == fl(row_scale[r] * W[r, c]) exactly (row_scale nullable)
 *   yolat_fusion_pair_eval_x6  same contract as yolat_fusion_pair_eval with both fusion blocks given as (hi, mid,
 *                              lo) = split of s (.) W and the folded shift s*b + t;  D in {64, 128}, F % 64 == 0   */
int yolat_split_bf16x3(const float* W, int64_t ldw, int64_t rows, int64_t cols, const float* row_scale, uint16_t* hi,
                       uint16_t* mid, uint16_t* lo, yolat_stream_t stream);
/* out [M, N] = act(A [M, K] . W^T + shift) with W given as its exact 3-way bfloat16 split, packed in MFMA operand
 * order by yolat_split_bf16x3_packed (any row scale already inside): the skinny, long-K classifier layers as a
 * bf16x6-emulated fp32 GEMM (fp32 accumulate, ~3e-7 relative to the fp32 product; deterministic).  shift NULL: none;
 * relu != 0: ReLU.  K % 16 == 0, lda % 4 == 0, A and Wp 16-byte aligned.  Replaces nn.Linear + BatchNorm1d(eval) +
 * ReLU of prediction_cls (architecture3cc_rpn_gp_iter2.py:91-93,127-128) for a few hundred rows.               */
/* Backward of  out[n] += mean_{e in CSR row n} relu(BatchNorm_train(Y[e]))  without materialising the [E, C] gradients
 * (bn_csr.hip; the aggregation of AttrRelativeEdgeConvGlobalPool2 on top of nn.4 / nn.5, torch_vertex.py:308,324,
 * 333-335): the gradient w.r.t. Y,  dY[e] = scale * (g - c1 - xhat * c2),  g = relu'(.) * d_out[dst[e]] * inv_deg[dst[e]],
 * is formed in the loaders of its consumers.  Call order per layer: yolat_bn_csr_bwd_stats (fills coef), then
 * yolat_linear_bwd_w_csr and yolat_linear_fwd_wt_csr with g->coef pointing at that buffer.                        */
typedef struct {
  const float* d_out; int64_t ld_out;   /* [N, C] gradient w.r.t. the aggregated output                        */
  const int32_t* dst;                   /* [E] destination node of every CSR slot                              */
  const float* inv_deg;                 /* [N] 1 / max(in-degree, 1)                                           */
  const void* Y; int64_t ldy;           /* [E, C] BatchNorm input (the second edge Linear's pre-activation): fp32,
                                         * or bfloat16-stored when `half` != 0                                  */
  const float *mean, *invstd;           /* [C] saved batch statistics                                          */
  const float *scale, *shift;           /* [C] gamma * invstd, beta - mean * scale                             */
  const float* coef;                    /* [2C] (c1 | c2) written by yolat_bn_csr_bwd_stats                    */
  int32_t relu;
  int32_t half;                         /* bfloat16 storage of Y (and of A / dA in yolat_bn_csr_l2_bwd); the loader-
                                         * fused GEMM entry points are fp32 only                                 */
} yolat_bn_csr_grad;
int yolat_inv_degree(const int32_t* row_ptr, int64_t N, float* inv_deg, yolat_stream_t stream);
size_t yolat_bn_csr_work_elems(int64_t E, int64_t C);
int yolat_bn_csr_bwd_stats(const yolat_bn_csr_grad* g, int64_t E, int64_t C, float* dgamma, float* dbeta, int accumulate,
                           float* coef_out, float* work, yolat_stream_t stream);
int yolat_linear_bwd_w_csr(const yolat_bn_csr_grad* g, int64_t E, int64_t C, const float* A, int64_t lda, int64_t K,
                           const float* a_scale, const float* a_shift, int a_relu, float* dW, int64_t lddw, float* db,
                           int accumulate, float* partial, yolat_stream_t stream);
int yolat_linear_fwd_wt_csr(const yolat_bn_csr_grad* g, int64_t E, int64_t C, const float* W, int64_t ldw, int64_t Nout,
                            float* dA, int64_t ldda, yolat_stream_t stream);
/* both of the above in ONE kernel for C = K = Nout = 64 (the edge MLP's second Linear): dY tiles formed once.  work:
 * yolat_bn_csr_l2_bwd_work_elems() floats.                                                                     */
size_t yolat_bn_csr_l2_bwd_work_elems(void);
/* next_mean != NULL: also the statistics of the BatchNorm in front of A (a_scale / a_shift / a_relu describe it) for ITS
 * backward on dA: next_dgamma / next_dbeta [64], next_coef [128] = (c1 | c2) for yolat_bn_relu_bwd_apply.             */
int yolat_bn_csr_l2_bwd(const yolat_bn_csr_grad* g, int64_t E, const void* A, int64_t lda, const float* a_scale,
                        const float* a_shift, int a_relu, const float* W, int64_t ldw, float* dW, int64_t lddw, float* db,
                        int accumulate, void* dA, int64_t ldda, float* work, const float* next_mean,
                        const float* next_invstd, float* next_dgamma, float* next_dbeta, float* next_coef,
                        yolat_stream_t stream);
/* the apply pass of yolat_bn_relu_bwd alone (coefficients given); half != 0: bfloat16-stored dZ / Y / dY           */
int yolat_bn_relu_bwd_apply(const void* dZ, int64_t lddz, const void* Y, int64_t ldy, int64_t M, int64_t C,
                            const float* save_mean, const float* save_invstd, const float* scale, const float* shift,
                            int relu, const float* coef, void* dY, int64_t lddy, int half, yolat_stream_t stream);

/* Training backward of the factorised first edge Linear, the part that reads the edge attributes (torch_vertex.py:331,
 * gconv.nn.0 restricted to its 4 attr inputs):  dWc4 [C, 4] = dH1^T . attr,  db1 [C] = column sums of dH1 (NULL: skipped).
 * dH1 [E, C] fp32 (half = 0) or bfloat16 (half != 0), rows in CSR order like attr_csr [E, 4].  One streaming pass + a
 * fixed-order reduction of per-workgroup partials (deterministic); `work`: yolat_edge_attr_dw_work_elems(E) floats.
 * C == 64, ldh % 4 == 0, dH1 16-byte (bf16: 8-byte) aligned.                                                        */
size_t yolat_edge_attr_dw_work_elems(int64_t E);
int yolat_edge_attr_dw(const void* dH1, int64_t ldh, int half, const float* attr_csr, int64_t E, int64_t C, float* dWc4,
                       float* db1, float* work, yolat_stream_t stream);

/* yolat_bn_apply_edge_sums (round 4): the apply pass of the BatchNorm + ReLU backward in front of the factorised first
 * edge Linear (torch_vertex.py:331-332 nn.1 / nn.2; yolat_bn_relu_bwd_apply) fused with the consumers of its result that
 * read it in CSR order — the dU half of yolat_edge_uv_sums and yolat_edge_attr_dw:
 *   dH1[q] = scale (relu'(.) dA1[q] - c1 - xhat[q] c2)   stored (dH1 may alias dA1; the dV gather still reads it)
 *   dUV[n, 0:64] = sum over CSR row n of dH1[q] (ascending q),  dWc4 [64, 4] = dH1^T . attr_csr,  db1 [64] = colsum (nullable)
 * dA1 / H1 / dH1 [E, 64] fp32 (half = 0) or bfloat16 (half != 0: the sums take the rounded, stored values); coef [128] =
 * (c1 | c2) as yolat_bn_csr_l2_bwd's next_coef.  Deterministic.  work: yolat_bn_apply_edge_sums_work_elems(N) floats.
 * yolat_edge_uv_sums_v: the dV half alone, dUV[n, 64:128] = sum over the CSC column of n of dH1[slots[t]].         */
size_t yolat_bn_apply_edge_sums_work_elems(int64_t N);
int yolat_bn_apply_edge_sums(const void* dA1, int64_t ldda, const void* H1, int64_t ldh, void* dH1, int64_t lddh, int half,
                             int64_t E, const float* save_mean, const float* save_invstd, const float* scale,
                             const float* shift, int relu, const float* coef, const int32_t* row_ptr,
                             const float* attr_csr, int64_t N, float* dUV, int64_t ld_uv, float* dWc4, float* db1,
                             float* work, yolat_stream_t stream);
int yolat_edge_uv_sums_v(const void* dH1, int64_t ldh, int half, const int32_t* col_ptr, const int32_t* slots, int64_t N,
                         int64_t C, float* dUV, int64_t ld_uv, yolat_stream_t stream);

/* LDS-tiled bf16x6-emulated fp32 GEMM (gemm_x6.hip): out [M, N] = act(A [M, K] . W'^T + shift), W' = row_scale (rows)
 * * W packed once per weight version by yolat_gemm_x6_pack (yolat_gemm_x6_packed_elems(N, K) bfloat16 values).
 * ~3e-7 relative to the fp32 product, deterministic (few rows: K is split over workgroups and the fp32 partials are
 * summed in a fixed order; `work`: yolat_gemm_x6_work_elems(M, N, K) floats, NULL allowed when that is 0).
 * K % 16 == 0, lda % 4 == 0, A / packed 16-byte aligned.  Replaces nn.Linear + BatchNorm1d(eval) + ReLU of
 * prediction_cls.0 (architecture3cc_rpn_gp_iter2.py:91,127).                                                  */
size_t yolat_gemm_x6_packed_elems(int64_t N, int64_t K);
int yolat_gemm_x6_pack(const float* W, int64_t ldw, int64_t N, int64_t K, const float* row_scale, uint16_t* packed,
                       yolat_stream_t stream);
/* the weight given transposed, Wt [K, N] row-major: packs it for dX [M, N] = dY [M, K] . Wt (backward of a Linear)   */
int yolat_gemm_x6_pack_t(const float* Wt, int64_t ldw, int64_t N, int64_t K, uint16_t* packed, yolat_stream_t stream);
size_t yolat_gemm_x6_work_elems(int64_t M, int64_t N, int64_t K);
int yolat_gemm_x6(const float* A, int64_t lda, int64_t M, int64_t K, const uint16_t* Wp, const float* shift, int relu,
                  int64_t N, float* out, int64_t ldo, float* work, yolat_stream_t stream);
/* yolat_gemm_x6 with bias only (pre-activation output) + the BatchNorm partial statistics of the output in
 * yolat_linear_fwd's `stats` layout: the training-mode Linear in front of a BatchNorm; shapes without K split only    */
int yolat_gemm_x6_stats(const float* A, int64_t lda, int64_t M, int64_t K, const uint16_t* Wp, const float* bias, int64_t N,
                        float* out, int64_t ldo, float* stats, yolat_stream_t stream);
size_t yolat_split_bf16x3_packed_elems(int64_t N, int64_t K);
int yolat_split_bf16x3_packed(const float* W, int64_t ldw, int64_t N, int64_t K, const float* row_scale,
                              uint16_t* packed, yolat_stream_t stream);
int yolat_linear_x6(const float* A, int64_t lda, int64_t M, int64_t K, const uint16_t* Wp, const float* shift, int relu,
                    int64_t N, float* out, int64_t ldo, yolat_stream_t stream);
/* the same with A given pre-split: Ap = yolat_split_bf16x3_packed(A, lda, M, K, NULL) — for long K                */
int yolat_linear_x6_pre(const uint16_t* Ap, int64_t M, int64_t K, const uint16_t* Wp, const float* shift, int relu,
                        int64_t N, float* out, int64_t ldo, yolat_stream_t stream);
/* Node side of a factorised conv layer (eval, Cin = C = 64) on the bf16x6 rows kernel: UV [N,128] = f_in . Wuv'^T + uvb,
 * f_out [N,64] = f_in . Wr^T + br (stacked weights Wfr [192,64] pre-split, shifts tfr [192]) and s_out [N,64] =
 * relu(s_in . (sn (.) Wn)^T + tn_fold) in one launch; same contract as yolat_node_uv_eval with the folded weights.   */
int yolat_node_uv_eval_x6(const float* f_in, int64_t ld_f, const float* s_in, int64_t ld_s, int64_t N,
                          const uint16_t* Wfr_h, const uint16_t* Wfr_m, const uint16_t* Wfr_l, const float* tfr,
                          const uint16_t* Wn_h, const uint16_t* Wn_m, const uint16_t* Wn_l, const float* tn_fold,
                          float* UV, int64_t ld_uv, float* f_out, int64_t ld_fo, float* s_out, int64_t ld_so,
                          yolat_stream_t stream);
/* Training-mode Linear for many rows on the bf16x6 rows kernel: Y [M, Nout] = pro(A) . W^T + bias (pre-activation; bias
 * required), optional BatchNorm+ReLU prologue on A, optional BatchNorm partial statistics of Y (same layout as
 * yolat_linear_fwd's `stats`).  K in {64, 128}, Nout % 64 == 0, lda % 4 == 0; wsplit: 3 * Nout * K bfloat16 values
 * (the weight is split on every call).  Replaces yolat_linear_fwd for the second edge Linear of a training conv layer. */
int yolat_linear_fwd_rows_x6(const float* A, int64_t lda, int64_t M, int64_t K, const float* a_scale, const float* a_shift,
                             int a_relu, const float* W, int64_t ldw, const float* bias, int64_t Nout, float* Y,
                             int64_t ldy, float* stats, uint16_t* wsplit, yolat_stream_t stream);
int yolat_fusion_pair_eval_x6(const float* A, int64_t lda, int64_t N, int64_t D, const uint16_t* Wh, const uint16_t* Wm,
                              const uint16_t* Wl, const float* tfold, int64_t F, const int32_t* node_seg, float* pool,
                              int64_t ldpool, const float* S, int64_t lds, int64_t P, const uint16_t* Wsh,
                              const uint16_t* Wsm, const uint16_t* Wsl, const float* tsfold, float* Ys, int64_t ldys,
                              yolat_stream_t stream);

/* nn.Dropout2d(p) of MLP in training mode (gcn_lib/sparse/torch_nn.py:67-68; only prediction_cls.1 can carry it,
 * architecture3cc_rpn_gp_iter2.py:92) on a [M,C] activation, with the producer's lazy BatchNorm + ReLU applied on the
 * way:  Z = relu?(Y*scale + shift) * keep / (1-p),  keep ~ Bernoulli(1-p) per ELEMENT (what torch 1.7.1's
 * feature_dropout does for a 2-D input), a pure function of (seed, element index); `mask` (uint8 [M*C]) is written
 * for yolat_dropout_bwd:  dX = dZ * keep / (1-p).                                                            */
int yolat_dropout_fwd(const float* Y, int64_t ldy, int64_t M, int64_t C, const float* scale, const float* shift,
                      int relu, float p, uint64_t seed, uint8_t* mask, float* Z, int64_t ldz, yolat_stream_t stream);
int yolat_dropout_bwd(const float* dZ, int64_t lddz, int64_t M, int64_t C, const uint8_t* mask, float p, float* dX,
                      int64_t lddx, yolat_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Training with bfloat16 STORAGE of the [E,*] tensors (the precision mode BASELINE.json configs[4] names, applied to
 * the train step: cad_recognition/train.py:263-284): the per-edge activations H1, H2 and their gradients cross HBM
 * as bfloat16 (round-to-nearest-even on store, exact widening on load); every accumulation, the BatchNorm
 * statistics, the parameters, their gradients and the optimizer state stay fp32.  `uint16_t*` = bfloat16 bits.
 * Same contracts as the fp32 entry points they shadow; rows 8-byte aligned, leading dimensions multiples of 4.
 *   yolat_edge_uv_lin1_fwd_h   H1 (bf16) + BatchNorm partial statistics of the fp32 values   (yolat_edge_uv_lin1_fwd)
 *   yolat_linear_fwd_h         Y (bf16) = pro(A bf16).W^T + bias, + statistics, on v_mfma_f32_32x32x16_bf16 (K % 64
 *                              == 0; W converted to bf16 into w_work [Nout*K] first)          (yolat_linear_fwd)
 *   yolat_csr_mean_fwd_h       out (fp32) (+)= mean over CSR rows of pro(H bf16)              (yolat_csr_mean_fwd)
 *   yolat_csr_mean_bwd_h       dM (bf16)[q] = dOut[dst_q] / deg                               (yolat_csr_mean_bwd)
 *   yolat_bn_relu_bwd_h        dY (bf16) from dZ, Y (bf16); dgamma / dbeta fp32                (yolat_bn_relu_bwd)
 *   yolat_linear_bwd_w_h       dW, db (fp32) = dY(bf16)^T . pro(A), A bf16 or fp32            (yolat_linear_bwd_w)
 *   yolat_linear_fwd_wt_h      Y (bf16) = A (bf16) . Wt, bf16 MFMAs, w_work as above          (yolat_linear_fwd_wt)
 *   yolat_edge_uv_sums_h       dUV (fp32) = per-node CSR / CSC sums of dH1 (bf16)             (yolat_edge_uv_sums)
 * ------------------------------------------------------------------------------------------ */
int yolat_edge_uv_lin1_fwd_h(const float* UV, int64_t ld_uv, const int32_t* src_csr, const int32_t* dst_csr,
                             const float* attr_csr, int64_t E, const float* Wc4, const float* b1, int64_t C,
                             uint16_t* H1, int64_t ldh, float* stats, yolat_stream_t stream);
int yolat_linear_fwd_h(const uint16_t* A, int64_t lda, int64_t M, int64_t K, const float* a_scale,
                       const float* a_shift, int a_relu, const float* W, int64_t ldw, const float* bias, int64_t Nout,
                       uint16_t* Y, int64_t ldy, float* stats, uint16_t* w_work, yolat_stream_t stream);
int yolat_csr_mean_fwd_h(const uint16_t* H, int64_t ldh, int64_t C, const float* h_scale, const float* h_shift,
                         int h_relu, const int32_t* row_ptr, int64_t N, float* out, int64_t ldo, int accumulate,
                         yolat_stream_t stream);
int yolat_csr_mean_bwd_h(const float* dOut, int64_t lddo, int64_t C, const int32_t* row_ptr, const int32_t* dst_csr,
                         int64_t E, uint16_t* dM, int64_t lddm, yolat_stream_t stream);
int yolat_bn_relu_bwd_h(const uint16_t* dZ, int64_t lddz, const uint16_t* Y, int64_t ldy, int64_t M, int64_t C,
                        const float* save_mean, const float* save_invstd, const float* scale, const float* shift,
                        int relu, float* dgamma, float* dbeta, int accumulate, uint16_t* dY, int64_t lddy, float* work,
                        yolat_stream_t stream);
int yolat_linear_bwd_w_h(const uint16_t* dY, int64_t lddy, int64_t M, int64_t Nout, const void* A, int a_is_half,
                         int64_t lda, int64_t K, const float* a_scale, const float* a_shift, int a_relu, float* dW,
                         int64_t lddw, float* db, int accumulate, float* partial, yolat_stream_t stream);
int yolat_linear_fwd_wt_h(const uint16_t* A, int64_t lda, int64_t M, int64_t K, const float* Wt, int64_t ldw,
                          int64_t Nout, uint16_t* Y, int64_t ldy, uint16_t* w_work, yolat_stream_t stream);
int yolat_edge_uv_sums_h(const uint16_t* dH1, int64_t ldh, const int32_t* row_ptr, const int32_t* col_ptr,
                         const int32_t* slots, int64_t N, int64_t C, float* dUV, int64_t ld_uv, yolat_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Box-proposal generation (dataset side): the integer core of
 * SESYDFloorPlan._get_proposal, /root/reference/Datasets/graph_dict3.py:309-789 — HOST code (proposals.hip), it
 * runs in DataLoader workers and its result is cached per SVG (:924-929).
 *   pos [n_nodes,2] float64: positions of the non-control nodes; (cc_ptr, cc_idx): the connected components as
 *   CSR lists of node ids (graph_dict['cc'] after the control-point renumbering :329-345); edge / edge_super
 *   [.,2] int64 node ids; bbox_sampling_step as in the dataset (10 Floorplans / 5 Diagrams).
 * For every component: distinct-coordinate grid (:392-404), sampling-grid windows with the reference's endpoint
 * scans (:471-528), point set of every window (:541-549), de-duplication (:557), then per sub-cluster the edges
 * with both end points inside in the reference's pick-up order (:582-613) and the rejection tests
 * (no edge :597, box thinner than 1e-4 :621, no node with two neighbours :681).
 * Proposals are emitted component by component, inside a component in lexicographic order of the sorted node-id
 * tuple (the reference iterates a Python set: hash order).
 * Errors: YOLAT_E_UNSUPPORTED for a component of zero width or height (numpy.arange raises ZeroDivisionError
 * there, :462-463).                                                                                           */
typedef struct yolat_proposals yolat_proposals;
int yolat_proposals_build(const double* pos, int64_t n_nodes, const int64_t* cc_ptr, const int64_t* cc_idx,
                          int64_t n_cc, const int64_t* edge, int64_t n_edge, const int64_t* edge_super,
                          int64_t n_edge_super, double bbox_sampling_step, yolat_proposals** out);
int64_t yolat_proposals_count(const yolat_proposals* p);
int64_t yolat_proposals_total(const yolat_proposals* p, int what);   /* 0 nodes, 1 edges, 2 super edges */
/* node_ptr / edge_ptr / sedge_ptr [count+1]; node_idx / edge_idx / sedge_idx [totals]; cc_of [count];
 * bbox [count,4] = min_x, min_y, max_x, max_y of the member points (index arrays may be NULL to skip them)   */
int yolat_proposals_get(const yolat_proposals* p, int64_t* node_ptr, int64_t* node_idx, int64_t* edge_ptr,
                        int64_t* edge_idx, int64_t* sedge_ptr, int64_t* sedge_idx, int64_t* cc_of, double* bbox);
int yolat_proposals_window_counts(const yolat_proposals* p, int64_t* windows, int64_t* distinct);   /* [n_cc] each */
/* ABI 6: the per-proposal assembly of _get_proposal (graph_dict3.py:577-753) on the handle's member lists — local
 * re-indexing (+ running node offset), e_attr rows, label / regression target / has_obj by IoU / IoS against the
 * ground-truth boxes of the proposal's component (valid_ptr [n_cc + 1] / valid_idx: utils/det_util.py:343-362), the 13
 * statistics of :644-705 (stat_feats = 0: zeros), positions normalised to the box (normalize != 0, :714).  float64 in the
 * reference's operation order; outputs sized by yolat_proposals_count / _total: new_pos [nodes, 2], new_is_super
 * [nodes, sw], new_edge [edges, 2], new_e_attr [edges, aw], new_edge_super [sedges, 2], new_e_attr_super [sedges, asw],
 * labels / has_obj [count], bbox_idx [nodes], bbox_targets [count, 4], stat [count, 13].                              */
int yolat_proposals_assemble(const yolat_proposals* p, const double* pos, const double* is_super, int64_t sw,
                             const int64_t* edge, const double* e_attr, int64_t aw, const int64_t* edge_super,
                             const double* e_attr_super, int64_t asw, const double* gt_bbox, const int64_t* gt_labels,
                             const int64_t* valid_ptr, const int64_t* valid_idx, int64_t n_classes, int normalize,
                             int stat_feats, double* new_pos, double* new_is_super, int64_t* new_edge, double* new_e_attr,
                             int64_t* new_edge_super, double* new_e_attr_super, int64_t* labels, int64_t* has_obj,
                             int64_t* bbox_idx, double* bbox_targets, double* stat);
void yolat_proposals_free(yolat_proposals* p);

#ifdef __cplusplus
}
#endif
#endif /* YOLAT_HIP_H */
