"""Tensor-level wrappers of the C ABI (include/yolat_hip.h).

Every function takes torch CUDA tensors, checks dtype / device / inner stride, and enqueues the HIP
kernel(s) on torch's current stream.  torch is used for device memory and streams only — the
arithmetic is in libyolat_hip.so.  There is no CPU path: a non-CUDA tensor raises.
"""
import ctypes
import os

import torch

from ._lib import lib, check

STATS_ROWS = 32
STATUS_EDGE_RANGE, STATUS_SEG_UNSORTED, STATUS_SEG_RANGE, STATUS_NOT_LOCAL = 1, 2, 4, 8

# The HIP kernels write parameters and BatchNorm buffers through raw pointers, which torch's per-tensor `_version`
# counters never see.  Every wrapper below that modifies model state (Adam step, running-statistic updates) bumps
# this epoch; every cache derived from model state (folded eval BatchNorm coefficients in engine._bn_eval, the
# eval plan's descriptor in plan.EvalPlan) carries it in its key.
_WEIGHT_EPOCH = [0]


def weight_epoch():
    return _WEIGHT_EPOCH[0]


def bump_weight_epoch():
    _WEIGHT_EPOCH[0] += 1


_RAW_STREAM = getattr(torch._C, "_cuda_getCurrentRawStream", None)
_GET_DEVICE = getattr(torch._C, "_cuda_getDevice", None)
_STREAM_OBJ = {}


def _stream():
    """Raw handle of the current HIP stream.  (torch.cuda.current_stream() resolves the device through several Python
    layers: ~3 us a call, three calls per forward; the two C entry points are what it ends in.)"""
    if _RAW_STREAM is not None and _GET_DEVICE is not None:
        return _RAW_STREAM(_GET_DEVICE())
    return torch.cuda.current_stream().cuda_stream


def current_stream_object():
    """torch.cuda.Stream object of the current stream, cached per (device, raw handle)."""
    if _RAW_STREAM is None or _GET_DEVICE is None:
        return torch.cuda.current_stream()
    dev = _GET_DEVICE()
    raw = _RAW_STREAM(dev)
    hit = _STREAM_OBJ.get(dev)
    if hit is None or hit[0] != raw:
        hit = _STREAM_OBJ[dev] = (raw, torch.cuda.current_stream())
    return hit[1]


def _f(t, name="tensor", allow_none=False):
    """data_ptr of an fp32 CUDA tensor whose last dim is dense."""
    if t is None:
        if allow_none:
            return None
        raise ValueError("%s is None" % name)
    if not t.is_cuda:
        raise RuntimeError("%s must be a CUDA (ROCm) tensor — the yolat HIP path has no CPU fallback" % name)
    if t.dtype != torch.float32:
        raise TypeError("%s must be float32, got %s" % (name, t.dtype))
    if t.dim() >= 1 and t.shape[-1] > 1 and t.stride(-1) != 1 and t.numel() > 0:      # (an empty tensor's strides mean nothing)
        raise ValueError("%s must be dense along its last dimension" % name)
    return t.data_ptr()


def _h(t, name="tensor"):
    """data_ptr of a bfloat16 CUDA tensor whose last dim is dense (bf16-storage training tensors)."""
    if not t.is_cuda or t.dtype != torch.bfloat16:
        raise TypeError("%s must be a bfloat16 CUDA tensor" % name)
    if t.dim() >= 1 and t.shape[-1] > 1 and t.stride(-1) != 1 and t.numel() > 0:      # (an empty tensor's strides mean nothing)
        raise ValueError("%s must be dense along its last dimension" % name)
    return t.data_ptr()


def _is_h(t):
    return t is not None and t.dtype == torch.bfloat16


def _i(t, dtype, name="index"):
    if t is None:
        return None
    if not t.is_cuda:
        raise RuntimeError("%s must be a CUDA (ROCm) tensor" % name)
    if t.dtype != dtype:
        raise TypeError("%s must be %s, got %s" % (name, dtype, t.dtype))
    if t.dim() == 1 and t.numel() > 1 and t.stride(0) != 1:
        raise ValueError("%s must be contiguous" % name)
    return t.data_ptr()


def _ld(t):
    return t.stride(0) if t.shape[0] > 1 else max(t.shape[-1], t.stride(0))


# ---------------------------------------------------------------------------------------------
# graph pre-processing
# ---------------------------------------------------------------------------------------------

_ZERO_STATUS = {}


class Graph(object):
    """Device-resident integer structure of one batch (CSR by destination, optional CSC by source,
    proposal segments).  Owned by the caller / cached on the batch object."""

    __slots__ = ("N", "E", "P", "row_ptr", "perm", "src", "dst", "attr", "col_ptr", "slots",
                 "seg_ptr", "node_seg", "status", "_work", "_inv_deg")

    @classmethod
    def from_arrays(cls, N, E, P, row_ptr, src, dst, attr, seg_ptr, node_seg, perm=None):
        """A Graph over device arrays prepared elsewhere (data.collate_to_device(csr=True): the batch's CSR merged on
        the host from the items' cached CSRs and shipped with the batch).  The ids were validated where the arrays
        were built, so the status word is zero."""
        g = cls()
        g._inv_deg = None
        g.N, g.E, g.P = int(N), int(E), int(P)
        g.row_ptr, g.src, g.dst, g.attr, g.seg_ptr, g.node_seg, g.perm = row_ptr, src, dst, attr, seg_ptr, node_seg, perm
        g.col_ptr = g.slots = None
        g._work = None
        st = _ZERO_STATUS.get(row_ptr.device)          # shared, never written for a prepared graph
        if st is None:
            st = _ZERO_STATUS[row_ptr.device] = torch.zeros(1, dtype=torch.int32, device=row_ptr.device)
        g.status = st
        return g

    def device_pointers(self):
        """(row_ptr, src, dst, attr, seg_ptr, node_seg) device addresses — what yolat_graph_csr carries."""
        return (self.row_ptr.data_ptr(), self.src.data_ptr(), self.dst.data_ptr(), self.attr.data_ptr(),
                self.seg_ptr.data_ptr(), self.node_seg.data_ptr())

    def inv_deg(self):
        """[N] fp32 1 / max(in-degree, 1) (the factor of the mean aggregation's backward), built once per graph."""
        if getattr(self, "_inv_deg", None) is None:
            inv = torch.empty(self.N, dtype=torch.float32, device=self.row_ptr.device)
            check(lib.yolat_inv_degree(self.row_ptr.data_ptr(), self.N, inv.data_ptr(), _stream()), "yolat_inv_degree")
            self._inv_deg = inv
        return self._inv_deg

    def ensure_csc(self):
        if self.col_ptr is None:
            dev = self.row_ptr.device
            self.col_ptr = torch.empty(self.N + 1, dtype=torch.int32, device=dev)
            self.slots = torch.empty(max(self.E, 1), dtype=torch.int32, device=dev)
            work = torch.empty(int(lib.yolat_csc_work_elems(self.N)), dtype=torch.int32, device=dev)
            check(lib.yolat_csc_by_source(self.src.data_ptr(), self.E, self.N, self.col_ptr.data_ptr(),
                                          self.slots.data_ptr(), work.data_ptr(), _stream()),
                  "yolat_csc_by_source")
            self._work = work  # keep alive until the stream has consumed it
        return self

    def check_status(self):
        """Synchronising validity check of the flags raised by the pre-processing kernels."""
        s = int(self.status.item())
        if s & STATUS_EDGE_RANGE:
            raise IndexError("edge_index contains a node id outside [0, N)")
        if s & STATUS_SEG_UNSORTED:
            raise ValueError("bbox_idx is not non-decreasing")
        if s & STATUS_SEG_RANGE:
            raise IndexError("bbox_idx contains a proposal id outside [0, P)")
        if s & STATUS_NOT_LOCAL:
            raise ValueError("a batch handed over as proposal-local (yolat_locality) is not: an edge leaves its proposal, "
                             "the edge list is not grouped by proposal, or a proposal does not fit a tile")
        return True


class PackedGraph(Graph):
    """A prepared graph whose six arrays live at known offsets of ONE device buffer (data.collate_to_device(csr=True)):
    the eval forward only needs their addresses (device_pointers), so the tensor views are made on first use — the
    training path and the tests read them — instead of twelve tensor operations per batch."""

    __slots__ = ("_buf", "_offs", "_views")
    _FIELDS = {"row_ptr": 0, "src": 1, "dst": 2, "attr": 3, "seg_ptr": 4, "node_seg": 5}

    @classmethod
    def from_buffer(cls, buf, offs, N, E, P):
        g = cls()
        g._inv_deg = None
        g.N, g.E, g.P = int(N), int(E), int(P)
        g._buf, g._offs, g._views = buf, tuple(int(o) for o in offs), {}
        g.perm = None
        g.col_ptr = g.slots = None
        g._work = None
        st = _ZERO_STATUS.get(buf.device)
        if st is None:
            st = _ZERO_STATUS[buf.device] = torch.zeros(1, dtype=torch.int32, device=buf.device)
        g.status = st
        return g

    def device_pointers(self):
        base = self._buf.data_ptr()
        return tuple(base + o for o in self._offs)

    def _view(self, name):
        v = self._views.get(name)
        if v is None:
            i = self._FIELDS[name]
            Ee = max(self.E, 1)
            shape = ((self.N + 1,), (Ee,), (Ee,), (Ee, 4), (self.P + 1,), (self.N,))[i]
            dtype = torch.float32 if name == "attr" else torch.int32
            n = 1
            for d in shape:
                n *= d
            o = self._offs[i] // 4
            v = self._views[name] = self._buf.view(dtype)[o:o + n].view(shape)
        return v

    row_ptr = property(lambda self: self._view("row_ptr"))
    src = property(lambda self: self._view("src"))
    dst = property(lambda self: self._view("dst"))
    attr = property(lambda self: self._view("attr"))
    seg_ptr = property(lambda self: self._view("seg_ptr"))
    node_seg = property(lambda self: self._view("node_seg"))


def build_graph(edge, e_attr, bbox_idx, num_nodes, num_proposals):
    """edge: int64 CUDA tensor, either [E,2] (data.edge) or its [2,E] transposed view;
    e_attr: fp32 [E,4]; bbox_idx: int64 [N] or None."""
    if edge.dim() != 2:
        raise ValueError("edge must be 2-D")
    if edge.shape[0] == 2 and edge.shape[1] != 2:
        E, se, sc = edge.shape[1], edge.stride(1), edge.stride(0)
    elif edge.shape[1] == 2:
        if edge.shape[0] == 2 and edge.stride(0) == 1:      # a [2,2] transposed view
            E, se, sc = 2, edge.stride(1), edge.stride(0)
        else:
            E, se, sc = edge.shape[0], edge.stride(0), edge.stride(1)
    else:
        raise ValueError("edge must be [E,2] or [2,E]")
    dev = edge.device
    N, P = int(num_nodes), int(num_proposals)
    g = Graph()
    g._inv_deg = None
    g.N, g.E, g.P = N, E, P
    g.col_ptr = g.slots = None
    g.row_ptr = torch.empty(N + 1, dtype=torch.int32, device=dev)
    g.perm = torch.empty(max(E, 1), dtype=torch.int32, device=dev)
    g.src = torch.empty(max(E, 1), dtype=torch.int32, device=dev)
    g.dst = torch.empty(max(E, 1), dtype=torch.int32, device=dev)
    g.attr = torch.empty(max(E, 1), 4, dtype=torch.float32, device=dev)
    g.seg_ptr = g.node_seg = None
    if bbox_idx is not None:
        # zero-initialised: with a malformed (unsorted) bbox_idx the kernels flag STATUS_SEG_UNSORTED and the
        # pointers stay inside [0, N] whatever the write order, so every downstream kernel is memory-safe.
        # (status word, segment pointers and node segments as slices of ONE zeroed buffer: one fill, not three)
        n_seg = (P + 1 + 3) // 4 * 4
        z = torch.zeros(4 + n_seg + max(N, 1), dtype=torch.int32, device=dev)
        g.status = z[0:1]
        g.seg_ptr = z[4:4 + P + 1]
        g.node_seg = z[4 + n_seg:4 + n_seg + max(N, 1)]
    else:
        g.status = torch.zeros(1, dtype=torch.int32, device=dev)
    if E > 0:
        if e_attr.shape[0] != E or e_attr.shape[1] != 4:
            raise ValueError("e_attr must be [E,4]")
        if not e_attr.is_contiguous():
            e_attr = e_attr.contiguous()
    work = torch.empty(int(lib.yolat_graph_work_elems(N, E)), dtype=torch.int32, device=dev)
    check(lib.yolat_graph_prepare(_i(edge, torch.int64, "edge"), se, sc, _f(e_attr, "e_attr") if E > 0 else None,
                                  _i(bbox_idx, torch.int64, "bbox_idx"), E, N, P, g.row_ptr.data_ptr(),
                                  g.perm.data_ptr(), g.src.data_ptr(), g.dst.data_ptr(), g.attr.data_ptr(),
                                  g.seg_ptr.data_ptr() if g.seg_ptr is not None else None,
                                  g.node_seg.data_ptr() if g.node_seg is not None else None,
                                  work.data_ptr(), g.status.data_ptr(), _stream()), "yolat_graph_prepare")
    g._work = work
    return g


# ---------------------------------------------------------------------------------------------
# dense
# ---------------------------------------------------------------------------------------------

def stats_buffer(M, C, device):
    return torch.empty(int(lib.yolat_bn_stats_elems(M, C)), dtype=torch.float32, device=device)


# YOLAT_STRICT_FP32=1: no bf16x6-emulated GEMM anywhere (csrc/x6.hpp; the C side reads the same variable)
STRICT_FP32 = os.environ.get("YOLAT_STRICT_FP32", "0") == "1"
X6_TRAIN_GEMM = not STRICT_FP32          # module flag (tests/test_gpu_ops.py flips it)
# the many-row training Linear on the bf16x6 rows kernel: measured EQUAL to the fp32-MFMA tiles (201 vs 199 us for
# [1.2 M, 64] -> [1.2 M, 64]: one 8-wave workgroup per CU, its load -> split -> MFMA -> store chain is not overlapped
# with a neighbour's), so it stays opt-in
X6_TRAIN_ROWS = False                    # module flag (tests/test_gpu_ops.py flips it)


def linear_fwd(A, W, bias, Y, a_pro=None, a_relu=False, o_pro=None, o_relu=False,
               accumulate=False, stats=None):
    """Y = epi(pro(A) @ W.T + bias).  a_pro / o_pro: (scale, shift) tensors or None."""
    M, K = A.shape
    Nout = W.shape[0]
    asc, ash = (a_pro if a_pro is not None else (None, None))
    osc, osh = (o_pro if o_pro is not None else (None, None))
    if _is_h(A) or _is_h(Y):
        if not (_is_h(A) and _is_h(Y)) or o_pro is not None or o_relu or accumulate:
            raise ValueError("bf16-storage linear_fwd: A and Y bfloat16, no output epilogue / accumulation")
        wwork = torch.empty(Nout * K, dtype=torch.bfloat16, device=A.device)
        check(lib.yolat_linear_fwd_h(_h(A, "A"), _ld(A), M, K, _f(asc, "a_scale", True), _f(ash, "a_shift", True),
                                     int(a_relu), _f(W, "W"), _ld(W), _f(bias, "bias", True), Nout, _h(Y, "Y"), _ld(Y),
                                     _f(stats, "stats", True), wwork.data_ptr(), _stream()), "yolat_linear_fwd_h")
        return Y
    if (X6_TRAIN_GEMM and stats is not None and a_pro is None and o_pro is None and not o_relu and not accumulate
            and bias is not None and M >= 1024 and K >= 256 and K % 16 == 0 and Nout >= 128 and _ld(A) % 4 == 0
            and A.data_ptr() % 16 == 0 and int(lib.yolat_gemm_x6_work_elems(M, Nout, K)) == 0):
        # many rows x long K in front of a training BatchNorm (the classifier's first layer at P = 8000): bf16x6-emulated
        # LDS-tiled GEMM with the statistics epilogue; the weight is packed per call
        packed = torch.empty(int(lib.yolat_gemm_x6_packed_elems(Nout, K)), dtype=torch.bfloat16, device=A.device)
        check(lib.yolat_gemm_x6_pack(_f(W, "W"), _ld(W), Nout, K, None, packed.data_ptr(), _stream()), "yolat_gemm_x6_pack")
        check(lib.yolat_gemm_x6_stats(_f(A, "A"), _ld(A), M, K, packed.data_ptr(), _f(bias, "bias"), Nout, _f(Y, "Y"), _ld(Y),
                                      _f(stats, "stats"), _stream()), "yolat_gemm_x6_stats")
        return Y
    if (X6_TRAIN_ROWS and o_pro is None and not o_relu and not accumulate and bias is not None and M >= 65536
            and K in (64, 128) and Nout % 64 == 0 and _ld(A) % 4 == 0 and A.data_ptr() % 16 == 0):
        # many rows x short K with a pre-activation output (the second edge Linear of a training conv layer, [E,64] ->
        # [E,64] + BatchNorm statistics): the bf16x6 rows kernel, A read once per 256 rows, weight split per call
        wsplit = torch.empty(3 * Nout * K, dtype=torch.bfloat16, device=A.device)
        check(lib.yolat_linear_fwd_rows_x6(_f(A, "A"), _ld(A), M, K, _f(asc, "a_scale", True), _f(ash, "a_shift", True),
                                           int(a_relu), _f(W, "W"), _ld(W), _f(bias, "bias"), Nout, _f(Y, "Y"), _ld(Y),
                                           _f(stats, "stats", True), wsplit.data_ptr(), _stream()),
              "yolat_linear_fwd_rows_x6")
        return Y
    check(lib.yolat_linear_fwd(_f(A, "A"), _ld(A), M, K, _f(asc, "a_scale", True), _f(ash, "a_shift", True),
                               int(a_relu), _f(W, "W"), _ld(W), _f(bias, "bias", True), Nout,
                               _f(osc, "o_scale", True), _f(osh, "o_shift", True), int(o_relu),
                               _f(Y, "Y"), _ld(Y), int(accumulate), _f(stats, "stats", True),
                               _stream()), "yolat_linear_fwd")
    return Y


def linear_fwd_wt(A, Wt, Y, accumulate=False):
    """Y = A @ Wt   (Wt: [K, Nout] row-major, e.g. dX = dY @ W)."""
    M, K = A.shape
    Nout = Wt.shape[1]
    if _is_h(A) or _is_h(Y):
        if not (_is_h(A) and _is_h(Y)) or accumulate:
            raise ValueError("bf16-storage linear_fwd_wt: A and Y bfloat16, no accumulation")
        wwork = torch.empty(Nout * K, dtype=torch.bfloat16, device=A.device)
        check(lib.yolat_linear_fwd_wt_h(_h(A, "A"), _ld(A), M, K, _f(Wt, "Wt"), _ld(Wt), Nout, _h(Y, "Y"), _ld(Y),
                                        wwork.data_ptr(), _stream()), "yolat_linear_fwd_wt_h")
        return Y
    if (X6_TRAIN_GEMM and not accumulate and M >= 1024 and K >= 256 and K % 16 == 0 and Nout >= 512
            and _ld(A) % 4 == 0 and A.data_ptr() % 16 == 0):
        # long-K, many-row backward GEMM (the classifier's first layer: dZ [P, 2304] = dC1 [P, 512] . Wc1): bf16x6-
        # emulated on the LDS-tiled kernel (gemm_x6.hip); the weight changes every step, so it is packed here (a few us)
        packed = torch.empty(int(lib.yolat_gemm_x6_packed_elems(Nout, K)), dtype=torch.bfloat16, device=A.device)
        check(lib.yolat_gemm_x6_pack_t(_f(Wt, "Wt"), _ld(Wt), Nout, K, packed.data_ptr(), _stream()),
              "yolat_gemm_x6_pack_t")
        work = torch.empty(max(1, int(lib.yolat_gemm_x6_work_elems(M, Nout, K))), dtype=torch.float32, device=A.device)
        check(lib.yolat_gemm_x6(_f(A, "A"), _ld(A), M, K, packed.data_ptr(), None, 0, Nout, _f(Y, "Y"), _ld(Y),
                                work.data_ptr(), _stream()), "yolat_gemm_x6")
        return Y
    check(lib.yolat_linear_fwd_wt(_f(A, "A"), _ld(A), M, K, _f(Wt, "Wt"), _ld(Wt), Nout, _f(Y, "Y"),
                                  _ld(Y), int(accumulate), _stream()), "yolat_linear_fwd_wt")
    return Y


def linear_bwd_w(dY, A, dW, db=None, a_pro=None, a_relu=False, accumulate=False):
    M, Nout = dY.shape
    K = A.shape[1]
    asc, ash = (a_pro if a_pro is not None else (None, None))
    work = torch.empty(int(lib.yolat_linear_bwd_w_work_elems(M, Nout, K)), dtype=torch.float32,
                       device=dY.device)
    if _is_h(dY):
        check(lib.yolat_linear_bwd_w_h(_h(dY, "dY"), _ld(dY), M, Nout, _h(A, "A") if _is_h(A) else _f(A, "A"),
                                       int(_is_h(A)), _ld(A), K, _f(asc, "a_scale", True), _f(ash, "a_shift", True),
                                       int(a_relu), _f(dW, "dW"), _ld(dW), _f(db, "db", True), int(accumulate),
                                       work.data_ptr(), _stream()), "yolat_linear_bwd_w_h")
        return dW
    check(lib.yolat_linear_bwd_w(_f(dY, "dY"), _ld(dY), M, Nout, _f(A, "A"), _ld(A), K,
                                 _f(asc, "a_scale", True), _f(ash, "a_shift", True), int(a_relu),
                                 _f(dW, "dW"), _ld(dW), _f(db, "db", True), int(accumulate),
                                 work.data_ptr(), _stream()), "yolat_linear_bwd_w")
    return dW


class BnCsrGrad(object):
    """The gradient w.r.t. the input Y [E,C] of BatchNorm(train)+ReLU followed by the CSR mean aggregation, never
    materialised (bn_csr.hip): built from d_out [N,C] (gradient w.r.t. the aggregated output), the graph and the saved
    BatchNorm coefficients; `stats()` reduces dgamma / dbeta and the two coefficients the consumers need, then
    `bwd_w()` (dW = dY^T.pro(A), db) and `fwd_wt()` (dA = dY.W) form dY rows in their GEMM loaders."""

    def __init__(self, d_out, g, Y, save_mean, save_invstd, scale, shift, relu=True):
        from ._lib import BnCsrGrad as _S
        self.E, self.C, self.dev = Y.shape[0], Y.shape[1], Y.device
        self.coef = torch.empty(2 * self.C, dtype=torch.float32, device=self.dev)
        self._keep = (d_out, g, Y, save_mean, save_invstd, scale, shift, g.inv_deg())
        d = _S()
        d.d_out, d.ld_out = _f(d_out, "d_out"), _ld(d_out)
        d.dst, d.inv_deg = g.dst.data_ptr(), g.inv_deg().data_ptr()
        d.half = int(_is_h(Y))
        d.Y, d.ldy = (_h(Y, "Y") if d.half else _f(Y, "Y")), _ld(Y)
        d.mean, d.invstd, d.scale, d.shift = _f(save_mean), _f(save_invstd), _f(scale), _f(shift)
        d.coef, d.relu = self.coef.data_ptr(), int(relu)
        self._d = d

    def stats(self, dgamma, dbeta, accumulate=False):
        work = torch.empty(int(lib.yolat_bn_csr_work_elems(self.E, self.C)), dtype=torch.float32, device=self.dev)
        check(lib.yolat_bn_csr_bwd_stats(ctypes.byref(self._d), self.E, self.C, _f(dgamma), _f(dbeta), int(accumulate),
                                         self.coef.data_ptr(), work.data_ptr(), _stream()), "yolat_bn_csr_bwd_stats")
        return self

    def bwd_w(self, A, dW, db=None, a_pro=None, a_relu=False, accumulate=False):
        K = A.shape[1]
        asc, ash = (a_pro if a_pro is not None else (None, None))
        work = torch.empty(int(lib.yolat_linear_bwd_w_work_elems(self.E, self.C, K)), dtype=torch.float32, device=self.dev)
        check(lib.yolat_linear_bwd_w_csr(ctypes.byref(self._d), self.E, self.C, _f(A, "A"), _ld(A), K,
                                         _f(asc, "a_scale", True), _f(ash, "a_shift", True), int(a_relu), _f(dW, "dW"),
                                         _ld(dW), _f(db, "db", True), int(accumulate), work.data_ptr(), _stream()),
              "yolat_linear_bwd_w_csr")
        return dW

    def bwd_w_and_x(self, A, W, dW, db, dA, a_pro=None, a_relu=False, accumulate=False, next_bn=None):
        """dW (+)= dY^T . pro(A), db (+)= column sums, dA = dY . W in one kernel (C = K = 64).  next_bn = (save_mean,
        save_invstd, dgamma, dbeta) of the BatchNorm behind a_pro: its backward statistics on dA come out of the same
        kernel; returns their coefficient vector [2C] for bn_relu_bwd_apply (else None)."""
        asc, ash = (a_pro if a_pro is not None else (None, None))
        work = torch.empty(int(lib.yolat_bn_csr_l2_bwd_work_elems()), dtype=torch.float32, device=self.dev)
        hp = bool(self._d.half)
        if _is_h(A) != hp or _is_h(dA) != hp:
            raise ValueError("BnCsrGrad.bwd_w_and_x: A and dA must use the storage type of Y")
        coef1 = None
        nm = ni = ng = nb = nc = None
        if next_bn is not None:
            coef1 = torch.empty(2 * self.C, dtype=torch.float32, device=self.dev)
            nm, ni, ng, nb, nc = _f(next_bn[0]), _f(next_bn[1]), _f(next_bn[2]), _f(next_bn[3]), coef1.data_ptr()
        check(lib.yolat_bn_csr_l2_bwd(ctypes.byref(self._d), self.E, _h(A, "A") if hp else _f(A, "A"), _ld(A),
                                      _f(asc, "a_scale", True), _f(ash, "a_shift", True), int(a_relu), _f(W, "W"), _ld(W),
                                      _f(dW, "dW"), _ld(dW), _f(db, "db", True), int(accumulate),
                                      _h(dA, "dA") if hp else _f(dA, "dA"), _ld(dA), work.data_ptr(), nm, ni, ng, nb, nc,
                                      _stream()), "yolat_bn_csr_l2_bwd")
        return coef1

    def fwd_wt(self, W, dA):
        check(lib.yolat_linear_fwd_wt_csr(ctypes.byref(self._d), self.E, self.C, _f(W, "W"), _ld(W), W.shape[1],
                                          _f(dA, "dA"), _ld(dA), _stream()), "yolat_linear_fwd_wt_csr")
        return dA


def bn_relu_bwd_apply(dZ, Y, save_mean, save_invstd, scale, shift, relu, coef, dY):
    """The apply pass of bn_relu_bwd alone, with the coefficient vector [2C] = (c1 | c2) given."""
    M, C = Y.shape
    hp = _is_h(Y)
    check(lib.yolat_bn_relu_bwd_apply(_h(dZ, "dZ") if hp else _f(dZ, "dZ"), _ld(dZ), _h(Y, "Y") if hp else _f(Y, "Y"), _ld(Y),
                                      M, C, _f(save_mean), _f(save_invstd), _f(scale), _f(shift), int(relu), _f(coef),
                                      _h(dY, "dY") if hp else _f(dY, "dY"), _ld(dY), int(hp), _stream()),
          "yolat_bn_relu_bwd_apply")
    return dY


def bn_finalize(stats, M, bn, scale, shift, save_mean, save_invstd, update_running=True):
    C = bn.num_features
    rm = bn.running_mean if (update_running and bn.track_running_stats) else None
    rv = bn.running_var if (update_running and bn.track_running_stats) else None
    mom = 0.1 if bn.momentum is None else float(bn.momentum)
    if rm is not None:
        bump_weight_epoch()
    check(lib.yolat_bn_finalize(_f(stats), M, C, _f(bn.weight), _f(bn.bias), _f(rm, "rm", True),
                                _f(rv, "rv", True), mom, float(bn.eps), _f(save_mean),
                                _f(save_invstd), _f(scale), _f(shift), _stream()), "yolat_bn_finalize")


def bn_eval_coeffs(bn, scale, shift):
    check(lib.yolat_bn_eval_coeffs(_f(bn.weight), _f(bn.bias), _f(bn.running_mean), _f(bn.running_var),
                                   float(bn.eps), bn.num_features, _f(scale), _f(shift), _stream()),
          "yolat_bn_eval_coeffs")


def scale_shift_relu(Y, scale, shift, relu, Z):
    M, C = Y.shape
    check(lib.yolat_scale_shift_relu(_f(Y), _ld(Y), M, C, _f(scale, "scale", True), _f(shift, "shift", True),
                                     int(relu), _f(Z), _ld(Z), _stream()), "yolat_scale_shift_relu")
    return Z


def bn_relu_bwd(dZ, Y, gamma, save_mean, save_invstd, scale, shift, relu, dgamma, dbeta, dY,
                accumulate=False):
    M, C = Y.shape
    work = torch.empty(int(lib.yolat_bn_bwd_work_elems(M, C)), dtype=torch.float32, device=Y.device)
    if _is_h(Y):
        check(lib.yolat_bn_relu_bwd_h(_h(dZ, "dZ"), _ld(dZ), _h(Y, "Y"), _ld(Y), M, C, _f(save_mean), _f(save_invstd),
                                      _f(scale), _f(shift), int(relu), _f(dgamma), _f(dbeta), int(accumulate),
                                      _h(dY, "dY"), _ld(dY), work.data_ptr(), _stream()), "yolat_bn_relu_bwd_h")
        return dY
    check(lib.yolat_bn_relu_bwd(_f(dZ), _ld(dZ), _f(Y), _ld(Y), M, C, _f(gamma), _f(save_mean),
                                _f(save_invstd), _f(scale), _f(shift), int(relu), _f(dgamma),
                                _f(dbeta), int(accumulate), _f(dY), _ld(dY), work.data_ptr(),
                                _stream()), "yolat_bn_relu_bwd")
    return dY


# ---------------------------------------------------------------------------------------------
# edge convolution
# ---------------------------------------------------------------------------------------------

def edge_lin1_fwd(x, g, W1, b1, H1, o_pro=None, o_relu=False, stats=None):
    N, Cin = x.shape
    C = W1.shape[0]
    osc, osh = (o_pro if o_pro is not None else (None, None))
    check(lib.yolat_edge_lin1_fwd(_f(x, "x"), _ld(x), N, Cin, g.src.data_ptr(), g.dst.data_ptr(),
                                  g.attr.data_ptr(), g.E, _f(W1), _ld(W1), _f(b1, "b1", True), C,
                                  _f(osc, "o_scale", True), _f(osh, "o_shift", True), int(o_relu),
                                  _f(H1), _ld(H1), _f(stats, "stats", True), _stream()),
          "yolat_edge_lin1_fwd")
    return H1


def split_w1(W1, Cin):
    """yolat_conv_split_w1: Wuv [2C,Cin] = [W1a - W1b | W1b] (rows), Wc4 [C,4] = the attr columns."""
    C = W1.shape[0]
    if not W1.is_contiguous():
        raise ValueError("W1 must be contiguous")
    wuv = torch.empty(2 * C, Cin, dtype=torch.float32, device=W1.device)
    wc4 = torch.empty(C, 4, dtype=torch.float32, device=W1.device)
    check(lib.yolat_conv_split_w1(_f(W1), Cin, C, wuv.data_ptr(), wc4.data_ptr(), _stream()), "yolat_conv_split_w1")
    return wuv, wc4


def attr_dw(dH1, g, dwc4, db1=None):
    """dWc4 [C, 4] = dH1^T . attr, db1 = column sums of dH1: the streaming reduction yolat_edge_attr_dw (C = 64, rows
    16-byte aligned); other shapes fall back to the general weight-gradient GEMM."""
    E, C = dH1.shape
    half = _is_h(dH1)
    if C == 64 and dH1.stride(1) == 1 and dH1.stride(0) % 4 == 0 and dH1.data_ptr() % 16 == 0 and E > 0:
        work = torch.empty(int(lib.yolat_edge_attr_dw_work_elems(E)), dtype=torch.float32, device=dH1.device)
        check(lib.yolat_edge_attr_dw(dH1.data_ptr(), dH1.stride(0), 1 if half else 0, g.attr.data_ptr(), E, C, _f(dwc4),
                                     _f(db1, "db1", True), work.data_ptr(), _stream()), "yolat_edge_attr_dw")
        return dwc4
    return linear_bwd_w(dH1, g.attr, dwc4, db1)


def bn_apply_edge_sums(dA1, H1, save_mean, save_invstd, scale, shift, relu, coef, g, db1):
    """The BatchNorm + ReLU backward apply pass in front of the factorised first edge Linear, fused with the consumers
    that read its result in CSR order (yolat_bn_apply_edge_sums): dA1 -> dH1 in place, returns (dUV [N, 128] with the dU
    half written, dWc4 [64, 4]); db1 is written.  Hand both to edge_lin1_bwd_factorised(..., partial=(dUV, dWc4))."""
    E, C = H1.shape
    N = g.N
    hp = _is_h(H1)
    dUV = torch.empty(N, 2 * C, dtype=torch.float32, device=H1.device)
    dwc4 = torch.empty(C, 4, dtype=torch.float32, device=H1.device)
    work = torch.empty(int(lib.yolat_bn_apply_edge_sums_work_elems(N)), dtype=torch.float32, device=H1.device)
    check(lib.yolat_bn_apply_edge_sums(_h(dA1, "dA1") if hp else _f(dA1, "dA1"), _ld(dA1), _h(H1, "H1") if hp else _f(H1, "H1"),
                                       _ld(H1), dA1.data_ptr(), _ld(dA1), int(hp), E, _f(save_mean), _f(save_invstd),
                                       _f(scale), _f(shift), int(relu), _f(coef), g.row_ptr.data_ptr(), g.attr.data_ptr(), N,
                                       dUV.data_ptr(), 2 * C, dwc4.data_ptr(), _f(db1, "db1", True), work.data_ptr(),
                                       _stream()), "yolat_bn_apply_edge_sums")
    return dUV, dwc4


def edge_lin1_bwd_factorised(dH1, x, g, W1, dW1, db1, dx=None, dx_accumulate=False, wuv=None, side=None, partial=None):
    """Backward of the first edge Linear through the per-node products (see yolat_edge_uv_sums): writes dW1, db1 and,
    when `dx` is given, (accumulates) the gradient w.r.t. the node features.  C = 64; pays when E >> N.
    partial = (dUV, dWc4) from bn_apply_edge_sums: the dU half, dWc4 and db1 exist already — only the gathered dV half
    is left to sum."""
    N, Cin = x.shape
    C = W1.shape[0]
    g.ensure_csc()
    if wuv is None:
        wuv, _ = split_w1(W1, Cin)
    dwc4_done = None
    if partial is not None:
        dUV, dwc4_done = partial
        check(lib.yolat_edge_uv_sums_v(dH1.data_ptr(), _ld(dH1), int(_is_h(dH1)), g.col_ptr.data_ptr(), g.slots.data_ptr(),
                                       N, C, dUV.data_ptr(), 2 * C, _stream()), "yolat_edge_uv_sums_v")
    else:
        dUV = torch.empty(N, 2 * C, dtype=torch.float32, device=x.device)
        if _is_h(dH1):
            check(lib.yolat_edge_uv_sums_h(_h(dH1), _ld(dH1), g.row_ptr.data_ptr(), g.col_ptr.data_ptr(),
                                           g.slots.data_ptr(), N, C, dUV.data_ptr(), 2 * C, _stream()),
                  "yolat_edge_uv_sums_h")
        else:
            check(lib.yolat_edge_uv_sums(_f(dH1), _ld(dH1), g.row_ptr.data_ptr(), g.col_ptr.data_ptr(),
                                         g.slots.data_ptr(), N, C, dUV.data_ptr(), 2 * C, _stream()), "yolat_edge_uv_sums")
    def weight_grads():
        dwuv = torch.empty(2 * C, Cin, dtype=torch.float32, device=x.device)
        linear_bwd_w(dUV, x, dwuv)
        if dwc4_done is not None:
            dwc4 = dwc4_done
        else:
            dwc4 = torch.empty(C, 4, dtype=torch.float32, device=x.device)
            attr_dw(dH1, g, dwc4, db1)
        check(lib.yolat_conv_merge_dw1(dwuv.data_ptr(), dwc4.data_ptr(), Cin, C, _f(dW1), _ld(dW1), 0, _stream()),
              "yolat_conv_merge_dw1")
    # `side` (engine._on_side): the three weight-gradient launches on a second stream, beside the dx GEMM below and
    # whatever the caller issues next — nothing in the backward reads dW1 / db1
    if side is not None:
        side(weight_grads, (dUV, x, dH1, g.attr, dwc4_done))
    else:
        weight_grads()
    if dx is not None:
        linear_fwd_wt(dUV, wuv, dx, accumulate=dx_accumulate)
    return dW1


def edge_lin1_fwd_factorised(x, g, W1, b1, H1, stats=None, keep=None):
    """Same result as edge_lin1_fwd (to fp32 rounding) through the per-node products: UV = x.[W1a-W1b | W1b]^T by a
    dense GEMM over the N nodes, then a gather-add over the E edges (csrc/edge.hip, yolat_edge_uv_lin1_fwd).  Pays
    when E >> N; Cin == C == 64 only."""
    N, Cin = x.shape
    C = W1.shape[0]
    wuv, wc4 = split_w1(W1, Cin)
    if keep is not None:
        keep["wuv"] = wuv               # the backward of the same step needs the same split (edge_lin1_bwd_factorised)
    uv = torch.empty(N, 2 * C, dtype=torch.float32, device=x.device)
    linear_fwd(x, wuv, None, uv)
    if _is_h(H1):
        check(lib.yolat_edge_uv_lin1_fwd_h(uv.data_ptr(), 2 * C, g.src.data_ptr(), g.dst.data_ptr(), g.attr.data_ptr(),
                                           g.E, wc4.data_ptr(), _f(b1, "b1", True), C, _h(H1, "H1"), _ld(H1),
                                           _f(stats, "stats", True), _stream()), "yolat_edge_uv_lin1_fwd_h")
        return H1
    check(lib.yolat_edge_uv_lin1_fwd(uv.data_ptr(), 2 * C, g.src.data_ptr(), g.dst.data_ptr(), g.attr.data_ptr(), g.E,
                                     wc4.data_ptr(), _f(b1, "b1", True), C, _f(H1), _ld(H1),
                                     _f(stats, "stats", True), _stream()), "yolat_edge_uv_lin1_fwd")
    return H1


def fold_factorised_layer(wuv, wc4, b1, s1, t1, b2, s2, t2):
    """Folded form of a factorised eval conv layer (include/yolat_hip.h, yolat_conv_eval.{Wuvf, uvb, Wc4f, t2f}):
    nn.1's folded BatchNorm (s1, t1) and the bias b1 move into the per-node products and the attr weights, b2 into the
    shift of nn.4, so the per-edge arithmetic is  h1 = relu(U'[dst] + V'[src] + Wc4f.attr),
    message = relu(s2 * (W2.h1) + t2f).  Elementwise work on [C]- / weight-sized tensors, once per weight version."""
    zeros = torch.zeros_like(s1)
    b1 = zeros if b1 is None else b1.detach()
    b2 = zeros if b2 is None else b2.detach()
    wuvf = (wuv * torch.cat([s1, s1]).unsqueeze(1)).contiguous()
    uvb = torch.cat([s1 * b1 + t1, zeros]).contiguous()
    wc4f = (wc4 * s1.unsqueeze(1)).contiguous()
    t2f = (s2 * b2 + t2).contiguous()
    return wuvf, uvb, wc4f, t2f


EDGE_AUTO, EDGE_TILES, EDGE_WS_F32, EDGE_WS_X6 = 0, 1, 2, 3


def edge_uv_mlp2_mean_eval(UV, g, wc4, b1, pro1, W2, b2, pro2, f_out, variant=EDGE_AUTO):
    """f_out[n] += mean over the CSR row of n of relu(s2*(W2.relu(s1*(U[dst]+V[src]+wc4.attr+b1)+t1)+b2)+t2)
    (yolat_edge_uv_mlp2_mean_eval_variant; b1 / pro1 / b2 may be None)."""
    s1, t1 = pro1 if pro1 is not None else (None, None)
    s2, t2 = pro2 if pro2 is not None else (None, None)
    check(lib.yolat_edge_uv_mlp2_mean_eval_variant(_f(UV, "UV"), _ld(UV), g.src.data_ptr(), g.dst.data_ptr(),
                                                   g.attr.data_ptr(), g.row_ptr.data_ptr(), g.N, g.E, _f(wc4),
                                                   _f(b1, "b1", True), _f(s1, "s1", True), _f(t1, "t1", True), _f(W2),
                                                   _f(b2, "b2", True), _f(s2, "s2", True), _f(t2, "t2", True),
                                                   W2.shape[0], _f(f_out), _ld(f_out), int(variant), _stream()),
          "yolat_edge_uv_mlp2_mean_eval_variant")
    return f_out


def edge_mlp2_eval(x, g, W1, b1, pro1, W2, b2, pro2, H2):
    """Eval-mode two-layer edge MLP in one kernel (BN folded into pro1/pro2 = (scale, shift))."""
    N, Cin = x.shape
    C = W1.shape[0]
    if not (W1.is_contiguous() and W2.is_contiguous()):
        raise ValueError("W1 / W2 must be contiguous")
    check(lib.yolat_edge_mlp2_eval(_f(x, "x"), _ld(x), N, Cin, g.src.data_ptr(), g.dst.data_ptr(), g.attr.data_ptr(),
                                   g.E, _f(W1), _f(b1), _f(pro1[0]), _f(pro1[1]), _f(W2), _f(b2), _f(pro2[0]),
                                   _f(pro2[1]), C, _f(H2), _ld(H2), _stream()), "yolat_edge_mlp2_eval")
    return H2


def edge_lin1_bwd_w(dH1, x, g, dW1, db1=None, accumulate=False):
    E, C = dH1.shape[0], dW1.shape[0]
    N, Cin = x.shape
    work = torch.empty(int(lib.yolat_linear_bwd_w_work_elems(g.E, C, 2 * Cin + 4)), dtype=torch.float32,
                       device=x.device)
    check(lib.yolat_edge_lin1_bwd_w(_f(dH1), _ld(dH1), g.E, C, _f(x), _ld(x), N, Cin, g.src.data_ptr(),
                                    g.dst.data_ptr(), g.attr.data_ptr(), _f(dW1), _ld(dW1),
                                    _f(db1, "db1", True), int(accumulate), work.data_ptr(), _stream()),
          "yolat_edge_lin1_bwd_w")
    return dW1


def edge_lin1_bwd_x(dH1, W1, Cin, dG):
    C = W1.shape[0]
    check(lib.yolat_edge_lin1_bwd_x(_f(dH1), _ld(dH1), dH1.shape[0], C, _f(W1), _ld(W1), Cin, _f(dG),
                                    _ld(dG), _stream()), "yolat_edge_lin1_bwd_x")
    return dG


def edge_scatter_bwd(dG, Cin, g, dX, accumulate=False):
    g.ensure_csc()
    check(lib.yolat_edge_scatter_bwd(_f(dG), _ld(dG), Cin, g.row_ptr.data_ptr(), g.col_ptr.data_ptr(),
                                     g.slots.data_ptr(), g.N, _f(dX), _ld(dX), int(accumulate),
                                     _stream()), "yolat_edge_scatter_bwd")
    return dX


def csr_mean_fwd(H, g, out, h_pro=None, h_relu=False, accumulate=False):
    C = out.shape[1]
    hs, hb = (h_pro if h_pro is not None else (None, None))
    if _is_h(H):
        check(lib.yolat_csr_mean_fwd_h(_h(H, "H"), _ld(H), C, _f(hs, "h_scale", True), _f(hb, "h_shift", True),
                                       int(h_relu), g.row_ptr.data_ptr(), g.N, _f(out), _ld(out), int(accumulate),
                                       _stream()), "yolat_csr_mean_fwd_h")
        return out
    check(lib.yolat_csr_mean_fwd(_f(H, "H", g.E == 0), _ld(H) if H is not None else C, C,
                                 _f(hs, "h_scale", True), _f(hb, "h_shift", True), int(h_relu),
                                 g.row_ptr.data_ptr(), g.N, _f(out), _ld(out), int(accumulate),
                                 _stream()), "yolat_csr_mean_fwd")
    return out


def csr_mean_bwd(dOut, g, dM):
    C = dOut.shape[1]
    if _is_h(dM):
        check(lib.yolat_csr_mean_bwd_h(_f(dOut), _ld(dOut), C, g.row_ptr.data_ptr(), g.dst.data_ptr(), g.E, _h(dM, "dM"),
                                       _ld(dM), _stream()), "yolat_csr_mean_bwd_h")
        return dM
    check(lib.yolat_csr_mean_bwd(_f(dOut), _ld(dOut), C, g.row_ptr.data_ptr(), g.dst.data_ptr(), g.E,
                                 _f(dM), _ld(dM), _stream()), "yolat_csr_mean_bwd")
    return dM


# ---------------------------------------------------------------------------------------------
# proposal pooling
# ---------------------------------------------------------------------------------------------

def segment_mean_fwd(X, g, Y, x_pro=None, x_relu=False):
    D = X.shape[1]
    xs, xb = (x_pro if x_pro is not None else (None, None))
    check(lib.yolat_segment_mean_fwd(_f(X), _ld(X), D, _f(xs, "x_scale", True), _f(xb, "x_shift", True),
                                     int(x_relu), g.seg_ptr.data_ptr(), g.P, _f(Y), _ld(Y), _stream()),
          "yolat_segment_mean_fwd")
    return Y


def segment_max_fwd(X, g, Y, arg=None, x_pro=None, x_relu=False):
    D = X.shape[1]
    xs, xb = (x_pro if x_pro is not None else (None, None))
    check(lib.yolat_segment_max_fwd(_f(X), _ld(X), D, _f(xs, "x_scale", True), _f(xb, "x_shift", True),
                                    int(x_relu), g.seg_ptr.data_ptr(), g.P, g.N, _f(Y), _ld(Y),
                                    _i(arg, torch.int32, "arg"), _stream()), "yolat_segment_max_fwd")
    return Y


def segment_mean_bwd(dY, g, dX):
    D = dX.shape[1]
    check(lib.yolat_segment_mean_bwd(_f(dY), _ld(dY), D, g.seg_ptr.data_ptr(), g.node_seg.data_ptr(),
                                     g.N, _f(dX), _ld(dX), _stream()), "yolat_segment_mean_bwd")
    return dX


def segment_max_bwd(dY, arg, g, dX):
    D = dX.shape[1]
    check(lib.yolat_segment_max_bwd(_f(dY), _ld(dY), D, _i(arg, torch.int32, "arg"),
                                    g.node_seg.data_ptr(), g.N, _f(dX), _ld(dX), _stream()),
          "yolat_segment_max_bwd")
    return dX


# ---------------------------------------------------------------------------------------------
# device-side sub-batch extraction for predict()'s two passes (csrc/subgraph.hip)
# ---------------------------------------------------------------------------------------------

def _ranges_to_ids(starts, ends, dev):
    """Concatenation of range(s, e) as an int32 CUDA tensor (ranges given as host int64 arrays)."""
    import numpy as np
    lens = ends - starts
    if (lens < 0).any():
        raise ValueError("idx range with end < start")
    prefix = np.zeros(len(lens) + 1, dtype=np.int64)
    np.cumsum(lens, out=prefix[1:])
    total = int(prefix[-1])
    out = torch.empty(total, dtype=torch.int32, device=dev)
    if total:
        st = torch.from_numpy(starts.astype(np.int32)).to(dev, non_blocking=True)
        pf = torch.from_numpy(prefix.astype(np.int32)).to(dev, non_blocking=True)
        check(lib.yolat_expand_ranges(st.data_ptr(), pf.data_ptr(), len(lens), total, out.data_ptr(), _stream()),
              "yolat_expand_ranges")
        out._keep = (st, pf)
    return out


def _gather(src, idx):
    """dst[r] = src[idx[r]] for a contiguous 2-D tensor of a 4-/8-byte dtype."""
    if not src.is_contiguous() or src.dim() != 2:
        raise ValueError("gather source must be a contiguous 2-D tensor")
    rb = src.shape[1] * src.element_size()
    dst = torch.empty(idx.numel(), src.shape[1], dtype=src.dtype, device=src.device)
    check(lib.yolat_gather_rows_bytes(src.data_ptr(), rb, idx.data_ptr(), idx.numel(), rb, dst.data_ptr(), rb, _stream()),
          "yolat_gather_rows_bytes")
    return dst


def extract_subgraph(data_cls, dev_batch, pos_s, pos_e, edge_s, edge_e, slice_bbox):
    """`build_data` of arch:166-242 on the device.  dev_batch: dict of CUDA tensors x, pos, edge [E,2] int64,
    e_attr, bbox_idx int64, bbox, stat_feats.  Returns (Data of CUDA tensors, status word)."""
    import numpy as np
    x, edge, bbox_idx = dev_batch["x"], dev_batch["edge"], dev_batch["bbox_idx"]
    dev = x.device
    N = x.shape[0]
    node_ids = _ranges_to_ids(pos_s, pos_e, dev)
    edge_ids = _ranges_to_ids(edge_s, edge_e, dev)
    bbox_ids = torch.from_numpy(np.asarray(slice_bbox, dtype=np.int32)).to(dev, non_blocking=True)
    n_sub, m_sub = node_ids.numel(), edge_ids.numel()
    new = data_cls(x=_gather(x, node_ids), pos=_gather(dev_batch["pos"], node_ids))
    new.e_attr = _gather(dev_batch["e_attr"], edge_ids)
    new.bbox = _gather(dev_batch["bbox"], bbox_ids)
    new.stat_feats = _gather(dev_batch["stat_feats"], bbox_ids)
    new.edge = torch.empty(m_sub, 2, dtype=torch.int64, device=dev)
    new.bbox_idx = torch.empty(n_sub, dtype=torch.int64, device=dev)
    status = torch.zeros(1, dtype=torch.int32, device=dev)
    work = torch.empty(int(lib.yolat_subgraph_work_elems(N, n_sub)), dtype=torch.int32, device=dev)
    if edge.dim() != 2 or edge.shape[1] != 2:
        raise ValueError("edge must be [E,2]")
    check(lib.yolat_subgraph_reindex(node_ids.data_ptr(), n_sub, N, _i(edge, torch.int64, "edge"), edge.stride(0),
                                     edge.stride(1), edge_ids.data_ptr(), m_sub, _i(bbox_idx, torch.int64, "bbox_idx"),
                                     new.edge.data_ptr(), new.bbox_idx.data_ptr(), work.data_ptr(), status.data_ptr(),
                                     _stream()), "yolat_subgraph_reindex")
    new._keep = (node_ids, edge_ids, bbox_ids, work)
    return new, status


# ---------------------------------------------------------------------------------------------
# training-mode fusion block + per-proposal max pooling (csrc/fusion_train.hip)
# ---------------------------------------------------------------------------------------------

def fusion_pool_train_fwd(A, lin, bn, g, Z):
    """Z[P,F] <- scatter_max(relu(bn(lin(A)))) with batch statistics, without materialising [N,F].
    Returns the state the backward needs."""
    N, K = A.shape
    F = lin.out_features
    dev = A.device
    coef = torch.empty(4, F, dtype=torch.float32, device=dev)
    saved = torch.empty(int(lib.yolat_fusion_pool_train_saved_elems(K, F, g.P)), dtype=torch.float32, device=dev)
    work = torch.empty(int(lib.yolat_fusion_pool_train_work_elems(N, K, F, g.P)), dtype=torch.float32, device=dev)
    track = bn.track_running_stats and bn.running_mean is not None
    mom = 0.1 if bn.momentum is None else float(bn.momentum)
    W = lin.weight
    if not W.is_contiguous():
        raise ValueError("fusion_block weight must be contiguous")
    if track:
        bump_weight_epoch()
    check(lib.yolat_fusion_pool_train_fwd(_f(A, "A"), _ld(A), N, K, _f(W), _f(lin.bias, "bias", True), F,
                                          _f(bn.weight), _f(bn.bias), _f(bn.running_mean if track else None, "rm", True),
                                          _f(bn.running_var if track else None, "rv", True), mom, float(bn.eps),
                                          g.node_seg.data_ptr(), g.P, _f(Z), _ld(Z), _f(coef), _f(saved), _f(work),
                                          _stream()), "yolat_fusion_pool_train_fwd")
    return {"A": A, "lin": lin, "bn": bn, "coef": coef, "saved": saved, "work": work}


def fusion_pool_train_bwd(sv, g, gZ, dW, dbias, dgamma, dbeta, dA, side=None):
    """`side` (engine._on_side): the weight gradient (sparse gather + finish, ~140 us at cfg 3) on a second stream
    beside the input gradient; the per-column reductions both read run first on the caller's stream."""
    A, lin = sv["A"], sv["lin"]
    N, K = A.shape
    F = lin.out_features

    def part(mask):
        check(lib.yolat_fusion_pool_train_bwd_parts(_f(A), _ld(A), N, K, _f(lin.weight), _f(sv["bn"].weight), F,
                                                    _f(sv["coef"]), _f(sv["saved"]), g.node_seg.data_ptr(),
                                                    g.seg_ptr.data_ptr(), g.P, _f(gZ, "gZ"), _ld(gZ), _f(dW),
                                                    _f(dbias, "dbias", True), _f(dgamma), _f(dbeta), _f(dA), _ld(dA),
                                                    _f(sv["work"]), mask, _stream()), "yolat_fusion_pool_train_bwd")
    if side is None:
        part(7)
        return
    part(1)
    side(lambda: part(2), (A, gZ, sv["saved"], sv["work"], sv["coef"], g.node_seg))
    part(4)


# ---------------------------------------------------------------------------------------------
# loss / optimiser
# ---------------------------------------------------------------------------------------------

def softmax_ce(logits, labels, loss, dlogits=None):
    P, K = logits.shape
    work = torch.empty(int(lib.yolat_softmax_ce_work_elems(P)), dtype=torch.float32, device=logits.device)
    check(lib.yolat_softmax_ce(_f(logits), _ld(logits), _i(labels, torch.int64, "labels"), P, K,
                               _f(loss), _f(dlogits, "dlogits", True),
                               _ld(dlogits) if dlogits is not None else K, work.data_ptr(), _stream()),
          "yolat_softmax_ce")
    return loss


def adam_step(param, grad, exp_avg, exp_avg_sq, lr, beta1, beta2, eps, weight_decay, step,
              grad_scale=1.0):
    bump_weight_epoch()
    check(lib.yolat_adam_step(_f(param), _f(grad), _f(exp_avg), _f(exp_avg_sq), param.numel(),
                              float(lr), float(beta1), float(beta2), float(eps), float(weight_decay),
                              int(step), float(grad_scale), _stream()), "yolat_adam_step")


def dropout_fwd(Y, scale, shift, relu, p, seed, Z):
    """Z = relu?(Y*scale+shift) * keep/(1-p) with an element-wise Bernoulli(1-p) mask (yolat_dropout_fwd); returns the
    uint8 mask for dropout_bwd."""
    M, C = Y.shape
    mask = torch.empty(M * C, dtype=torch.uint8, device=Y.device)
    check(lib.yolat_dropout_fwd(_f(Y), _ld(Y), M, C, _f(scale, "scale", True), _f(shift, "shift", True), int(relu),
                                float(p), int(seed), mask.data_ptr(), _f(Z), _ld(Z), _stream()), "yolat_dropout_fwd")
    return mask


def dropout_bwd(dZ, mask, p, dX):
    M, C = dZ.shape
    check(lib.yolat_dropout_bwd(_f(dZ), _ld(dZ), M, C, mask.data_ptr(), float(p), _f(dX), _ld(dX), _stream()),
          "yolat_dropout_bwd")
    return dX


def nms(boxes, scores, iou_threshold):
    """torchvision.ops.nms(boxes, scores, iou_threshold) on the HIP path (csrc/nms.hip, yolat_nms): indices of the
    kept boxes in descending score order.  boxes [n,4] (x1,y1,x2,y2), scores [n]; both CUDA tensors."""
    n = int(boxes.shape[0])
    if boxes.dim() != 2 or boxes.shape[1] != 4 or scores.dim() != 1 or scores.shape[0] != n:
        raise ValueError("nms expects boxes [n,4] and scores [n]")
    if n == 0:
        return torch.empty(0, dtype=torch.int64, device=boxes.device)
    b = boxes.detach().to(torch.float32).contiguous()
    s = scores.detach().to(torch.float32).contiguous()
    if not b.is_cuda or not s.is_cuda:
        raise ValueError("nms expects CUDA tensors")
    need = int(lib.yolat_nms_work_bytes(n))
    if need == 0:
        raise ValueError("nms supports up to 524288 boxes")
    work = torch.empty(need, dtype=torch.uint8, device=b.device)
    keep = torch.empty(n, dtype=torch.int64, device=b.device)
    cnt = torch.empty(1, dtype=torch.int32, device=b.device)
    check(lib.yolat_nms(b.data_ptr(), s.data_ptr(), n, float(iou_threshold), keep.data_ptr(), cnt.data_ptr(),
                        work.data_ptr(), work.numel(), _stream()), "yolat_nms")
    return keep[:int(cnt.item())]          # D2H sync, as torchvision's sized result implies
