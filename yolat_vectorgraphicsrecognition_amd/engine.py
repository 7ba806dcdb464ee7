"""Hand-scheduled forward / backward of the YOLaT hot path on top of the C ABI (ops.py).

There is no tracing compiler and no autograd tape inside the model: each block below launches its
kernels in a fixed order and keeps exactly the tensors its backward needs.

Lazy activations.  A training-mode ``Linear -> BatchNorm1d -> ReLU`` (gcn_lib/sparse/torch_nn.py:58-66)
needs the statistics of *all* rows before it can normalise, so the GEMM stores the pre-activation
and a per-column (scale, shift); the consumer applies ``relu(scale*y+shift)`` while it loads its
operand ("prologue").  ``Lazy`` is that pair.  In eval mode BatchNorm is folded into the GEMM
epilogue and ``Lazy.scale`` is None (materialised).

Blocks
    conv_fwd / conv_bwd       AttrRelativeEdgeConvGlobalPool2      torch_vertex.py:288-341
    lbr_fwd / lbr_bwd         Linear(+BN+ReLU)                      torch_nn.py:50-71
    model_fwd / model_bwd     SparseCADGCN.forward                  architecture3cc_rpn_gp_iter2.py:44-71,106-137
"""
import os

import torch

from . import ops

# training-mode fusion block + max pooling without the [N, 1024] activation (csrc/fusion_train.hip);
# FUSED_FUSION_TRAIN = False selects the materialising schedule (kept as the cross-check in the tests)
# (module flags, flipped in-process by the tests that cross-check the schedules against each other; no environment switches)
FACTORISED_TRAIN = True
# edges per node from which the factorised first edge Linear is used (forward / backward chosen separately)
FACT_FWD_RATIO = 2.0
FACT_BWD_RATIO = 1.0     # measured: cfg 3 (E = 1.2 N) 3.77 -> 3.66 ms, cfg 4 3.63 -> 3.53 ms
FUSED_FUSION_TRAIN = True


class Lazy(object):
    __slots__ = ("t", "scale", "shift", "relu")

    def __init__(self, t, scale=None, shift=None, relu=False):
        self.t, self.scale, self.shift, self.relu = t, scale, shift, relu

    @property
    def pro(self):
        return None if self.scale is None else (self.scale, self.shift)

    def materialise(self, out=None):
        if self.scale is None and not self.relu:
            return self.t
        out = torch.empty_like(self.t) if out is None else out
        ops.scale_shift_relu(self.t, self.scale, self.shift, self.relu, out)
        return out


def _empty(rows, cols, dev):
    return torch.empty(rows, cols, dtype=torch.float32, device=dev)


def _vec(n, dev):
    return torch.empty(n, dtype=torch.float32, device=dev)


class GradSink(object):
    """Where parameter gradients are written: fresh tensors (autograd path) or views of one flat
    buffer (trainer / data-parallel path)."""

    def __init__(self, views=None, on_head_done=None):
        self.views = views          # dict id(param) -> tensor view, or None
        self.out = {}
        # called once by model_bwd when every gradient EXCEPT the conv layers' is final (classifier, fusion
        # blocks = 93 % of the parameters): the data-parallel trainer starts their all-reduce there, so that it
        # overlaps the conv-layer backward
        self.on_head_done = on_head_done

    def get(self, p):
        if self.views is not None:
            t = self.views[id(p)]
        else:
            t = torch.empty_like(p)
        self.out[id(p)] = t
        return t


# ---------------------------------------------------------------------------------------------
# BatchNorm coefficient helpers
# ---------------------------------------------------------------------------------------------

def _bn_train(stats, M, bn, dev, coef_out=None):
    """Returns the four coefficient vectors (scale, shift, mean, invstd), indexable 0..3.  ``coef_out`` = (scale, shift)
    destinations the caller wants them in (slices of a wider vector: the concatenated prologue of the per-proposal mean
    over the node branches) — written there directly instead of copied afterwards."""
    C = bn.num_features
    coef = torch.empty(4, C, dtype=torch.float32, device=dev)   # scale, shift, mean, invstd
    if coef_out is not None:
        coef = (coef_out[0], coef_out[1], coef[2], coef[3])
    ops.bn_finalize(stats, M, bn, coef[0], coef[1], coef[2], coef[3])
    if bn.num_batches_tracked is not None:
        _PENDING_NBT.append(bn.num_batches_tracked)
    return coef


# BatchNorm1d.num_batches_tracked += 1 for every layer of a training forward, as ONE multi-tensor kernel at the
# end of the forward (ten separate int64 adds were 45 us of a 4.2 ms cfg-3 step)
_PENDING_NBT = []


def flush_batch_counters():
    if _PENDING_NBT:
        torch._foreach_add_(_PENDING_NBT, 1)
        del _PENDING_NBT[:]


def _bn_eval(bn, dev):
    """Folded eval coefficients, cached on the module until any of its tensors changes (torch-side edits show up
    in `_version`, raw-pointer writes of the HIP kernels — Adam, running statistics — in ops.weight_epoch())."""
    key = (bn.weight._version, bn.bias._version, bn.running_mean._version, bn.running_var._version,
           bn.weight.data_ptr(), bn.running_mean.data_ptr(), ops.weight_epoch())
    cache = getattr(bn, "_yolat_eval", None)
    if cache is not None and cache[0] == key:
        return cache[1]
    coef = torch.empty(2, bn.num_features, dtype=torch.float32, device=dev)
    ops.bn_eval_coeffs(bn, coef[0], coef[1])
    bn._yolat_eval = (key, coef)
    return coef


# ---------------------------------------------------------------------------------------------
# Linear (+ BatchNorm1d + ReLU)
# ---------------------------------------------------------------------------------------------

def lbr_fwd(a, lin, bn, relu, training, out=None, coef_out=None):
    """a: Lazy input [M,K].  Returns (Lazy output, saved).  ``out`` optionally names the [M,C]
    destination (may be a column slice of a wider buffer)."""
    A = a.t
    M, dev = A.shape[0], A.device
    C = lin.out_features
    y = _empty(M, C, dev) if out is None else out
    sv = {"a": a, "lin": lin, "bn": bn, "relu": relu, "y": y}
    if bn is None:
        ops.linear_fwd(A, lin.weight, lin.bias, y, a_pro=a.pro, a_relu=a.relu, o_relu=relu)
        return Lazy(y), sv
    if training:
        if M == 0:
            raise ValueError("BatchNorm1d in training mode needs at least one row")
        stats = ops.stats_buffer(M, C, dev)
        ops.linear_fwd(A, lin.weight, lin.bias, y, a_pro=a.pro, a_relu=a.relu, stats=stats)
        coef = _bn_train(stats, M, bn, dev, coef_out)
        sv["coef"] = coef
        return Lazy(y, coef[0], coef[1], relu), sv
    coef = _bn_eval(bn, dev)
    ops.linear_fwd(A, lin.weight, lin.bias, y, a_pro=a.pro, a_relu=a.relu, o_pro=(coef[0], coef[1]),
                   o_relu=relu)
    return Lazy(y), sv


def dropout_fwd(a, p):
    """Training-mode nn.Dropout2d(p) behind a Linear+BN+ReLU block (torch_nn.py:67-68): materialises the lazy
    activation with the element-wise mask applied.  Returns (Lazy output, saved (mask, p))."""
    M, C = a.t.shape
    z = _empty(M, C, a.t.device)
    seed = int(torch.randint(0, 2 ** 62, (1,)).item())          # host generator: follows torch.manual_seed
    mask = ops.dropout_fwd(a.t, a.scale, a.shift, a.relu, p, seed, z)
    return Lazy(z), (mask, p)


def dropout_bwd(saved, dz):
    """dz (gradient w.r.t. the dropped-out activation) -> gradient w.r.t. the block's post-activation output,
    in place."""
    mask, p = saved
    return ops.dropout_bwd(dz, mask, p, dz)


def _drop_p(mlp):
    """p of the nn.Dropout2d a gcn_lib MLP ends with (0 when there is none)."""
    last = list(mlp.children())[-1]
    return float(last.p) if last.__class__.__name__.startswith("Dropout") else 0.0


# ---------------------------------------------------------------------------------------------
# Weight gradients on a second stream
# ---------------------------------------------------------------------------------------------
# In a Linear's backward dW = dY^T . A and dX = dY . W depend on the same dY and on nothing of each other, and nothing
# in the rest of the backward reads dW: it is only consumed by the gradient exchange / Adam.  Both are tall-skinny
# launches far from saturating the chip, so dW goes to a side stream — ordered after everything issued
# so far on the current stream by an event — and runs beside the dX chain; the streams join before the first consumer
# (model_bwd: the head bucket's all-reduce, the end of the backward).  Tensors handed to the side stream are marked
# with record_stream so that the caching allocator does not recycle them while it still reads them.
SIDE_STREAM = True           # module flag (tests/test_gpu_model.py flips it: both schedules bit-identical)
_SIDE = {}


SIDE_PROBE = {}              # device index -> what the one-time probe below found (diagnostics, tests)


def _pick_side_stream(cur):
    """A stream that really runs BESIDE `cur`.  The runtime multiplexes its streams onto a few hardware queues
    (GPU_MAX_HW_QUEUES = 4), torch hands out streams from a pool of 32 round-robin, and two streams that share a hardware
    queue run one after the other: whether the n-th pool stream shares `cur`'s queue depends on how many streams the process
    has drawn before (bench.py's train legs ran at the ONE-stream time, 3.25 vs 2.80 ms at cfg 3, after the hand-over legs had
    drawn three more).  So the first use probes: a spin of a fixed number of cycles on `cur` alone, then on `cur` and on the
    candidate together — a candidate that doubles the time is discarded (one pool stream in four does; up to 8 are tried,
    a few ms, once per device)."""
    sleep = getattr(torch.cuda, "_sleep", None)
    if sleep is None:
        return torch.cuda.Stream(device=cur.device)
    cycles = 400000
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)

    def timed(cand):
        torch.cuda.synchronize(cur.device)
        e0.record(cur)
        if cand is not None:            # the candidate's spin is ordered behind e0 and issued first: both start together
            cand.wait_stream(cur)
            with torch.cuda.stream(cand):
                sleep(cycles)
        sleep(cycles)
        if cand is not None:
            cur.wait_stream(cand)
        e1.record(cur)
        e1.synchronize()
        return e0.elapsed_time(e1)

    timed(None)
    alone = min(timed(None), timed(None))
    tried = []
    best = None
    for _ in range(8):
        cand = torch.cuda.Stream(device=cur.device)
        timed(cand)                     # (a stream's first submission sets its queue up: milliseconds)
        both = min(timed(cand), timed(cand))
        tried.append(round(both / alone, 2))
        if best is None or both < best[0]:
            best = (both, cand)
        if both < 1.5 * alone:
            break
    SIDE_PROBE[cur.device_index] = {"alone_ms": alone, "pair_over_alone": tried}
    return best[1]


def side_stream_for(cur):
    """the side stream of `cur`'s device (picked by the probe on first use) — also what trainer.TrainPlan hands to
    yolat_train_step"""
    ent = _SIDE.get(cur.device_index)
    if ent is None:
        ent = _SIDE[cur.device_index] = {"stream": _pick_side_stream(cur), "dirty": False}
    return ent["stream"]


def _on_side(fn, keep):
    if not SIDE_STREAM or torch.cuda.is_current_stream_capturing():
        return fn()
    cur = ops.current_stream_object()
    side_stream_for(cur)
    ent = _SIDE[cur.device_index]
    side = ent["stream"]
    side.wait_stream(cur)
    _set = torch._C._cuda_setStream
    _set(stream_id=side.stream_id, device_index=side.device_index, device_type=side.device_type)
    try:
        res = fn()
    finally:
        _set(stream_id=cur.stream_id, device_index=cur.device_index, device_type=cur.device_type)
    for t in keep:
        if t is not None:
            t.record_stream(side)
    ent["dirty"] = True
    return res


def _adopt(*tensors):
    """Tensors allocated while the side stream was current (its allocator pool) that live on into work of the current
    stream: tell the allocator."""
    if not SIDE_STREAM:
        return
    cur = ops.current_stream_object()
    for t in tensors:
        if isinstance(t, (tuple, list)):
            _adopt(*t)
        elif t is not None:
            t.record_stream(cur)


# Fault injection for tests/test_gpu_dist.py ONLY: issue the head bucket's exchange at the START of the backward, before
# any of its gradients exists — the program-order form of "the exchange does not wait for the gradients".  (Removing the
# join in front of the exchange alone does not reproduce on demand: measured with the side stream held back by 10 ms per
# submission, the step stayed bit-identical — the caching allocator's device-wide synchronisations, whenever a block
# recorded on the side stream cannot be reused yet, serialise the two streams in practice.  The join stays: correctness
# must not depend on that.)
_FAULT_EARLY_HEAD_EXCHANGE = False


class fault_early_head_exchange(object):
    """`with engine.fault_early_head_exchange():` — the only way the fault is switched on (tests/test_gpu_dist.py): it is off
    again however the block ends, and Trainer.step refuses to run with it outside a doubled-bucket (exchange_premul) test."""

    def __enter__(self):
        global _FAULT_EARLY_HEAD_EXCHANGE
        _FAULT_EARLY_HEAD_EXCHANGE = True
        return self

    def __exit__(self, *exc):
        global _FAULT_EARLY_HEAD_EXCHANGE
        _FAULT_EARLY_HEAD_EXCHANGE = False
        return False


def _join_side():
    """The current stream waits for the side-stream work issued so far."""
    cur = ops.current_stream_object()
    ent = _SIDE.get(cur.device_index)
    if ent is not None and ent["dirty"]:
        cur.wait_stream(ent["stream"])
        ent["dirty"] = False


def lbr_bwd(sv, dz, sink, dx_out=None, dx_accumulate=False, need_dx=True, dz_inplace=True):
    """dz: gradient w.r.t. the block's (post-activation) output [M,C].  Writes parameter gradients
    into ``sink``; returns the gradient w.r.t. the block's *post-prologue* input (i.e. w.r.t. the
    producer's post-activation output), written to ``dx_out`` if given."""
    a, lin, bn, relu, y = sv["a"], sv["lin"], sv["bn"], sv["relu"], sv["y"]
    M, dev = y.shape[0], y.device
    if bn is not None:
        coef = sv["coef"]
        dy = dz if dz_inplace else torch.empty(M, y.shape[1], dtype=torch.float32, device=dev)
        ops.bn_relu_bwd(dz, y, bn.weight, coef[2], coef[3], coef[0], coef[1], relu,
                        sink.get(bn.weight), sink.get(bn.bias), dy)
    else:
        if relu:
            raise NotImplementedError("Linear+ReLU without BatchNorm is not on the reference path")
        dy = dz
    db = sink.get(lin.bias) if lin.bias is not None else None
    dW = sink.get(lin.weight)
    _on_side(lambda: ops.linear_bwd_w(dy, a.t, dW, db, a_pro=a.pro, a_relu=a.relu), (dy, a.t, a.scale, a.shift))
    if not need_dx:
        return None
    dx = _empty(M, lin.in_features, dev) if dx_out is None else dx_out
    ops.linear_fwd_wt(dy, lin.weight, dx, accumulate=dx_accumulate)
    return dx


# ---------------------------------------------------------------------------------------------
# AttrRelativeEdgeConvGlobalPool2
# ---------------------------------------------------------------------------------------------

FUSED_BN_CSR_BWD = True
# the BatchNorm-1 backward apply pass also takes dU, dWc4 and db1 of the factorised first edge Linear (round 4)
FUSED_BN_APPLY_SUMS = True


def conv_fwd(conv, g, x, xn, out_f, out_s, training, half=False, node_coef_out=None):
    """x [N,Cin] materialised; xn Lazy [N,Cin]; out_f / out_s: [N,C] destinations (or None).
    Returns (f tensor, s Lazy, saved)."""
    nn0, bn1, nn3, bn4 = conv.nn[0], conv.nn[1], conv.nn[3], conv.nn[4]
    N, dev = x.shape[0], x.device
    C = nn0.out_features
    E = g.E
    out_f = _empty(N, C, dev) if out_f is None else out_f
    sv = {"conv": conv, "x": x, "E": E}
    # root term first, the aggregation accumulates onto it:  out = lin_r(x) + mean_e(m_e)   (:325)
    ops.linear_fwd(x, conv.lin_r.weight, conv.lin_r.bias, out_f)
    if E > 0:
        factorised = (training and FACTORISED_TRAIN and C == 64 and E >= FACT_FWD_RATIO * N and nn0.weight.is_contiguous()
                      and nn0.bias is not None)
        # bf16 STORAGE of the two [E,C] activations (and, in conv_bwd, of their gradients): halves the traffic of the
        # bandwidth-bound part of the step; accumulation, statistics, parameters and node tensors stay fp32
        hdt = torch.bfloat16 if (half and factorised) else torch.float32
        H1 = torch.empty(E, C, dtype=hdt, device=dev)
        H2 = torch.empty(E, C, dtype=hdt, device=dev)
        if training:
            st1 = ops.stats_buffer(E, C, dev)
            if factorised:
                # per-node products + gather-add instead of the gathered K = 132 GEMM (pays when E >> N)
                ops.edge_lin1_fwd_factorised(x, g, nn0.weight, nn0.bias, H1, stats=st1, keep=sv)
            else:
                ops.edge_lin1_fwd(x, g, nn0.weight, nn0.bias, H1, stats=st1)
            c1 = _bn_train(st1, E, bn1, dev)
            st2 = ops.stats_buffer(E, C, dev)
            ops.linear_fwd(H1, nn3.weight, nn3.bias, H2, a_pro=(c1[0], c1[1]), a_relu=True, stats=st2)
            c2 = _bn_train(st2, E, bn4, dev)
            ops.csr_mean_fwd(H2, g, out_f, h_pro=(c2[0], c2[1]), h_relu=True, accumulate=True)
            sv.update(H1=H1, H2=H2, c1=c1, c2=c2)
        else:
            c1, c2 = _bn_eval(bn1, dev), _bn_eval(bn4, dev)
            ops.edge_lin1_fwd(x, g, nn0.weight, nn0.bias, H1, o_pro=(c1[0], c1[1]), o_relu=True)
            ops.linear_fwd(H1, nn3.weight, nn3.bias, H2, o_pro=(c2[0], c2[1]), o_relu=True)
            ops.csr_mean_fwd(H2, g, out_f, accumulate=True)
    # node branch (mlp_node: Linear + BatchNorm + ReLU on the previous layer's node branch): independent of this layer's
    # edge side and read again only by the next layer's node branch and, at the very end, by the per-proposal mean — in
    # training it runs on the side stream beside the following edge-side kernels (model_fwd joins before the mean)
    if training:
        s, sv_n = _on_side(lambda: lbr_fwd(xn, conv.mlp_node[0], conv.mlp_node[1], True, training, out=out_s,
                                           coef_out=node_coef_out), (xn.t, xn.scale, xn.shift))
        _adopt(sv_n["y"], sv_n.get("coef"))
    else:
        s, sv_n = lbr_fwd(xn, conv.mlp_node[0], conv.mlp_node[1], True, training, out=out_s, coef_out=node_coef_out)
    sv["node"] = sv_n
    return out_f, s, sv


def conv_bwd(sv, g, d_f, d_s, sink, dx=None, dx_acc=False, dxn=None, dxn_acc=False, need_dx=True):
    """d_f [N,C]: grad w.r.t. out;  d_s [N,C]: grad w.r.t. the node-branch post-activation output.
    If need_dx: accumulates/writes grad w.r.t. x into ``dx`` and w.r.t. the (post-activation) x_node
    into ``dxn``.  d_s is consumed in place."""
    conv, x, E = sv["conv"], sv["x"], sv["E"]
    nn0, bn1, nn3, bn4 = conv.nn[0], conv.nn[1], conv.nn[3], conv.nn[4]
    N, Cin = x.shape
    C = nn0.out_features
    dev = x.device
    # node branch: its backward chain (BatchNorm+ReLU backward, dW, dX) feeds nothing but the previous layer's node branch,
    # so across the layers it forms a chain of its own: on the side stream, beside the edge-side backward below
    svn, dxn_dst = sv["node"], dxn
    dxn = _on_side(lambda: lbr_bwd(svn, d_s, sink, dx_out=dxn_dst, dx_accumulate=dxn_acc, need_dx=need_dx), (d_s,))
    _adopt(dxn)
    # root term
    dWr, dbr = sink.get(conv.lin_r.weight), sink.get(conv.lin_r.bias)
    _on_side(lambda: ops.linear_bwd_w(d_f, x, dWr, dbr), (d_f, x))
    if need_dx:
        dx = _empty(N, Cin, dev) if dx is None else dx
        ops.linear_fwd_wt(d_f, conv.lin_r.weight, dx, accumulate=dx_acc)
    if E > 0:
        H1, H2, c1, c2 = sv["H1"], sv["H2"], sv["c1"], sv["c2"]
        dA1 = torch.empty(E, C, dtype=H1.dtype, device=dev)
        bn1_done = False
        partial, db1 = None, None
        fact_bwd = H1.dtype == torch.bfloat16 or (FACTORISED_TRAIN and C == 64 and E >= FACT_BWD_RATIO * N
                                                  and nn0.weight.is_contiguous() and nn0.bias is not None)
        one_kernel = C == 64 and nn3.in_features == 64 and nn3.weight.is_contiguous()
        # the fused kernels read the parameter / coefficient vectors with 16-byte loads: views of a flat parameter buffer
        # start wherever the preceding parameters end, so the alignment is part of the gate (else: the materialising path)
        aligned = all(t.data_ptr() % 16 == 0 for t in (nn3.weight, d_f, H1, H2, c1[0], c1[1], c1[2], c1[3], c2[0], c2[1],
                                                       c2[2], c2[3]))
        if (FUSED_BN_CSR_BWD and aligned and C % 4 == 0 and d_f.stride(0) % 4 == 0
                and (H1.dtype == torch.float32 or one_kernel)):
            # the gradient w.r.t. H2 (mean aggregation -> ReLU -> BatchNorm backward) is formed inside its two consumers
            # instead of being written and re-read: 5 instead of 11 passes over [E,C] (bn_csr.hip)
            dh2 = ops.BnCsrGrad(d_f, g, H2, c2[2], c2[3], c2[0], c2[1], relu=True)
            dh2.stats(sink.get(bn4.weight), sink.get(bn4.bias))
            if one_kernel:
                # ... which also takes the statistics of BatchNorm 1's backward on dA1: only its apply pass is left
                coef1 = dh2.bwd_w_and_x(H1, nn3.weight, sink.get(nn3.weight), sink.get(nn3.bias), dA1,
                                        a_pro=(c1[0], c1[1]), a_relu=True,
                                        next_bn=(c1[2], c1[3], sink.get(bn1.weight), sink.get(bn1.bias)))
                if fact_bwd and FUSED_BN_APPLY_SUMS and g.attr.data_ptr() % 16 == 0:
                    # the apply pass forms dH1 row by row in CSR order: its per-node sums (dU), the attr weight gradient and
                    # db1 are taken while the rows are in registers — dH1 is read again only by the gathered dV sums
                    db1 = sink.get(nn0.bias)              # ONE get per parameter: the autograd sink hands out a fresh tensor each time
                    partial = ops.bn_apply_edge_sums(dA1, H1, c1[2], c1[3], c1[0], c1[1], True, coef1, g, db1)
                else:
                    ops.bn_relu_bwd_apply(dA1, H1, c1[2], c1[3], c1[0], c1[1], True, coef1, dA1)
                bn1_done = True
            else:
                dh2.bwd_w(H1, sink.get(nn3.weight), sink.get(nn3.bias), a_pro=(c1[0], c1[1]), a_relu=True)
                dh2.fwd_wt(nn3.weight, dA1)
        else:
            dM = torch.empty(E, C, dtype=H1.dtype, device=dev)
            ops.csr_mean_bwd(d_f, g, dM)
            ops.bn_relu_bwd(dM, H2, bn4.weight, c2[2], c2[3], c2[0], c2[1], True,
                            sink.get(bn4.weight), sink.get(bn4.bias), dM)            # dM -> dH2 in place
            ops.linear_bwd_w(dM, H1, sink.get(nn3.weight), sink.get(nn3.bias), a_pro=(c1[0], c1[1]), a_relu=True)
            ops.linear_fwd_wt(dM, nn3.weight, dA1)
        if not bn1_done:
            ops.bn_relu_bwd(dA1, H1, bn1.weight, c1[2], c1[3], c1[0], c1[1], True,
                            sink.get(bn1.weight), sink.get(bn1.bias), dA1)           # dA1 -> dH1 in place
        if fact_bwd:
            # per-node sums of dH1 + N-row dense algebra instead of the gathered E-row GEMMs (pays when E >> N)
            ops.edge_lin1_bwd_factorised(dA1, x, g, nn0.weight, sink.get(nn0.weight), db1 if db1 is not None else sink.get(nn0.bias),
                                         dx=dx if need_dx else None, dx_accumulate=True, side=_on_side, partial=partial,
                                         wuv=sv.get("wuv"))
        else:
            ops.edge_lin1_bwd_w(dA1, x, g, sink.get(nn0.weight), sink.get(nn0.bias))
            if need_dx:
                dG = _empty(E, 2 * Cin, dev)
                ops.edge_lin1_bwd_x(dA1, nn0.weight, Cin, dG)
                ops.edge_scatter_bwd(dG, Cin, g, dx, accumulate=True)
    else:
        for p in (nn0.weight, nn0.bias, bn1.weight, bn1.bias, nn3.weight, nn3.bias, bn4.weight, bn4.bias):
            sink.get(p).zero_()
    return dx, dxn


# ---------------------------------------------------------------------------------------------
# whole model
# ---------------------------------------------------------------------------------------------

def model_convs(net):
    """The conv modules of a Backbone in layer order (head, then ResBlock bodies)."""
    return [net.head.gconv] + [blk.body.gconv for blk in net.backbone]


def model_fwd(model, g, x, training):
    """SparseCADGCN.forward on device tensors.  Returns (logits [P,K], saved-or-None)."""
    net = model.cls_net
    convs = model_convs(net)
    L, n_out = net.n_blocks, net.n_blocks_out
    lo = L - n_out
    N, dev = x.shape[0], x.device
    P = g.P
    C = convs[0].nn[0].out_features
    F = net.fusion_block[0].out_features
    half = model.__dict__.get("_yolat_train_precision", "fp32") == "bf16"
    D = C * n_out                       # fusion_dims

    feats = _empty(N, D, dev)           # cat of the last n_out conv outputs          (arch:60)
    fsup = _empty(N, D, dev)            # cat of the last n_out node-branch outputs   (arch:65)
    sup_coef = torch.empty(2, D, dtype=torch.float32, device=dev) if training else None
    sv = {"convs": [], "training": training}
    f, s = x, Lazy(x)
    for l, conv in enumerate(convs):
        slot = l - lo
        of = feats[:, slot * C:(slot + 1) * C] if slot >= 0 else None
        os_ = fsup[:, slot * C:(slot + 1) * C] if slot >= 0 else None
        # the node branch's BatchNorm coefficients of an output layer land in their slice of sup_coef (no copies)
        nco = ((sup_coef[0, slot * C:(slot + 1) * C], sup_coef[1, slot * C:(slot + 1) * C])
               if training and slot >= 0 else None)
        f, s, sv_c = conv_fwd(conv, g, f, s, of, os_, training, half=training and half, node_coef_out=nco)
        sv["convs"].append(sv_c)

    Z = _empty(P, 2 * (F + D), dev)     # [max(fusion) | max(feats) | fusion_super | mean(sup)]  (arch:127)
    # fusion block over nodes, then per-proposal max                                  (arch:61-63,122)
    arg_fus = None
    arg_feat = torch.empty(P, D, dtype=torch.int32, device=dev) if training else None
    if training and FUSED_FUSION_TRAIN and D == 128 and net.fusion_block[0].weight.is_contiguous():
        # no [N,F] activation: batch statistics from the Gram matrix of feats, extreme-of-z GEMM epilogue,
        # sparse backward (csrc/fusion_train.hip)
        sv_fus = ops.fusion_pool_train_fwd(feats, net.fusion_block[0], net.fusion_block[1], g, Z[:, 0:F])
        sv_fus["fused"] = True
        if net.fusion_block[1].num_batches_tracked is not None:
            _PENDING_NBT.append(net.fusion_block[1].num_batches_tracked)
    else:
        fus, sv_fus = lbr_fwd(Lazy(feats), net.fusion_block[0], net.fusion_block[1], True, training)
        arg_fus = torch.empty(P, F, dtype=torch.int32, device=dev) if training else None
        ops.segment_max_fwd(fus.t, g, Z[:, 0:F], arg_fus, x_pro=fus.pro, x_relu=fus.relu)
    ops.segment_max_fwd(feats, g, Z[:, F:F + D], arg_feat)
    # super branch: per-proposal mean, then fusion_block_super                        (arch:65-69)
    sup = Z[:, 2 * F + D:2 * F + 2 * D]
    if training:
        _join_side()                     # the node branches (fsup, sup_coef) were computed on the side stream
        ops.segment_mean_fwd(fsup, g, sup, x_pro=(sup_coef[0], sup_coef[1]), x_relu=True)
    else:
        ops.segment_mean_fwd(fsup, g, sup)
    fs, sv_fs = lbr_fwd(Lazy(sup), net.fusion_block_super[0], net.fusion_block_super[1], True, training,
                        out=(None if training else Z[:, F + D:2 * F + D]))
    if training:
        ops.scale_shift_relu(fs.t, fs.scale, fs.shift, True, Z[:, F + D:2 * F + D])
    # classifier                                                                      (arch:91-93,128)
    m1, m2, m3 = model.prediction_cls[0], model.prediction_cls[1], model.prediction_cls[2]
    c1, sv1 = lbr_fwd(Lazy(Z), m1[0], m1[1], True, training)
    c2, sv2 = lbr_fwd(c1, m2[0], m2[1], True, training)
    p_drop = _drop_p(m2) if training else 0.0                     # arch:92: only prediction_cls.1 carries dropout
    sv_drop = None
    if p_drop > 0:
        c2, sv_drop = dropout_fwd(c2, p_drop)
    logits, sv3 = lbr_fwd(c2, m3[0], None, False, training)
    if not training:
        return logits.t, None
    flush_batch_counters()
    sv.update(feats=feats, fsup=fsup, Z=Z, fus=sv_fus, fs=sv_fs, arg_fus=arg_fus, arg_feat=arg_feat,
              cls=(sv1, sv2, sv3), drop=sv_drop, dims=(N, P, C, F, D, L, lo))
    return logits.t, sv


def model_bwd(model, g, sv, dlogits, sink):
    """Backward of model_fwd (training mode).  dlogits [P,K] is consumed."""
    net = model.cls_net
    N, P, C, F, D, L, lo = sv["dims"]
    dev = dlogits.device
    if sink.on_head_done is not None and _FAULT_EARLY_HEAD_EXCHANGE:
        sink.on_head_done()                                           # (tests only: see _FAULT_EARLY_HEAD_EXCHANGE)
    sv1, sv2, sv3 = sv["cls"]
    d2 = lbr_bwd(sv3, dlogits, sink)
    if sv.get("drop") is not None:
        d2 = dropout_bwd(sv["drop"], d2)
    d1 = lbr_bwd(sv2, d2, sink)
    dZ = lbr_bwd(sv1, d1, sink)                                      # [P, 2(F+D)]
    # fusion_block_super: input = sup (Z[:, 2F+D:]), output post-activation = Z[:, F+D:2F+D]
    d_sup = dZ[:, 2 * F + D:2 * F + 2 * D]
    lbr_bwd(sv["fs"], dZ[:, F + D:2 * F + D], sink, dx_out=d_sup, dx_accumulate=True)
    d_fsup = _empty(N, D, dev)                                        # grad w.r.t. post-activation s's
    ops.segment_mean_bwd(d_sup, g, d_fsup)
    # fusion_block + max pooling
    d_feats = _empty(N, D, dev)
    ops.segment_max_bwd(dZ[:, F:F + D], sv["arg_feat"], g, d_feats)
    if sv["fus"].get("fused"):
        lin, bn = sv["fus"]["lin"], sv["fus"]["bn"]
        ops.fusion_pool_train_bwd(sv["fus"], g, dZ[:, 0:F], sink.get(lin.weight),
                                  sink.get(lin.bias) if lin.bias is not None else None, sink.get(bn.weight),
                                  sink.get(bn.bias), d_feats, side=_on_side)
    else:
        d_fus = _empty(N, F, dev)
        ops.segment_max_bwd(dZ[:, 0:F], sv["arg_fus"], g, d_fus)
        lbr_bwd(sv["fus"], d_fus, sink, dx_out=d_feats, dx_accumulate=True)
    if sink.on_head_done is not None and not _FAULT_EARLY_HEAD_EXCHANGE:
        _join_side()                                                  # the head bucket's gradients are complete
        sink.on_head_done()
    # conv layers, last to first
    d_f_next, d_s_next = None, None      # grads flowing into layer l's outputs from layer l+1
    for l in range(L - 1, -1, -1):
        slot = l - lo
        if slot >= 0:
            d_f = d_feats[:, slot * C:(slot + 1) * C]
            d_s = d_fsup[:, slot * C:(slot + 1) * C]
        else:
            d_f, d_s = d_f_next, d_s_next
        need_dx = l > 0
        dx = dxn = None
        acc = False
        if need_dx:
            pslot = l - 1 - lo
            if pslot >= 0:
                dx = d_feats[:, pslot * C:(pslot + 1) * C]
                dxn = d_fsup[:, pslot * C:(pslot + 1) * C]
                acc = True
        d_f_next, d_s_next = conv_bwd(sv["convs"][l], g, d_f, d_s, sink, dx=dx, dx_acc=acc, dxn=dxn,
                                      dxn_acc=acc, need_dx=need_dx)
    _join_side()                                                      # every gradient is complete on the caller's stream
    return sink
