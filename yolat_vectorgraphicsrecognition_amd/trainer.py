"""Training step + pure data parallelism for the YOLaT hot path.

* ``FlatParams`` — all parameters (and, separately, their gradients) live in ONE contiguous fp32
  buffer each; every ``nn.Parameter`` is a view.  Adam is then one kernel over ~1.6 M elements
  (``yolat_adam_step``) instead of ~50 per-tensor launches, and the data-parallel exchange is ONE
  RCCL all-reduce of a 6.45 MB buffer over xGMI per step (SURVEY.md §8 e).
* ``FlatAdam`` — ``torch.optim.Adam(lr, weight_decay)`` semantics of cad_recognition/train.py:212
  (L2 added to the gradient, betas (0.9, 0.999), eps 1e-8).
* ``Trainer.step(data)`` — train.py:263-284: zero_grad -> forward -> CE -> backward -> Adam, with the
  gradient all-reduce (mean over ranks) between backward and Adam when world_size > 1.

Data parallelism shards whole graphs (dataset items) across ranks; there are no cross-GPU edges
(Datasets/graph_dict3.py:594-600), BatchNorm statistics stay per replica (the reference has no
SyncBN), parameters and optimizer state are replicated.
"""
import os

import torch
import torch.distributed as dist

from . import ops
from .architecture import DetectionLoss


class FlatParams(object):
    def __init__(self, model):
        params = [p for p in model.parameters()]
        if not params:
            raise ValueError("model has no parameters")
        dev = params[0].device
        n = sum(p.numel() for p in params)
        self.numel = n
        self.param = torch.empty(n, dtype=torch.float32, device=dev)
        self.grad = torch.zeros(n, dtype=torch.float32, device=dev)
        self.grad_views = {}
        self.params = params
        off = 0
        for p in params:
            k = p.numel()
            self.param[off:off + k].copy_(p.data.reshape(-1))
            p.data = self.param[off:off + k].view(p.shape)
            gv = self.grad[off:off + k].view(p.shape)
            self.grad_views[id(p)] = gv
            p.grad = gv
            off += k
        self.direct_grads = True      # _ModelFn.backward writes straight into self.grad
        self.grads_ready = False
        # [0, conv_end) = parameters of the conv layers (cls_net.head / cls_net.backbone come first in
        # model.parameters()); [conv_end, n) = fusion blocks + classifier, whose gradients are final first
        self.conv_end = 0
        net = getattr(model, "cls_net", None)
        if net is not None:
            conv_ids = {id(p) for m in [net.head] + list(net.backbone) for p in m.parameters()}
            lead = 0
            while lead < len(params) and id(params[lead]) in conv_ids:
                lead += 1
            if 0 < lead < len(params) and not any(id(p) in conv_ids for p in params[lead:]):
                self.conv_end = sum(p.numel() for p in params[:lead])
        self.on_head_done = None
        model._yolat_flat = self


class FlatAdam(torch.optim.Optimizer):
    """``torch.optim.Adam(model.parameters(), lr, weight_decay)`` of cad_recognition/train.py:212 as ONE kernel over
    the flat buffers.  It is a ``torch.optim.Optimizer`` (one param group holding the model's parameters), so the
    reference's ``StepLR(optimizer, lr_adjust_freq, lr_decay_rate)`` (train.py:214) drives its learning rate, and
    ``state_dict()`` / ``load_state_dict()`` speak ``torch.optim.Adam``'s format — per-parameter ``step`` /
    ``exp_avg`` / ``exp_avg_sq`` in ``model.parameters()`` order — so ``optimizer_state_dict`` entries of reference
    checkpoints (utils/ckpt_util.py:86-104) load."""

    def __init__(self, flat, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0):
        self.flat = flat
        self.exp_avg = torch.zeros_like(flat.param)
        self.exp_avg_sq = torch.zeros_like(flat.param)
        self.step_count = 0
        super(FlatAdam, self).__init__(list(flat.params), dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))
        if len(self.param_groups) != 1:
            raise ValueError("FlatAdam keeps one param group (the whole flat buffer)")

    # hyper-parameters live in the param group (that is what LR schedulers edit)
    lr = property(lambda self: self.param_groups[0]["lr"])
    betas = property(lambda self: self.param_groups[0]["betas"])
    eps = property(lambda self: self.param_groups[0]["eps"])
    weight_decay = property(lambda self: self.param_groups[0]["weight_decay"])

    def set_lr(self, lr):
        self.param_groups[0]["lr"] = float(lr)

    def zero_grad(self, set_to_none=False):
        # every gradient is overwritten (not accumulated) by the next backward; nothing to clear
        self.flat.grads_ready = False

    @torch.no_grad()
    def step(self, closure=None, grad_scale=1.0):
        loss = closure() if closure is not None else None
        g = self.param_groups[0]
        self.step_count += 1
        ops.adam_step(self.flat.param, self.flat.grad, self.exp_avg, self.exp_avg_sq, g["lr"], g["betas"][0],
                      g["betas"][1], g["eps"], g["weight_decay"], self.step_count, grad_scale)
        return loss

    def _slices(self):
        off = 0
        for i, p in enumerate(self.flat.params):
            yield i, p, off, off + p.numel()
            off += p.numel()

    def state_dict(self):
        """torch.optim.Adam layout: {'state': {i: {'step', 'exp_avg', 'exp_avg_sq'}}, 'param_groups': [...]}."""
        state = {}
        if self.step_count > 0:
            for i, p, lo, hi in self._slices():
                state[i] = {"step": torch.tensor(float(self.step_count)),
                            "exp_avg": self.exp_avg[lo:hi].view(p.shape).clone(),
                            "exp_avg_sq": self.exp_avg_sq[lo:hi].view(p.shape).clone()}
        group = {k: v for k, v in self.param_groups[0].items() if k != "params"}
        group["params"] = list(range(len(self.flat.params)))
        return {"state": state, "param_groups": [group]}

    def load_state_dict(self, sd):
        if "param_groups" not in sd:          # round-1 format: flat tensors
            self.step_count = int(sd["step"])
            self.exp_avg.copy_(sd["exp_avg"])
            self.exp_avg_sq.copy_(sd["exp_avg_sq"])
            for k in ("lr", "betas", "eps", "weight_decay"):
                if k in sd:
                    self.param_groups[0][k] = sd[k]
            return
        groups = sd["param_groups"]
        if len(groups) != 1 or len(groups[0]["params"]) != len(self.flat.params):
            raise ValueError("optimizer state_dict does not match the model: expected one param group with %d "
                             "parameters" % len(self.flat.params))
        for k, v in groups[0].items():
            if k != "params":
                self.param_groups[0][k] = v
        ids = groups[0]["params"]
        state = sd["state"]
        steps = set()
        self.exp_avg.zero_()
        self.exp_avg_sq.zero_()
        for (i, p, lo, hi), pid in zip(self._slices(), ids):
            st = state.get(pid, state.get(str(pid)))
            if st is None:
                continue
            if tuple(st["exp_avg"].shape) != tuple(p.shape):
                raise ValueError("optimizer state of parameter %d has shape %s, expected %s"
                                 % (i, tuple(st["exp_avg"].shape), tuple(p.shape)))
            self.exp_avg[lo:hi].copy_(st["exp_avg"].reshape(-1))
            self.exp_avg_sq[lo:hi].copy_(st["exp_avg_sq"].reshape(-1))
            steps.add(int(float(st["step"])))
        if len(steps) > 1:
            raise ValueError("per-parameter step counts differ (%s): not an Adam state FlatAdam can represent" % steps)
        self.step_count = steps.pop() if steps else 0


def _load_checkpoint_file(path, trust_pickle):
    """torch.load of a reference-format checkpoint WITHOUT arbitrary unpickling: weights_only=True with the handful of
    numpy reconstructors a reference checkpoint needs allow-listed (`best_value` is a numpy.float64, train.py:311,508).
    Only `trust_pickle=True` (or YOLAT_TRUST_PICKLE=1 — checkpoints are third-party downloads) falls back to the full
    unpickler."""
    import numpy as np
    allow = [np.dtype, np.float64, np.float32, np.int64, np.ndarray]
    core = getattr(np, "_core", None) if hasattr(np, "_core") else getattr(np, "core", None)
    for name in ("scalar", "_reconstruct"):
        fn = getattr(getattr(core, "multiarray", None), name, None) if core is not None else None
        if fn is None:
            continue
        allow.append(fn)
        # the pickle names the module path of the numpy that SAVED the file: numpy 1.x (the reference's vintage,
        # deepgcn_env_install.sh) writes numpy.core.multiarray.*, numpy 2.x numpy._core.multiarray.* — allow both spellings
        # for the one function object (the (callable, "qualified name") form of safe_globals)
        for legacy in ("numpy.core.multiarray.", "numpy._core.multiarray."):
            allow.append((fn, legacy + name))
    for tname in ("Float64DType", "Float32DType", "Int64DType"):
        t = getattr(getattr(np, "dtypes", None), tname, None)
        if t is not None:
            allow.append(t)
    # torch releases before the (callable, "qualified name") form of safe_globals reject the tuples when the context is
    # entered: probe once and keep the plain callables as the baseline there, instead of reporting a TypeError of the allow
    # list as "the checkpoint does not load"
    try:
        with torch.serialization.safe_globals(allow):
            pass
    except (TypeError, ValueError, AttributeError):
        allow = [a for a in allow if not isinstance(a, tuple)]
    try:
        with torch.serialization.safe_globals(allow):
            return torch.load(path, map_location="cpu", weights_only=True)
    except Exception as exc:
        if trust_pickle or os.environ.get("YOLAT_TRUST_PICKLE", "0") == "1":
            return torch.load(path, map_location="cpu", weights_only=False)
        raise RuntimeError("checkpoint %r does not load with the safe unpickler (%s: %s); pass trust_pickle=True only for "
                           "files you trust — full unpickling executes code from the file" %
                           (path, type(exc).__name__, exc)) from exc


def load_reference_checkpoint(model, checkpoint, optimizer=None, strict=True, trust_pickle=False):
    """Load a checkpoint in the reference's format (cad_recognition/train.py:313-321: {'epoch', 'state_dict',
    'optimizer_state_dict', 'scheduler_state_dict', 'best_value'}) into the HIP modules: the same `module.` prefix
    fix-up as utils/ckpt_util.py:51-64 (checkpoints saved from a multi-GPU wrapper), then load_state_dict — the
    parameter / buffer names are the reference's.  `checkpoint`: a path or the loaded dict.
    Returns (epoch, best_value)."""
    if not isinstance(checkpoint, dict):
        # reference checkpoints (train.py:313-321) carry `best_value` as a numpy.float64 (train.py:311,508) next to
        # optimizer / scheduler dicts: torch's safe unpickler with numpy's scalar reconstructors allow-listed
        checkpoint = _load_checkpoint_file(checkpoint, trust_pickle)
    sd = checkpoint["state_dict"] if "state_dict" in checkpoint else checkpoint
    own_multi = next(iter(model.state_dict())).startswith("module.")
    ckpt_multi = next(iter(sd)).startswith("module.")
    if own_multi != ckpt_multi:
        sd = {(k[7:] if ckpt_multi else "module." + k): v for k, v in sd.items()}
    flat = getattr(model, "_yolat_flat", None)
    if flat is not None:
        # parameters are views of the flat buffer: copy in place so the views stay views
        own = model.state_dict()
        missing = [k for k in own if k not in sd]
        unexpected = [k for k in sd if k not in own]
        if strict and (missing or unexpected):
            raise RuntimeError("checkpoint does not match the model: missing %s, unexpected %s" % (missing, unexpected))
        with torch.no_grad():
            for k, v in sd.items():
                if k in own:
                    own[k].copy_(v)
        ops.bump_weight_epoch()
    else:
        model.load_state_dict(sd, strict=strict)
    if optimizer is not None and "optimizer_state_dict" in checkpoint:
        optimizer.load_state_dict(checkpoint["optimizer_state_dict"])
    return checkpoint.get("epoch", -1), checkpoint.get("best_value", None)


def dp_buckets():
    """YOLAT_DP_BUCKETS = 2 (default): the head bucket (fusion blocks + classifier, 93 % of the gradient bytes) is
    all-reduced from inside the backward, under the conv layers' backward, the conv bucket after it; 1: ONE all-reduce of the
    whole flat gradient after the backward (fewer, larger collectives: xGMI rings are per-link bound) — read per step so
    that a driver can quote both."""
    return 1 if os.environ.get("YOLAT_DP_BUCKETS", "2") == "1" else 2


def shard_graph_ids(num_graphs, rank, world_size):
    """Round-robin partition of dataset items (graph ids) over ranks — DistributedSampler-style.
    Every edge joins two nodes of the same image (Datasets/graph_dict3.py:594-600), so a shard never
    needs data from another rank."""
    return list(range(rank, num_graphs, world_size))


def broadcast_parameters(flat, model, src=0):
    """Make every replica start from rank ``src``'s parameters and BatchNorm buffers."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return
    dist.broadcast(flat.param, src)
    for b in model.buffers():
        dist.broadcast(b, src)


def allreduce_mean_(flat_grad):
    """One collective per step: SUM all-reduce of the flat gradient; the 1/world factor is folded
    into the Adam kernel's ``grad_scale``.  Returns that factor."""
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(flat_grad, op=dist.ReduceOp.SUM)
        return 1.0 / dist.get_world_size()
    return 1.0


# The training step as ONE native call per phase (csrc/train_plan.hip, yolat_train_step): the schedule of engine.model_fwd /
# model_bwd enqueued from C.  Module flag; False (or a model / batch outside the plan's shapes): the Python schedule.
TRAIN_PLAN = True


class TrainPlan(object):
    """Descriptor (pointers into the flat parameter / gradient buffers + BatchNorm buffers), grow-only workspace and status
    word of yolat_train_step for one Trainer.  `step_phases` enqueues phases of one training step on the current stream
    (+ the side stream engine.py uses); results are bit-identical to the Python schedule (tests/test_gpu_train_plan.py)."""

    def __init__(self, trainer):
        self.trainer = trainer
        self.model = trainer.model
        self._key = None
        self._desc = None
        self._ws = None
        self._status = None
        self._need = {}
        self.last = None           # (logits, loss) of the most recent step

    # -- eligibility --------------------------------------------------------------------------------------------------
    @staticmethod
    def schedule_is_default():
        """the plan restates engine.py's DEFAULT schedule: any flipped module flag (the tests flip them to cross-check
        schedules against each other) hands the step back to Python"""
        from . import engine
        return (engine.FACTORISED_TRAIN and engine.FUSED_FUSION_TRAIN and engine.FUSED_BN_CSR_BWD and
                engine.FUSED_BN_APPLY_SUMS and engine.FACT_FWD_RATIO == 2.0 and engine.FACT_BWD_RATIO == 1.0 and
                not engine._FAULT_EARLY_HEAD_EXCHANGE and not ops.X6_TRAIN_ROWS and
                ops.X6_TRAIN_GEMM == (os.environ.get("YOLAT_STRICT_FP32", "0") != "1"))

    def model_fits(self):
        from . import engine
        m = self.model
        net = getattr(m, "cls_net", None)
        if net is None or m.classifier != "softmax":
            return False
        try:
            convs = engine.model_convs(net)
            m1, m2, m3 = m.prediction_cls[0], m.prediction_cls[1], m.prediction_cls[2]
            if engine._drop_p(m2) > 0 or len(list(m3.children())) != 1:
                return False
            for cv in convs:
                if cv.nn[0].bias is None or cv.nn[3].bias is None or cv.lin_r.bias is None or cv.mlp_node[0].bias is None:
                    return False
                if not isinstance(cv.nn[1], torch.nn.BatchNorm1d) or not isinstance(cv.nn[4], torch.nn.BatchNorm1d):
                    return False
            for blk in (net.fusion_block, net.fusion_block_super, m1, m2):
                if blk[0].bias is None or not isinstance(blk[1], torch.nn.BatchNorm1d):
                    return False
            return m3[0].bias is not None and net.n_blocks_out == 2 and convs[0].nn[0].out_features == 64
        except (AttributeError, IndexError, TypeError):
            return False

    # -- descriptor ---------------------------------------------------------------------------------------------------
    def _tensors(self):
        return [self.trainer.flat.param, self.trainer.flat.grad] + list(self.model.buffers())

    def _build(self):
        from . import engine
        from ._lib import TrainModel
        m, flat = self.model, self.trainer.flat
        net = m.cls_net
        convs = engine.model_convs(net)
        d = TrainModel()
        d.n_blocks, d.n_blocks_out, d.n_classes = net.n_blocks, net.n_blocks_out, m.n_classes
        d.half = 1 if m.__dict__.get("_yolat_train_precision", "fp32") == "bf16" else 0
        d.C = convs[0].nn[0].out_features
        d.F = net.fusion_block[0].out_features
        d.H1, d.H2 = m.prediction_cls[0][0].out_features, m.prediction_cls[1][0].out_features
        base, end = flat.param.data_ptr(), flat.param.data_ptr() + 4 * flat.numel

        def lin(dst, mod):
            for t in (mod.weight, mod.bias):
                if not (base <= t.data_ptr() < end) or not t.is_contiguous():
                    raise ValueError("parameter outside the flat buffer")
            dst.W, dst.b = mod.weight.data_ptr(), mod.bias.data_ptr()

        def bn(dst, mod):
            for t in (mod.weight, mod.bias):
                if not (base <= t.data_ptr() < end):
                    raise ValueError("parameter outside the flat buffer")
            dst.gamma, dst.beta = mod.weight.data_ptr(), mod.bias.data_ptr()
            track = mod.track_running_stats and mod.running_mean is not None
            dst.running_mean = mod.running_mean.data_ptr() if track else None
            dst.running_var = mod.running_var.data_ptr() if track else None
            dst.num_batches_tracked = mod.num_batches_tracked.data_ptr() if mod.num_batches_tracked is not None else None
            dst.momentum = 0.1 if mod.momentum is None else float(mod.momentum)
            dst.eps = float(mod.eps)

        for l, cv in enumerate(convs):
            c = d.conv[l]
            c.Cin = cv.in_channels
            lin(c.nn0, cv.nn[0]); bn(c.bn1, cv.nn[1]); lin(c.nn3, cv.nn[3]); bn(c.bn4, cv.nn[4])
            lin(c.lin_r, cv.lin_r); lin(c.node, cv.mlp_node[0]); bn(c.bn_node, cv.mlp_node[1])
        lin(d.fus, net.fusion_block[0]); bn(d.fus_bn, net.fusion_block[1])
        lin(d.fus_s, net.fusion_block_super[0]); bn(d.fus_s_bn, net.fusion_block_super[1])
        m1, m2, m3 = m.prediction_cls[0], m.prediction_cls[1], m.prediction_cls[2]
        lin(d.c1, m1[0]); bn(d.c1_bn, m1[1]); lin(d.c2, m2[0]); bn(d.c2_bn, m2[1]); lin(d.c3, m3[0])
        d.param_base, d.grad_base = flat.param.data_ptr(), flat.grad.data_ptr()
        self._desc = d
        self._need.clear()
        if self._status is None:
            self._status = torch.zeros(1, dtype=torch.int32, device=flat.param.device)

    def prepare(self):
        """(Re)builds the descriptor when a tensor it points at has moved; returns False when the model is outside the plan."""
        key = tuple(t.data_ptr() for t in self._tensors()) + (self.model.__dict__.get("_yolat_train_precision", "fp32"),)
        if key != self._key:
            if not self.model_fits():
                self._desc = None
            else:
                try:
                    self._build()
                except ValueError:
                    self._desc = None
            self._key = key
        return self._desc is not None

    # -- one step -----------------------------------------------------------------------------------------------------
    def stage(self, data):
        """device operands of the step: (x, edge-or-None, strides, e_attr, bbox_idx, prepared graph-or-None, labels, N, E, P)"""
        model = self.model
        pre = data.__dict__.get("_yolat_graph") if hasattr(data, "__dict__") else None
        if pre is not None:
            x = data.x if data.x.dtype == torch.float32 else data.x.float()
            P = pre.P
            ops_in = (x, None, 0, 0, None, None, pre, pre.N, pre.E, P)
        else:
            st = model._stage_tensors(data)
            x, edge = st["x"], st["edge"]
            if edge.dim() != 2 or (edge.shape[1] != 2 and edge.shape[0] != 2):
                raise ValueError("edge must be [E,2] or [2,E]")
            if edge.shape[1] == 2 and not (edge.shape[0] == 2 and edge.stride(0) == 1):
                E, se, sc = edge.shape[0], edge.stride(0), edge.stride(1)
            else:
                E, se, sc = edge.shape[1], edge.stride(1), edge.stride(0)
            e_attr = st["e_attr"] if st["e_attr"].is_contiguous() else st["e_attr"].contiguous()
            ops_in = (x, edge, se, sc, e_attr, st["bbox_idx"], None, x.shape[0], E, st["bbox"].shape[0])
        labels = data.labels
        if not labels.is_cuda and labels.numel():
            lo, hi = int(labels.min()), int(labels.max())
            if lo < 0 or hi >= model.n_classes:
                raise IndexError("Target %d is out of bounds." % (lo if lo < 0 else hi))
        return ops_in + (labels.cuda(non_blocking=True),)

    def run(self, staged, phases, adam=None):
        """Enqueue `phases` (bit mask, yolat_train_step) of the step on `staged`; returns False when the C side declines the
        shapes (nothing was enqueued; only possible with phase 1)."""
        import ctypes
        from . import engine
        from ._lib import lib, check, GraphCsr, AdamArgs
        x, edge, se, sc, e_attr, bbox_idx, g, N, E, P, labels = staged
        d = self._desc
        if E < N or E <= 0:
            return False
        nk = (N, E, P)
        need = self._need.get(nk)
        if need is None:
            need = int(lib.yolat_train_step_workspace_bytes(ctypes.byref(d), N, E, P))
            if need == 0:
                return False
            if len(self._need) > 64:
                self._need.clear()
            self._need[nk] = need
        if self._ws is None or self._ws.numel() < need + 256:
            if phases & 1 == 0:
                raise RuntimeError("TrainPlan: the workspace of phase 1 is gone")
            self._ws = torch.empty(int(need * 1.1) + 4096, dtype=torch.uint8, device=x.device)
        ws_ptr = (self._ws.data_ptr() + 255) // 256 * 256
        ws_bytes = self._ws.numel() - (ws_ptr - self._ws.data_ptr())
        if phases & 1:
            K = self.model.n_classes
            self.last = (torch.empty(P, K, dtype=torch.float32, device=x.device),
                         torch.empty(1, dtype=torch.float32, device=x.device))
        logits, loss = self.last
        cur = ops.current_stream_object()
        side = engine.side_stream_for(cur) if engine.SIDE_STREAM else None
        gc = GraphCsr(*g.device_pointers()) if g is not None else None
        aa = None
        if adam is not None:
            aa = AdamArgs(*adam)
        rc = lib.yolat_train_step(ctypes.byref(d), ops._f(x, "x"), ops._ld(x),
                                  ops._i(edge, torch.int64, "edge") if edge is not None else None, se, sc,
                                  ops._f(e_attr, "e_attr") if e_attr is not None else None,
                                  ops._i(bbox_idx, torch.int64, "bbox_idx") if bbox_idx is not None else None,
                                  ctypes.byref(gc) if gc is not None else None, ops._i(labels, torch.int64, "labels"), N, E, P,
                                  logits.data_ptr(), logits.stride(0), loss.data_ptr(), ws_ptr, ws_bytes,
                                  self._status.data_ptr() if g is None else g.status.data_ptr(),
                                  ctypes.byref(aa) if aa is not None else None, int(phases), cur.cuda_stream,
                                  side.cuda_stream if side is not None else None)
        if rc == -2 and phases & 1:
            return False
        check(rc, "yolat_train_step")
        ops.bump_weight_epoch()       # BatchNorm running statistics (phase 1) / parameters (phase 4) changed behind torch's back
        return True

    def check_status(self):
        g = ops.Graph()
        g.status = self._status
        return ops.Graph.check_status(g)


class Trainer(object):
    def __init__(self, model, opt, lr=2.5e-4, weight_decay=1e-5, precision=None, check_inputs_every=0,
                 force_exchange=None, exchange_premul=None):
        self.model = model
        if precision is not None:
            model.set_train_precision(precision)
        self.flat = FlatParams(model)
        broadcast_parameters(self.flat, model)
        self.optimizer = FlatAdam(self.flat, lr=lr, weight_decay=weight_decay)
        self.criterion = DetectionLoss(opt)
        self._steps = 0
        # the input-validity word of the forward (edge ids inside [0, N), bbox_idx sorted) is read back — one
        # synchronisation — on the first step and then every `check_inputs_every` steps (0: first step only),
        # always BEFORE Adam applies the update, so a malformed batch never reaches the weights
        self.check_inputs_every = int(check_inputs_every)
        self.exchange_gradients = True      # False: skip the all-reduce (bench.py: cost of the exchange after overlap)
        # run the exchange branch (async bucket from inside backward, second bucket, wait) in a process group of ONE
        # rank as well: exercises the RCCL stream ordering on a single GPU (tests/test_gpu_dist.py, bench.py)
        self.force_exchange = bool(force_exchange)
        # exchange_premul = c: every gradient bucket is multiplied by c on the compute stream right before its
        # all-reduce and Adam's grad_scale carries 1 / (world c) — with c a power of two the step is bit-identical to the
        # plain one, but a gradient that lands in the bucket AFTER the exchange was issued (a missing join with the
        # side stream of the backward) stays unscaled and changes the result.  It makes the ordering of the exchange
        # observable in a process group of ONE rank, where SUM is the identity (tests/test_gpu_dist.py).
        self.exchange_premul = None if exchange_premul is None else float(exchange_premul)
        self.plan = TrainPlan(self)
        self.plan_steps = 0                 # steps that went through yolat_train_step (diagnostics, tests, bench.py)

    def _adam_args(self, scale):
        o = self.optimizer
        g = o.param_groups[0]
        return (o.exp_avg.data_ptr(), o.exp_avg_sq.data_ptr(), self.flat.numel, o.step_count, float(g["lr"]),
                float(g["betas"][0]), float(g["betas"][1]), float(g["eps"]), float(g["weight_decay"]), float(scale))

    def _plan_step(self, data):
        """The step through yolat_train_step; returns the loss tensor, or None when the plan declines (nothing enqueued)."""
        plan = self.plan
        if not (TRAIN_PLAN and TrainPlan.schedule_is_default() and plan.prepare()):
            return None
        staged = plan.stage(data)
        grouped = dist.is_available() and dist.is_initialized()
        world = dist.get_world_size() if grouped else 1
        exchange = self.exchange_gradients and grouped and (world > 1 or self.force_exchange)
        check = self._steps == 0 or (self.check_inputs_every > 0 and self._steps % self.check_inputs_every == 0)
        if not self.model.training:          # (nn.Module.train() walks ~100 modules: not once per step)
            self.model.train()
        self.optimizer.zero_grad()
        self.model.__dict__["_yolat_plan"] = plan if staged[6] is None else staged[6]
        if not exchange and not check:
            # the whole step in ONE call: graph + forward + loss + backward + Adam
            self.optimizer.step_count += 1
            if not plan.run(staged, 7, self._adam_args(1.0)):
                self.optimizer.step_count -= 1
                return None
            self.flat.grads_ready = True
            self._steps += 1
            self.plan_steps += 1
            return plan.last[1][0]
        if not plan.run(staged, 1):
            return None
        premul = self.exchange_premul if exchange else None
        scale = 1.0

        def reduce_(bucket, async_op):
            if premul is not None:
                bucket.mul_(premul)
            return dist.all_reduce(bucket, op=dist.ReduceOp.SUM, async_op=async_op)

        if exchange and self.flat.conv_end > 0 and dp_buckets() == 2:
            handles = [reduce_(self.flat.grad[self.flat.conv_end:], True)]      # the head bucket, final after phase 1
            plan.run(staged, 2)
            handles.append(reduce_(self.flat.grad[:self.flat.conv_end], True))
            for h in handles:
                h.wait()
            scale = 1.0 / world
        else:
            plan.run(staged, 2)
            if exchange:
                reduce_(self.flat.grad, False)
                scale = 1.0 / world
        if premul is not None:
            scale /= premul
        self.flat.grads_ready = True
        if check:
            self.model.check_last_status()      # raises before the update is applied
        self._steps += 1
        self.plan_steps += 1
        self.optimizer.step(grad_scale=scale)
        return plan.last[1][0]

    def step(self, data, slices=None):
        """One training step on this rank's batch.  Returns the (device) loss tensor."""
        from . import engine
        if engine._FAULT_EARLY_HEAD_EXCHANGE and self.exchange_premul is None:
            raise RuntimeError("engine._FAULT_EARLY_HEAD_EXCHANGE is set outside its test (a leaked fault-injection flag): "
                               "the head bucket would be exchanged before its gradients exist")
        loss = self._plan_step(data)
        if loss is not None:
            return loss.detach()
        self.model.train()
        self.optimizer.zero_grad()
        out = self.model(data, slices)
        loss = self.criterion(out, data)["loss"]
        grouped = dist.is_available() and dist.is_initialized()
        world = dist.get_world_size() if grouped else 1
        exchange = self.exchange_gradients and grouped and (world > 1 or self.force_exchange)
        if not exchange:
            world = 1
        premul = self.exchange_premul if exchange else None

        def reduce_(bucket, async_op):
            if premul is not None:
                bucket.mul_(premul)
            return dist.all_reduce(bucket, op=dist.ReduceOp.SUM, async_op=async_op)

        if exchange and self.flat.conv_end > 0 and dp_buckets() == 2:
            # bucket 1 (fusion blocks + classifier, 93 % of the bytes) is all-reduced while the conv layers'
            # backward still runs; bucket 2 (conv layers) after the backward
            handles = []
            tail = self.flat.grad[self.flat.conv_end:]
            self.flat.on_head_done = lambda: handles.append(reduce_(tail, True))
            try:
                loss.backward()
            finally:
                self.flat.on_head_done = None
            handles.append(reduce_(self.flat.grad[:self.flat.conv_end], True))
            for h in handles:
                h.wait()
            scale = 1.0 / world
        else:
            loss.backward()
            scale = 1.0
            if exchange:
                reduce_(self.flat.grad, False)
                scale = 1.0 / world
        if premul is not None:
            scale /= premul
        if self._steps == 0 or (self.check_inputs_every > 0 and self._steps % self.check_inputs_every == 0):
            self.model.check_last_status()      # raises before the update is applied
        self._steps += 1
        self.optimizer.step(grad_scale=scale)
        return loss.detach()
