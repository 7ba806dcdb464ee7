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

    def step(self, data, slices=None):
        """One training step on this rank's batch.  Returns the (device) loss tensor."""
        from . import engine
        if engine._FAULT_EARLY_HEAD_EXCHANGE and self.exchange_premul is None:
            raise RuntimeError("engine._FAULT_EARLY_HEAD_EXCHANGE is set outside its test (a leaked fault-injection flag): "
                               "the head bucket would be exchanged before its gradients exist")
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

        if exchange and self.flat.conv_end > 0:
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
