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


class FlatAdam(object):
    def __init__(self, flat, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0):
        self.flat = flat
        self.lr, self.betas, self.eps, self.weight_decay = lr, betas, eps, weight_decay
        self.exp_avg = torch.zeros_like(flat.param)
        self.exp_avg_sq = torch.zeros_like(flat.param)
        self.step_count = 0

    def zero_grad(self):
        # every gradient is overwritten (not accumulated) by the next backward; nothing to clear
        self.flat.grads_ready = False

    def step(self, grad_scale=1.0):
        self.step_count += 1
        ops.adam_step(self.flat.param, self.flat.grad, self.exp_avg, self.exp_avg_sq, self.lr,
                      self.betas[0], self.betas[1], self.eps, self.weight_decay, self.step_count,
                      grad_scale)

    def state_dict(self):
        return {"step": self.step_count, "exp_avg": self.exp_avg, "exp_avg_sq": self.exp_avg_sq,
                "lr": self.lr, "betas": self.betas, "eps": self.eps, "weight_decay": self.weight_decay}

    def load_state_dict(self, sd):
        self.step_count = int(sd["step"])
        self.exp_avg.copy_(sd["exp_avg"])
        self.exp_avg_sq.copy_(sd["exp_avg_sq"])


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
    def __init__(self, model, opt, lr=2.5e-4, weight_decay=1e-5):
        self.model = model
        self.flat = FlatParams(model)
        broadcast_parameters(self.flat, model)
        self.optimizer = FlatAdam(self.flat, lr=lr, weight_decay=weight_decay)
        self.criterion = DetectionLoss(opt)

    def step(self, data, slices=None):
        """One training step on this rank's batch.  Returns the (device) loss tensor."""
        self.model.train()
        self.optimizer.zero_grad()
        out = self.model(data, slices)
        loss = self.criterion(out, data)["loss"]
        world = dist.get_world_size() if (dist.is_available() and dist.is_initialized()) else 1
        if world > 1 and self.flat.conv_end > 0:
            # bucket 1 (fusion blocks + classifier, 93 % of the bytes) is all-reduced while the conv layers'
            # backward still runs; bucket 2 (conv layers) after the backward
            handles = []
            tail = self.flat.grad[self.flat.conv_end:]
            self.flat.on_head_done = lambda: handles.append(dist.all_reduce(tail, op=dist.ReduceOp.SUM, async_op=True))
            try:
                loss.backward()
            finally:
                self.flat.on_head_done = None
            handles.append(dist.all_reduce(self.flat.grad[:self.flat.conv_end], op=dist.ReduceOp.SUM, async_op=True))
            for h in handles:
                h.wait()
            scale = 1.0 / world
        else:
            loss.backward()
            scale = allreduce_mean_(self.flat.grad)
        self.optimizer.step(grad_scale=scale)
        return loss.detach()
