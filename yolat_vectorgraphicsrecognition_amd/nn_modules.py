"""MI355X-native mirror of the ``gcn_lib.sparse`` surface the YOLaT architecture uses.

Same class names, constructor signatures, sub-module attribute names (=> same state_dict keys,
SURVEY.md App. C) and error behaviour as the reference:

    MultiSeq, MLP            gcn_lib/sparse/torch_nn.py:37-71
    GraphConv, ResBlock      gcn_lib/sparse/torch_vertex.py:730-775, 808-829
    AttrRelativeEdgeConvGlobalPool2   gcn_lib/sparse/torch_vertex.py:288-341

Their ``forward`` runs the hand-written HIP kernels (engine.py / libyolat_hip.so); nothing here falls
back to torch arithmetic.  The ``nn.Linear`` / ``nn.BatchNorm1d`` / ``nn.ReLU`` children are parameter
containers only (their own ``forward`` is never called).
"""
import torch
from torch import nn

from . import engine, ops
from .engine import Lazy, GradSink


# ---------------------------------------------------------------------------------------------
# basic layers (torch_nn.py)
# ---------------------------------------------------------------------------------------------

def act_layer(act_type, inplace=False, neg_slope=0.2, n_prelu=1):
    """torch_nn.py:9-20.  Only 'relu' has a HIP implementation (the only one the arch uses)."""
    act = act_type.lower()
    if act == "relu":
        return nn.ReLU(inplace)
    if act in ("leakyrelu", "prelu"):
        raise NotImplementedError("activation layer [%s] has no MI355X kernel (arch uses relu)" % act)
    raise NotImplementedError("activation layer [%s] is not found" % act)


def norm_layer(norm_type, nc):
    """torch_nn.py:23-34.  Only 'batch' has a HIP implementation."""
    norm = norm_type.lower()
    if norm == "batch":
        return nn.BatchNorm1d(nc, affine=True)
    if norm in ("layer", "instance"):
        raise NotImplementedError("normalization layer [%s] has no MI355X kernel (arch uses batch)" % norm)
    raise NotImplementedError("normalization layer [%s] is not found" % norm)


class MultiSeq(nn.Sequential):
    """torch_nn.py:37-47: a Sequential that splats tuple outputs into the next module."""

    def __init__(self, *args):
        super(MultiSeq, self).__init__(*args)

    def forward(self, *inputs):
        for module in self._modules.values():
            if type(inputs) == tuple:
                inputs = module(*inputs)
            else:
                inputs = module(inputs)
        return inputs


def _groups(seq):
    """Split an MLP's children into (Linear, BatchNorm1d|None, relu, dropout_p) groups."""
    groups = []
    for m in seq.children():
        if isinstance(m, nn.Linear):
            groups.append([m, None, False, 0.0])
        elif isinstance(m, nn.BatchNorm1d):
            groups[-1][1] = m
        elif isinstance(m, nn.ReLU):
            groups[-1][2] = True
        elif isinstance(m, (nn.Dropout, nn.Dropout2d)):
            groups[-1][3] = float(m.p)
        else:
            raise NotImplementedError("layer %s has no MI355X kernel" % m.__class__.__name__)
    return groups


class _MLPFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, mlp, x, *params):
        training = mlp.training
        a = Lazy(x)
        saved = []
        for lin, bn, relu, drop in _groups(mlp):
            a, sv = engine.lbr_fwd(a, lin, bn, relu, training)
            if drop > 0 and training:
                a, sv["drop"] = engine.dropout_fwd(a, drop)
            saved.append(sv)
        engine.flush_batch_counters()
        ctx.mlp, ctx.saved_blocks, ctx.params = mlp, saved, params
        if not training:
            ctx.saved_blocks = None
        return a.materialise()

    @staticmethod
    def backward(ctx, dz):
        if ctx.saved_blocks is None:
            raise RuntimeError("backward through an eval-mode MLP is not supported by the HIP path")
        sink = GradSink()
        d = dz.contiguous().clone()
        blocks = ctx.saved_blocks
        for i in range(len(blocks) - 1, -1, -1):
            if blocks[i].get("drop") is not None:
                d = engine.dropout_bwd(blocks[i]["drop"], d)
            d = engine.lbr_bwd(blocks[i], d, sink, need_dx=(i > 0 or ctx.needs_input_grad[1]))
        return (None, d if ctx.needs_input_grad[1] else None) + tuple(sink.out.get(id(p)) for p in ctx.params)


class MLP(nn.Sequential):
    """torch_nn.py:50-71: [Linear, (norm), (act), (Dropout2d)] per layer."""

    def __init__(self, channels, act="relu", norm=None, bias=True, drop=0., last_lin=False):
        m = []
        for i in range(1, len(channels)):
            m.append(nn.Linear(channels[i - 1], channels[i], bias))
            if (i == len(channels) - 1) and last_lin:
                pass
            else:
                if norm is not None and norm.lower() != "none":
                    m.append(norm_layer(norm, channels[i]))
                if act is not None and act.lower() != "none":
                    m.append(act_layer(act))
                if drop > 0:
                    m.append(nn.Dropout2d(drop))
        self.m = m
        super(MLP, self).__init__(*self.m)

    def forward(self, x):
        return _MLPFn.apply(self, x, *list(self.parameters()))


# ---------------------------------------------------------------------------------------------
# graph convolution (torch_vertex.py)
# ---------------------------------------------------------------------------------------------

_GRAPH_CACHE = []          # [(key, Graph)] most recent first, tiny LRU


def graph_for(edge_index, edge_attr, num_nodes, bbox_idx=None, num_proposals=1):
    """CSR/CSC structure for an ``edge_index`` tensor, cached by tensor identity + version."""
    key = (edge_index.data_ptr(), edge_index._version, tuple(edge_index.shape), edge_index.stride(),
           edge_attr.data_ptr(), edge_attr._version, num_nodes,
           None if bbox_idx is None else (bbox_idx.data_ptr(), bbox_idx._version), num_proposals)
    for k, g in _GRAPH_CACHE:
        if k == key:
            return g
    g = ops.build_graph(edge_index, edge_attr, bbox_idx, num_nodes, num_proposals)
    _GRAPH_CACHE.insert(0, (key, g))
    del _GRAPH_CACHE[4:]
    return g


class _ConvFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, conv, g, x, x_node, *params):
        training = conv.training
        f, s, sv = engine.conv_fwd(conv, g, x, Lazy(x_node), None, None, training)
        engine.flush_batch_counters()
        ctx.conv, ctx.g, ctx.params = conv, g, params
        ctx.sv = sv if training else None
        return f, s.materialise()

    @staticmethod
    def backward(ctx, d_f, d_s):
        if ctx.sv is None:
            raise RuntimeError("backward through an eval-mode GraphConv is not supported by the HIP path")
        sink = GradSink()
        need_dx = ctx.needs_input_grad[2] or ctx.needs_input_grad[3]
        d_f = d_f.contiguous()
        d_s = d_s.contiguous().clone()
        dx, dxn = engine.conv_bwd(ctx.sv, ctx.g, d_f, d_s, sink, need_dx=need_dx)
        return (None, None, dx if ctx.needs_input_grad[2] else None,
                dxn if ctx.needs_input_grad[3] else None) + tuple(sink.out.get(id(p)) for p in ctx.params)


class AttrRelativeEdgeConvGlobalPool2(nn.Module):
    """torch_vertex.py:288-341:  out_i = mean_{e:(j->i)} nn([x_i, x_j-x_i, a_e]) + lin_r(x_i);
    x_node' = mlp_node(x_node).  aggr='mean', flow source_to_target (edge_index[0]=j, [1]=i)."""

    def __init__(self, in_channels, out_channels, **kwargs):
        super(AttrRelativeEdgeConvGlobalPool2, self).__init__()
        self.nn = MLP([in_channels * 2 + 4, out_channels, out_channels], "relu", "batch")
        self.lin_r = nn.Linear(in_channels, out_channels, bias=True)
        self.mlp_node = MLP([in_channels, out_channels], "relu", "batch")
        self.in_channels = in_channels
        self.aggr = "mean"

    def forward(self, x, x_node, edge_index, edge_weight=None, edge_attr=None):
        if edge_weight is not None:
            raise NotImplementedError("edge_weight (norm) is always None on the reference path "
                                      "(architecture3cc_rpn_gp_iter2.py:114)")
        g = edge_index if isinstance(edge_index, ops.Graph) else graph_for(edge_index, edge_attr, x.shape[0])
        return _ConvFn.apply(self, g, x, x_node, *list(self.parameters()))

    def __repr__(self):
        return "{}(nn={})".format(self.__class__.__name__, self.nn)


class GraphConv(nn.Module):
    """torch_vertex.py:730-775 — static graph convolution layer; string dispatch on ``conv``.
    Only 'attr_edge_gp2' (the one the only shipped architecture hard-codes,
    architecture3cc_rpn_gp_iter2.py:22) exists here; every other name raises like an unknown conv."""

    def __init__(self, in_channels, out_channels, conv="gcn", act="relu", norm=None, bias=True, heads=8):
        super(GraphConv, self).__init__()
        self.conv = conv.lower()
        if self.conv == "attr_edge_gp2":
            self.gconv = AttrRelativeEdgeConvGlobalPool2(in_channels, out_channels)
        else:
            raise NotImplementedError("conv {} is not implemented".format(conv))

    def forward(self, x, edge_index, edge_weight=None, edge_attr=None, pos=None, x_node=None):
        return self.gconv(x, x_node, edge_index, edge_weight, edge_attr)


class ResBlock(nn.Module):
    """torch_vertex.py:808-829 — for attr_edge_gp2 the residual adds are commented out (:825-826):
    returns (out, out_node) unchanged."""

    def __init__(self, channels, conv="edge", act="relu", norm=None, bias=True, res_scale=1, **kwargs):
        super(ResBlock, self).__init__()
        self.body = GraphConv(channels, channels, conv, act, norm, bias, **kwargs)
        self.res_scale = res_scale
        self.channels = channels

    def forward(self, x, edge, edge_weight=None, edge_attr=None, pos=None, x_node=None):
        out, out_node = self.body(x, edge, edge_weight, edge_attr, x_node=x_node)
        return out, out_node


class _Unavailable(nn.Module):
    """Names imported by the architecture file but never instantiated by it
    (architecture3cc_rpn_gp_iter2.py:7): PlainDynBlock, DenseDynBlock, DilatedKnnGraph."""

    def __init__(self, *a, **k):
        raise NotImplementedError("%s is not used by architecture3cc_rpn_gp_iter2 and is out of scope"
                                  % self.__class__.__name__)


class PlainDynBlock(_Unavailable):
    pass


class DenseDynBlock(_Unavailable):
    pass


class DilatedKnnGraph(_Unavailable):
    pass


# ---------------------------------------------------------------------------------------------
# torch_scatter.scatter stand-in for sorted indices (architecture3cc_rpn_gp_iter2.py:67,122)
# ---------------------------------------------------------------------------------------------

class _SegFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, src, g, reduce):
        P, D = g.P, src.shape[1]
        out = torch.empty(P, D, dtype=torch.float32, device=src.device)
        ctx.g, ctx.reduce, ctx.shape = g, reduce, src.shape
        if reduce == "mean":
            ops.segment_mean_fwd(src, g, out)
        else:
            arg = torch.empty(P, D, dtype=torch.int32, device=src.device)
            ops.segment_max_fwd(src, g, out, arg)
            ctx.arg = arg
        return out

    @staticmethod
    def backward(ctx, dout):
        dout = dout.contiguous()
        dx = torch.empty(ctx.shape, dtype=torch.float32, device=dout.device)
        if ctx.reduce == "mean":
            ops.segment_mean_bwd(dout, ctx.g, dx)
        else:
            ops.segment_max_bwd(dout, ctx.arg, ctx.g, dx)
        return dx, None, None


def scatter(src, index, dim=0, dim_size=None, reduce="sum"):
    """``torch_scatter.scatter`` for the two calls on the path: dim 0, reduce 'mean' | 'max', and a
    NON-DECREASING index (bbox_idx, Datasets/graph_dict3.py:732).  Output rows = dim_size or
    index.max()+1 (= index[-1]+1 for a sorted index).  Anything else raises."""
    if dim != 0 or reduce not in ("mean", "max"):
        raise NotImplementedError("scatter(dim=%r, reduce=%r) is not on the YOLaT hot path" % (dim, reduce))
    if dim_size is None:
        dim_size = int(index[-1].item()) + 1
    N = src.shape[0]
    dev = src.device
    g = ops.Graph()
    g.N, g.E, g.P = N, 0, int(dim_size)
    g.col_ptr = g.slots = None
    g.status = torch.zeros(1, dtype=torch.int32, device=dev)
    g.seg_ptr = torch.empty(g.P + 1, dtype=torch.int32, device=dev)
    g.node_seg = torch.empty(max(N, 1), dtype=torch.int32, device=dev)
    from ._lib import lib, check
    check(lib.yolat_segment_ptr(index.data_ptr(), N, g.P, g.seg_ptr.data_ptr(), g.node_seg.data_ptr(),
                                g.status.data_ptr(), torch.cuda.current_stream().cuda_stream),
          "yolat_segment_ptr")
    g.check_status()
    return _SegFn.apply(src.contiguous(), g, reduce)
