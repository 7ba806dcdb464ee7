"""GPU tests (-m gpu) of the one-launch proposal-local conv stack of the bf16-storage eval forward
(csrc/conv_local.hip: yolat_conv_local_pack, yolat_conv_stack_local_bf16, and its use inside yolat_forward_eval_bf16).

Reference path: Backbone.forward, cad_recognition/architecture3cc_rpn_gp_iter2.py:44-69 (conv layers, concat, segment
mean of the node branch) + the segment max of :122 over the concat columns; AttrRelativeEdgeConvGlobalPool2,
gcn_lib/sparse/torch_vertex.py:319-337.  Oracle: oracle/oracle_torch.py (fp32 / fp64 on the CPU).  Tolerances are the
bf16-storage ones of tests/test_gpu_bf16.py (<= 1e-2 of the scale, rms)."""
import ctypes
import os

import numpy as np
import pytest
import torch

import golden_util as gu
from oracle import oracle_torch as orc

pytestmark = pytest.mark.gpu

RTOL_BF16 = 2e-2
RMS_BF16 = 1e-2


def _yv():
    import yolat_vectorgraphicsrecognition_amd as yv
    return yv


def _model(yv, optkw, seed):
    return gu.fill_state_(yv.SparseCADGCN(yv.Opt(**optkw)), seed).cuda().eval().set_eval_precision("bf16")


class _mode(object):
    """YOLAT_CONV_LOCAL for the duration of a block (read per call by yolat_forward_eval_bf16)."""

    def __init__(self, v):
        self.v = str(v)

    def __enter__(self):
        self.old = os.environ.get("YOLAT_CONV_LOCAL")
        os.environ["YOLAT_CONV_LOCAL"] = self.v

    def __exit__(self, *a):
        if self.old is None:
            os.environ.pop("YOLAT_CONV_LOCAL", None)
        else:
            os.environ["YOLAT_CONV_LOCAL"] = self.old


def _rel(got, want):
    got, want = got.double().cpu(), want.double().cpu()
    scale = float(want.abs().max())
    return (float((got - want).abs().max()) / scale,
            float((got - want).pow(2).mean().sqrt() / want.pow(2).mean().sqrt().clamp(min=1e-30)))


@pytest.fixture(params=[4, 8], autouse=True)
def _waves(request):
    """every test on both tile shapes of the kernel: 4-wave workgroups / 64-node tiles and 8-wave / 128-node tiles"""
    from yolat_vectorgraphicsrecognition_amd._lib import lib
    lib.yolat_conv_local_tune(request.param, 0, 0, None)
    yield request.param
    lib.yolat_conv_local_tune(0, 0, 0, None)


def _ragged(yv, P, seed, lo=2, hi=40, edge_factor=2.1, classes=17, edges_per_proposal=None):
    d = yv.synth_graph(num_proposals=P, nodes_lo=lo, nodes_hi=hi, edge_factor=edge_factor, n_classes=classes, seed=seed,
                       edges_per_proposal=edges_per_proposal)
    return d


def _run_stack(yv, model, d):
    """yolat_conv_stack_local_bf16 on the model's packed weights: (feats [N,D] bf16, Z [P, 2(F+D)] fp32, flag)."""
    from yolat_vectorgraphicsrecognition_amd import ops
    from yolat_vectorgraphicsrecognition_amd._lib import lib, check, GraphCsr
    with torch.no_grad():
        model(d, None)                      # builds the plan (weights folded / packed)
    plan = model._yolat_plan
    h = plan._desc_h
    assert h is not None and h.conv_local, "the plan did not pack the conv stack"
    base = plan._desc
    N, P = d.x.shape[0], d.bbox.shape[0]
    dev = torch.device("cuda")
    g = ops.build_graph(d.edge.to(dev), d.e_attr.to(dev), d.bbox_idx.to(dev), N, P)
    g.check_status()
    D, F = base.C * base.n_blocks_out, base.F
    feats = torch.full((N, D), float("nan"), dtype=torch.bfloat16, device=dev)
    Z = torch.full((P, 2 * (F + D)), float("nan"), dtype=torch.float32, device=dev)
    flag = torch.zeros(1, dtype=torch.int32, device=dev)
    x = d.x.to(dev).contiguous()
    gc = GraphCsr(*g.device_pointers())
    check(lib.yolat_conv_stack_local_bf16(ctypes.byref(h), h.conv_local, x.data_ptr(), x.stride(0), ctypes.byref(gc), N, g.E,
                                          P, feats.data_ptr(), D, Z.data_ptr(), Z.stride(0), flag.data_ptr(),
                                          torch.cuda.current_stream().cuda_stream), "yolat_conv_stack_local_bf16")
    torch.cuda.synchronize()
    return feats, Z, int(flag.item()), (F, D)


def _oracle_backbone(optkw, seed, d):
    ref = gu.fill_state_(orc.SparseCADGCN(orc.Opt(**optkw)), seed).eval()
    with torch.no_grad():
        x = d.x
        edge_index = d.edge.t()
        out_feat, out_sup = ref.cls_net(x, [edge_index], [None], [d.e_attr], d.bbox_idx)
    return ref, out_feat, out_sup


@pytest.mark.parametrize("blocks,blocks_out,cin,P,seed,shape", [
    (2, 2, 5, 300, 1, dict(lo=2, hi=40, edge_factor=2.1)),                 # ragged, nodes without in-edges
    (4, 2, 5, 1500, 2, dict(lo=25, hi=25, edges_per_proposal=150)),       # cfg-5-like: 6 in-edges per node
    (4, 4, 5, 257, 3, dict(lo=3, hi=24, edge_factor=1.2)),                 # ~1.2 edges per node (Diagrams-like)
    (3, 1, 3, 64, 4, dict(lo=30, hi=60, edges_per_proposal=500)),         # one proposal per tile, in-degree ~11
    (2, 2, 8, 5, 5, dict(lo=5, hi=9, edge_factor=2.0)),                    # a handful of proposals, in_channels 8
])
def test_conv_stack_local_matches_cpu_oracle(blocks, blocks_out, cin, P, seed, shape):
    """feats (the concat of the last n_blocks_out layer outputs), their per-proposal max, the per-proposal mean of the node
    branch and the zeroed fusion columns against the fp32 CPU oracle's Backbone; flag stays 0; bit-identical run to run."""
    yv = _yv()
    optkw = dict(n_classes=17, n_blocks=blocks, n_blocks_out=blocks_out, in_channels=cin)
    d = _ragged(yv, P, 100 + seed, **shape)
    if cin != 5:
        d.x = torch.randn(d.x.shape[0], cin, generator=torch.Generator().manual_seed(cin)) * 0.7
    model = _model(yv, optkw, 40 + seed)
    feats, Z, flag, (F, D) = _run_stack(yv, model, d)
    feats2, Z2, flag2, _ = _run_stack(yv, model, d)
    assert flag == 0 and flag2 == 0
    assert torch.equal(feats.view(torch.int16), feats2.view(torch.int16))
    written = torch.cat([Z[:, :F + D], Z[:, 2 * F + D:]], 1)
    assert torch.isfinite(written).all() and torch.equal(written, torch.cat([Z2[:, :F + D], Z2[:, 2 * F + D:]], 1))
    _, out_feat, out_sup = _oracle_backbone(optkw, 40 + seed, d)
    want_feats = out_feat[:, F:]
    depth = max(1.0, blocks / 4.0)
    for name, got, want in (("feats", feats.float(), want_feats),
                            ("mean(node branch)", Z[:, 2 * F + D:], out_sup[:, F:])):
        assert torch.isfinite(got).all(), name
        mx, rms = _rel(got, want)
        assert mx <= RTOL_BF16 * depth and rms <= RMS_BF16 * depth, "%s: max %.2e rms %.2e" % (name, mx, rms)
    # the pooled maximum is exactly the maximum of the feats rows the kernel stored
    bb = d.bbox_idx.cuda()
    want_max = torch.full((P, D), -float("inf"), device="cuda").scatter_reduce(
        0, bb[:, None].expand(-1, D), feats.float(), "amax", include_self=True)
    assert torch.equal(Z[:, F:F + D], want_max)
    assert bool((Z[:, :F] == 0).all())


def _with_big_proposal(d, p_mid, n_nodes):
    """merge the proposals around p_mid into ONE proposal of n_nodes nodes (ids renumbered: consecutive and sorted)"""
    bb = d.bbox_idx.clone()
    n_lo = int((bb < p_mid).sum())
    bb[n_lo:n_lo + n_nodes] = p_mid
    _, inv = torch.unique_consecutive(bb, return_inverse=True)
    d.bbox_idx = inv
    P2 = int(inv.max()) + 1
    d.bbox = d.bbox[:P2].clone()
    d.stat_feats = d.stat_feats[:P2].clone()
    d.labels = d.labels[:P2].clone()
    return d


def test_conv_stack_local_raises_the_flag_for_batches_without_the_property(_waves):
    yv = _yv()
    optkw = dict(n_classes=17, n_blocks=2, n_blocks_out=2)
    model = _model(yv, optkw, 3)
    T = 16 * _waves
    # (a) a proposal of T + 36 nodes does not fit a T-node tile (and one of exactly T nodes does)
    assert _run_stack(yv, model, _with_big_proposal(_ragged(yv, 60, 7, lo=3, hi=20), 20, T + 36))[2] == 1
    assert _run_stack(yv, model, _ragged(yv, 5, 8, lo=T + 1, hi=T + 1, edges_per_proposal=300))[2] == 1
    assert _run_stack(yv, model, _ragged(yv, 5, 8, lo=T, hi=T, edges_per_proposal=300))[2] == 0
    # (b) an edge that leaves its proposal (and its tile)
    d3 = _ragged(yv, 40, 9, lo=3, hi=20)
    d3.edge = d3.edge.clone()
    d3.edge[5, 0] = d3.x.shape[0] - 1
    assert _run_stack(yv, model, d3)[2] == 1
    # (c) more edges than a tile holds (8 T)
    assert _run_stack(yv, model, _ragged(yv, 3, 10, lo=40, hi=40, edges_per_proposal=8 * T + 100))[2] == 1
    # (d) and the ordinary batch keeps it down
    assert _run_stack(yv, model, _ragged(yv, 40, 11, lo=3, hi=20))[2] == 0


@pytest.mark.parametrize("blocks,blocks_out,P,seed,shape", [
    (2, 2, 211, 1, dict(lo=2, hi=40, edge_factor=2.1)),
    (4, 2, 1200, 2, dict(lo=25, hi=25, edges_per_proposal=150)),
    (4, 2, 1100, 3, dict(lo=4, hi=40, edge_factor=1.2)),
])
def test_forward_with_the_local_conv_stack_matches_oracle_and_per_layer_path(blocks, blocks_out, P, seed, shape):
    """the whole bf16 forward with the one-launch conv stack forced on (YOLAT_CONV_LOCAL=2) against the CPU oracle (same
    bounds as the per-layer path) and against the per-layer path (YOLAT_CONV_LOCAL=0): two bf16 evaluations of the same
    function, a fraction of the bf16 error budget apart; deterministic."""
    yv = _yv()
    optkw = dict(n_classes=17, n_blocks=blocks, n_blocks_out=blocks_out)
    d = _ragged(yv, P, 200 + seed, **shape)
    d.edge = torch.cat([d.edge, d.edge[:17]], 0)            # duplicate edges are defined by the PyG semantics
    d.e_attr = torch.cat([d.e_attr, d.e_attr[:17]], 0)
    model = _model(yv, optkw, 60 + seed)
    ref = gu.fill_state_(orc.SparseCADGCN(orc.Opt(**optkw)), 60 + seed).eval()
    with torch.no_grad():
        want = ref(d, None)[0]
        with _mode(2):
            got = model(d, None)[0].clone()
            again = model(d, None)[0].clone()
        with _mode(0):
            layers = model(d, None)[0].clone()
    model._yolat_plan.check_status()
    assert torch.equal(got, again)
    depth = max(1.0, blocks / 4.0)
    for name, t in (("local", got), ("per-layer", layers)):
        mx, rms = _rel(t, want)
        assert mx <= RTOL_BF16 * depth and rms <= RMS_BF16 * depth, "%s: max %.2e rms %.2e" % (name, mx, rms)
    mx, rms = _rel(got, layers)
    assert mx <= RTOL_BF16 * depth and rms <= RMS_BF16 * depth, "local vs per-layer: max %.2e rms %.2e" % (mx, rms)


def test_forward_falls_back_to_the_per_layer_launches_bit_exactly(_waves):
    """a batch without the property (a 100-node proposal; an edge between two proposals): the gated per-layer launches run
    and the logits are bit-identical to the per-layer path's."""
    yv = _yv()
    optkw = dict(n_classes=17, n_blocks=3, n_blocks_out=2)
    model = _model(yv, optkw, 5)
    base = _ragged(yv, 1500, 21, lo=3, hi=30)
    cases = []
    d = _ragged(yv, 1500, 21, lo=3, hi=30)
    d.edge = d.edge.clone()
    d.edge[7, 0] = d.x.shape[0] - 2                          # an edge from the last proposal into the first
    cases.append(("crossing edge", d))
    d = _with_big_proposal(_ragged(yv, 1500, 22, lo=3, hi=30), 700, 16 * _waves + 36)
    cases.append(("oversize proposal", d))
    for name, d in cases:
        with torch.no_grad():
            with _mode(2):
                got = model(d, None)[0].clone()
            with _mode(0):
                want = model(d, None)[0].clone()
        assert torch.equal(got, want), name
    # and on an ordinary batch the two paths are NOT the same arithmetic (the test above would be vacuous otherwise)
    with torch.no_grad():
        with _mode(2):
            got = model(base, None)[0].clone()
        with _mode(0):
            want = model(base, None)[0].clone()
    assert not torch.equal(got, want)
    mx, rms = _rel(got, want)
    assert mx <= RTOL_BF16 and rms <= RMS_BF16


def test_cfg5_full_size_local_conv_stack_vs_fp32_path():
    """configs[4] (N = 200 k / E = 1.2 M / P = 8000, n_blocks 4): the default bf16 forward takes the one-launch conv stack;
    against the fp32 HIP forward (itself pinned to 1e-4 by the other tests), deterministic, and block-diagonal: the two
    proposal halves run separately give the whole graph's logits up to the bf16 noise of a different tiling."""
    yv = _yv()
    data, slices, optkw, _ = yv.config("5")
    model = gu.fill_state_(yv.SparseCADGCN(yv.Opt(**optkw)), 9).cuda().eval()
    with torch.no_grad():
        want = model(data, slices)[0].clone()
        model.set_eval_precision("bf16")
        with _mode(1):
            got = model(data, slices)[0].clone()
            again = model(data, slices)[0].clone()
        with _mode(0):
            layers = model(data, slices)[0].clone()
    assert torch.equal(got, again)
    assert not torch.equal(got, layers)                      # the local path ran (different arithmetic)
    mx, rms = _rel(got, want)
    assert mx <= RTOL_BF16 and rms <= RMS_BF16, (mx, rms)
    mx, rms = _rel(got, layers)
    assert mx <= RTOL_BF16 and rms <= RMS_BF16, (mx, rms)


# ----------------------------------------------------------------------------------------------------------------------
# Round 6: the locality property decided BEFORE the forward is enqueued (yolat_batch_locality / yolat_locality), and the
# COO instantiation of the kernel whose tiles sort their own edges (no global COO -> CSR build).
# Reference: a proposal's edges are ONE contiguous block of the edge list (Datasets/graph_dict3.py:725, :752-764), bbox_idx
# is sorted (:732), an edge never leaves its proposal (:582-600,733).
# ----------------------------------------------------------------------------------------------------------------------

def _stages(fn, n=1):
    """names of the C-side stages `fn` ran through (the HIP-event stage profiler of the eval plan)"""
    from yolat_vectorgraphicsrecognition_amd._lib import lib
    lib.yolat_profile_reset()
    lib.yolat_profile_enable(1)
    try:
        for _ in range(n):
            fn()
        torch.cuda.synchronize()
    finally:
        lib.yolat_profile_enable(0)
    names = []
    buf = ctypes.create_string_buffer(128)
    ms, calls, fl, by = ctypes.c_float(), ctypes.c_int(), ctypes.c_double(), ctypes.c_double()
    for i in range(lib.yolat_profile_count()):
        lib.yolat_profile_get(i, buf, 128, ctypes.byref(ms), ctypes.byref(calls), ctypes.byref(fl), ctypes.byref(by))
        names.append(buf.value.decode())
    lib.yolat_profile_reset()
    return names


def _shuffle_inside_proposals(d, seed):
    """the edge list permuted INSIDE each proposal's block (still grouped by proposal): what a dataset item looks like —
    a proposal's edges in pick-up order, not in destination order"""
    g = torch.Generator().manual_seed(seed)
    owner = d.bbox_idx[d.edge[:, 1]]
    key = owner.double() + torch.rand(owner.shape[0], generator=g).double() * 0.5
    perm = torch.argsort(key, stable=True)
    d.edge = d.edge[perm].contiguous()
    d.e_attr = d.e_attr[perm].contiguous()
    return d


def _run_stack_coo(yv, model, d):
    """yolat_conv_stack_local_bf16_coo on the raw edge list: (feats, Z, flag, status)"""
    from yolat_vectorgraphicsrecognition_amd import ops
    from yolat_vectorgraphicsrecognition_amd._lib import lib, check
    with torch.no_grad():
        model(d, None)
    plan = model._yolat_plan
    h, base = plan._desc_h, plan._desc
    N, P, E = d.x.shape[0], d.bbox.shape[0], d.edge.shape[0]
    dev = torch.device("cuda")
    D, F = base.C * base.n_blocks_out, base.F
    feats = torch.full((N, D), float("nan"), dtype=torch.bfloat16, device=dev)
    Z = torch.full((P, 2 * (F + D)), float("nan"), dtype=torch.float32, device=dev)
    flag = torch.zeros(1, dtype=torch.int32, device=dev)
    status = torch.zeros(1, dtype=torch.int32, device=dev)
    x, edge, attr, bb = d.x.to(dev).contiguous(), d.edge.to(dev), d.e_attr.to(dev).contiguous(), d.bbox_idx.to(dev)
    need = int(lib.yolat_batch_locality_workspace_bytes(N, E, P))
    ws = torch.empty(need + 16, dtype=torch.uint8, device=dev)
    check(lib.yolat_conv_stack_local_bf16_coo(ctypes.byref(h), h.conv_local, x.data_ptr(), x.stride(0), edge.data_ptr(),
                                              edge.stride(0), edge.stride(1), attr.data_ptr(), bb.data_ptr(), N, E, P,
                                              feats.data_ptr(), D, Z.data_ptr(), Z.stride(0), flag.data_ptr(),
                                              status.data_ptr(), ws.data_ptr(), ws.numel(),
                                              torch.cuda.current_stream().cuda_stream), "yolat_conv_stack_local_bf16_coo")
    torch.cuda.synchronize()
    return feats, Z, int(flag.item()), int(status.item()), (F, D)


@pytest.mark.parametrize("blocks,blocks_out,P,seed,shape", [
    (2, 2, 300, 1, dict(lo=2, hi=40, edge_factor=2.1)),                 # ragged, nodes without in-edges
    (4, 2, 1500, 2, dict(lo=25, hi=25, edges_per_proposal=150)),       # cfg-5-like
    (3, 1, 64, 4, dict(lo=30, hi=60, edges_per_proposal=500)),         # one proposal per tile, ~500 edges each
    (2, 2, 7, 5, dict(lo=3, hi=9, edge_factor=1.0)),                    # a handful of proposals
])
def test_conv_stack_coo_tiles_sort_their_edges_like_the_global_csr_build(blocks, blocks_out, P, seed, shape):
    """The COO instantiation (stable counting sort of each tile's edges in LDS) against the prepared-graph instantiation
    (global stable sort, graph.hip): the same edges in the same slots -> feats and the pooled rows BIT-identical, for an
    edge list in pick-up order inside each proposal and with duplicate edges."""
    yv = _yv()
    optkw = dict(n_classes=17, n_blocks=blocks, n_blocks_out=blocks_out)
    d = _ragged(yv, P, 300 + seed, **shape)
    # duplicates of one proposal's edges, appended inside that proposal's block
    owner = d.bbox_idx[d.edge[:, 1]]
    dup = (owner == int(owner[len(owner) // 2])).nonzero()[:3, 0]
    d.edge = torch.cat([d.edge, d.edge[dup]], 0)
    d.e_attr = torch.cat([d.e_attr, d.e_attr[dup] + 0.01], 0)
    d = _shuffle_inside_proposals(d, seed)
    model = _model(yv, optkw, 70 + seed)
    want_feats, want_Z, flag, (F, D) = _run_stack(yv, model, d)
    feats, Z, flag2, status, _ = _run_stack_coo(yv, model, d)
    assert flag == 0 and flag2 == 0 and status == 0
    assert torch.equal(feats.view(torch.int16), want_feats.view(torch.int16))
    for lo, hi in ((0, F + D), (2 * F + D, 2 * (F + D))):
        assert torch.equal(Z[:, lo:hi], want_Z[:, lo:hi])


def test_batch_locality_reports_structure_and_violations(_waves):
    yv = _yv()
    from yolat_vectorgraphicsrecognition_amd._lib import lib, check, Locality

    def examine(d):
        N, P, E = d.x.shape[0], d.bbox.shape[0], d.edge.shape[0]
        edge, bb = d.edge.cuda(), d.bbox_idx.cuda()
        ws = torch.empty(int(lib.yolat_batch_locality_workspace_bytes(N, E, P)) + 16, dtype=torch.uint8, device="cuda")
        info = torch.full((4,), -7, dtype=torch.int32, device="cuda")
        check(lib.yolat_batch_locality(edge.data_ptr(), edge.stride(0), edge.stride(1), bb.data_ptr(), N, E, P,
                                       info.data_ptr(), ws.data_ptr(), ws.numel(), torch.cuda.current_stream().cuda_stream),
              "yolat_batch_locality")
        return info.tolist()

    d = _shuffle_inside_proposals(_ragged(yv, 700, 31, lo=3, hi=30), 1)
    owner = d.bbox_idx[d.edge[:, 1]].numpy()
    want_n = int(np.bincount(d.bbox_idx.numpy()).max())
    want_e = int(np.bincount(owner, minlength=700).max())
    assert examine(d) == [0, want_n, want_e, 0]
    T = 16 * _waves
    for nodes, edges, fits in ((T, 8 * T, 1), (T + 1, 8 * T, 0), (T, 8 * T + 1, 0)):
        loc = Locality(1, 0, nodes, edges)
        assert lib.yolat_conv_local_fits(ctypes.byref(loc), 700) == fits
    assert lib.yolat_conv_local_fits(ctypes.byref(Locality(0, 0, 3, 3)), 700) == 0          # not examined
    assert lib.yolat_conv_local_fits(ctypes.byref(Locality(1, 2, 3, 3)), 700) == 0          # a violation
    # an edge that joins two proposals
    d2 = _ragged(yv, 40, 9, lo=3, hi=20)
    d2.edge = d2.edge.clone()
    d2.edge[5, 0] = d2.x.shape[0] - 1
    assert examine(d2)[0] == 2
    # the edge list not grouped by proposal: the first proposal's first edge moved to the end
    d3 = _ragged(yv, 40, 10, lo=3, hi=20)
    d3.edge = torch.cat([d3.edge[1:], d3.edge[:1]], 0)
    d3.e_attr = torch.cat([d3.e_attr[1:], d3.e_attr[:1]], 0)
    assert examine(d3)[0] == 1
    # malformed ids are reported with the status bits of the graph preparation
    d4 = _ragged(yv, 40, 11, lo=3, hi=20)
    d4.edge = d4.edge.clone()
    d4.edge[3, 1] = d4.x.shape[0] + 5
    assert examine(d4)[3] & 1


def _fresh(d):
    """the same batch as new tensor objects (a batch that has not been examined)"""
    for k in ("x", "edge", "e_attr", "bbox_idx", "bbox"):
        setattr(d, k, getattr(d, k).clone())
    d.__dict__.pop("_yolat_stage", None)
    return d


def test_forward_examines_a_resident_batch_once_and_drops_graph_prep_and_gated_launches():
    """A resident, proposal-local batch: the first forward examines it (one host read), every forward then runs local prep +
    ONE conv launch — no destination sort, no gated fall-back launches — with logits bit-identical to the gated form; an
    in-place edit of the batch is seen (same invalidation as the stage cache) and an unfit batch goes straight to the
    per-layer launches, bit-identical to YOLAT_CONV_LOCAL=0."""
    yv = _yv()
    from yolat_vectorgraphicsrecognition_amd import plan as plan_mod
    optkw = dict(n_classes=17, n_blocks=3, n_blocks_out=2)
    model = _model(yv, optkw, 5)
    d = _shuffle_inside_proposals(_ragged(yv, 1500, 21, lo=3, hi=30), 3)
    for k in ("x", "edge", "e_attr", "bbox_idx", "bbox"):
        setattr(d, k, getattr(d, k).cuda())

    def fwd():
        with torch.no_grad():
            return model(d, None)[0].clone()

    with _mode(2):
        plan_mod.LOCALITY_CACHE = False
        try:
            names_gated = _stages(fwd)
            gated = fwd()
        finally:
            plan_mod.LOCALITY_CACHE = True
        assert any("gated fall-back" in n for n in names_gated) and any(n.startswith("conv_local") for n in names_gated)
        got = fwd()
        names = _stages(fwd, 3)
        assert any(n.startswith("conv_local") for n in names), names
        assert not any("gated fall-back" in n or "node_uv_bf16" in n or "edge_uv" in n or "pool_prepare" in n for n in names), names
        assert torch.equal(got, gated)                       # the tile sort is stable: same arithmetic as the CSR form
        assert torch.equal(fwd(), got)
        model._yolat_plan.check_status()
        assert len(model._yolat_plan._loc) == 1              # examined once
        # in-place edit: an edge now leaves its proposal -> re-examined -> per-layer path, no conv_local launch
        d.edge[7, 0] = d.x.shape[0] - 2
        names = _stages(fwd)
        assert not any(n.startswith("conv_local") for n in names) and not any("gated" in n for n in names), names
        unfit = fwd()
        assert len(model._yolat_plan._loc) == 2
    with _mode(0):
        want = fwd()
    assert torch.equal(unfit, want)
    model._yolat_plan.check_status()


def test_an_ungrouped_edge_list_still_gives_the_oracle_result():
    """the edge list in RANDOM order (not grouped by proposal; nothing the reference's datasets produce, but a legal
    edge_index): examined -> unfit -> per-layer launches; logits within the bf16 band of the CPU oracle and bit-identical to
    the forward with the one-launch stack switched off"""
    yv = _yv()
    optkw = dict(n_classes=17, n_blocks=2, n_blocks_out=2)
    d = _ragged(yv, 1200, 77, lo=3, hi=30)
    perm = torch.randperm(d.edge.shape[0], generator=torch.Generator().manual_seed(5))
    d.edge, d.e_attr = d.edge[perm].contiguous(), d.e_attr[perm].contiguous()
    model = _model(yv, optkw, 8)
    ref = gu.fill_state_(orc.SparseCADGCN(orc.Opt(**optkw)), 8).eval()
    for k in ("x", "edge", "e_attr", "bbox_idx", "bbox"):
        setattr(d, k, getattr(d, k).cuda())
    with torch.no_grad():
        with _mode(2):
            got = model(d, None)[0].clone()
            assert model._yolat_plan._loc and list(model._yolat_plan._loc.values())[0][2].flags & 1
        with _mode(0):
            layers = model(d, None)[0].clone()
        for k in ("x", "edge", "e_attr", "bbox_idx", "bbox"):
            setattr(d, k, getattr(d, k).cpu())
        want = ref(d, None)[0]
    assert torch.equal(got, layers)
    mx, rms = _rel(got, want)
    assert mx <= RTOL_BF16 and rms <= RMS_BF16, (mx, rms)
    model._yolat_plan.check_status()


def test_a_stale_locality_record_is_reported_in_the_status_word():
    """yolat_forward_eval_bf16_loc with a record that claims the property for a batch that does not have it (what a caller
    gets who edits a batch behind the cache's back): YOLAT_STATUS_NOT_LOCAL is raised — by the local prep for a crossing
    edge / an ungrouped list, by the conv kernel for a proposal that does not fit — and check_status raises."""
    yv = _yv()
    from yolat_vectorgraphicsrecognition_amd import ops
    from yolat_vectorgraphicsrecognition_amd._lib import lib, check, Locality
    optkw = dict(n_classes=17, n_blocks=2, n_blocks_out=2)
    model = _model(yv, optkw, 3)
    good = _ragged(yv, 1100, 41, lo=3, hi=20)
    with torch.no_grad():
        model(good, None)
    plan = model._yolat_plan
    cases = []
    d = _ragged(yv, 1100, 42, lo=3, hi=20)
    d.edge = d.edge.clone()
    d.edge[5, 0] = d.x.shape[0] - 1
    cases.append(d)
    d = _ragged(yv, 1100, 43, lo=3, hi=20)
    d.edge = torch.cat([d.edge[1:], d.edge[:1]], 0)
    cases.append(d)
    cases.append(_with_big_proposal(_ragged(yv, 1100, 44, lo=3, hi=20), 500, 150))
    for d in cases + [good]:
        N, P, E = d.x.shape[0], d.bbox.shape[0], d.edge.shape[0]
        x, edge, attr, bb = d.x.cuda(), d.edge.cuda(), d.e_attr.cuda(), d.bbox_idx.cuda()
        need = int(lib.yolat_forward_eval_bf16_workspace_bytes(ctypes.byref(plan._desc_h), N, E, P))
        ws = torch.empty(need + 4096, dtype=torch.uint8, device="cuda")
        status = torch.zeros(1, dtype=torch.int32, device="cuda")
        logits = torch.empty(P, 17, device="cuda")
        lie = Locality(1, 0, 8, 8)
        with _mode(2):
            check(lib.yolat_forward_eval_bf16_loc(ctypes.byref(plan._desc_h), x.data_ptr(), x.stride(0), edge.data_ptr(),
                                                  edge.stride(0), edge.stride(1), attr.data_ptr(), bb.data_ptr(), None, N, E, P,
                                                  logits.data_ptr(), logits.stride(0), ws.data_ptr(), ws.numel(),
                                                  status.data_ptr(), ctypes.byref(lie), 0,
                                                  torch.cuda.current_stream().cuda_stream), "yolat_forward_eval_bf16_loc")
        torch.cuda.synchronize()
        g = ops.Graph()
        g.status = status
        if d is good:
            assert int(status.item()) == 0
            assert g.check_status()
        else:
            assert int(status.item()) & 8
            with pytest.raises(ValueError, match="proposal-local"):
                g.check_status()


def test_batches_without_edges_and_without_children_take_the_ordinary_paths():
    """edge cases of the round-6 paths: a batch with NO edges (E = 0: never vouched — nothing to sort; the forward matches the
    oracle through whichever path the mode picks), and predict() on a tree without children / with a single image"""
    yv = _yv()
    optkw = dict(n_classes=17, n_blocks=2, n_blocks_out=2)
    d = _ragged(yv, 1100, 5, lo=2, hi=6, edges_per_proposal=0)
    assert d.edge.shape[0] == 0
    model = _model(yv, optkw, 12)
    ref = gu.fill_state_(orc.SparseCADGCN(orc.Opt(**optkw)), 12).eval()
    with torch.no_grad():
        want = ref(d, None)[0]
        for mode in (2, 0):
            with _mode(mode):
                d.__dict__.pop("_yolat_stage", None)
                got = model(d, None)[0].clone()
            mx, rms = _rel(got, want)
            assert mx <= RTOL_BF16 and rms <= RMS_BF16, (mode, mx, rms)
    model._yolat_plan.check_status()
    # predict() with a tree of roots only (no children anywhere): one submission, rows = the roots in order
    data, slices = yv.synth_batch(2, 31, num_proposals=60, nodes_lo=4, nodes_hi=12, edge_factor=1.4, with_roots=True)
    for r in data.roots:
        r.children = []
    m32 = gu.fill_state_(yv.SparseCADGCN(yv.Opt(**optkw)), 13).cuda().eval()
    with torch.no_grad():
        out = m32.predict(data, slices)
    assert out[0].shape[0] == len(data.roots) == len(out[3]) and out[4][-1] == len(data.roots)


def test_cfg5_full_size_vouched_forward_equals_the_gated_forward_bit_for_bit():
    """configs[4] at FULL size: the forward with the locality decided before enqueue (local prep + tile-local sort, no gated
    launches) against the forward that finds it out on the device (global destination sort + gated fall-back): the same
    logits bit for bit, and the examination reports the batch's structure (25 nodes / 150 edges per proposal, no violation)."""
    yv = _yv()
    from yolat_vectorgraphicsrecognition_amd import plan as plan_mod
    data, slices, optkw, _ = yv.config("5")
    for k in ("x", "edge", "e_attr", "bbox_idx", "bbox"):
        data[k] = data[k].cuda()
    model = gu.fill_state_(yv.SparseCADGCN(yv.Opt(**optkw)), 9).cuda().eval().set_eval_precision("bf16")
    with torch.no_grad():
        with _mode(1):
            vouched = model(data, slices)[0].clone()
            (loc,) = [v[2] for v in model._yolat_plan._loc.values()]
            assert (loc.known, loc.flags, loc.max_nodes, loc.max_edges) == (1, 0, 25, 150)
            plan_mod.LOCALITY_CACHE = False
            try:
                gated = model(data, slices)[0].clone()
            finally:
                plan_mod.LOCALITY_CACHE = True
    assert torch.equal(vouched, gated)
    model._yolat_plan.check_status()
