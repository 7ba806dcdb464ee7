"""GPU tests (-m gpu) of the one-call training step (csrc/train_plan.hip, yolat_train_step, trainer.TrainPlan) against the
Python schedule it restates (engine.model_fwd / model_bwd + FlatAdam): the same kernels on the same operands in the same
order per stream, so loss, every gradient, every parameter, the Adam moments and the BatchNorm buffers must be BIT-identical
after several steps.  The Python schedule itself is pinned to the reference's golden training step and to the fp64 CPU
oracle at full size by tests/test_gpu_model.py / tests/test_gpu_configs.py — which now run through the plan as well
(Trainer.step takes it by default).

Reference: the loop body of cad_recognition/train.py:263-284 over architecture3cc_rpn_gp_iter2.py:106-137,358-379."""
import copy

import pytest
import torch

import golden_util as gu

pytestmark = pytest.mark.gpu


def _yv():
    import yolat_vectorgraphicsrecognition_amd as yv
    return yv


def _pair(yv, optkw, seed, precision="fp32"):
    opt = yv.Opt(**optkw)
    a = gu.fill_state_(yv.SparseCADGCN(opt), seed).cuda()
    b = copy.deepcopy(a)
    ta = yv.Trainer(a, opt, lr=1e-3, weight_decay=1e-4, precision=precision)
    tb = yv.Trainer(b, opt, lr=1e-3, weight_decay=1e-4, precision=precision)
    return ta, tb


def _state(tr):
    out = {"param": tr.flat.param.clone(), "grad": tr.flat.grad.clone(), "m": tr.optimizer.exp_avg.clone(),
           "v": tr.optimizer.exp_avg_sq.clone(), "step": tr.optimizer.step_count}
    for k, b in tr.model.named_buffers():
        out["buf/" + k] = b.clone()
    return out


def _same(sa, sb, what):
    assert sa.keys() == sb.keys()
    for k in sa:
        if isinstance(sa[k], torch.Tensor):
            assert torch.equal(sa[k], sb[k]), "%s: %s differs" % (what, k)
        else:
            assert sa[k] == sb[k], (what, k)


def _run(yv, ta, tb, batches, steps=3):
    """ta through the plan, tb through the Python schedule; bit-identical after every step"""
    from yolat_vectorgraphicsrecognition_amd import trainer as T
    for i in range(steps):
        data, slices = batches[i % len(batches)]
        T.TRAIN_PLAN = True
        la = ta.step(data, slices)
        T.TRAIN_PLAN = False
        try:
            lb = tb.step(data, slices)
        finally:
            T.TRAIN_PLAN = True
        assert torch.equal(la, lb), "loss of step %d" % i
        _same(_state(ta), _state(tb), "step %d" % i)
    assert ta.plan_steps == steps and tb.plan_steps == 0
    ta.model.check_last_status()


@pytest.mark.parametrize("name,optkw,gkw,precision", [
    ("E >= 2N: factorised forward", dict(n_classes=17), dict(num_proposals=120, nodes_lo=8, nodes_hi=20, edges_per_proposal=60), "fp32"),
    ("N <= E < 2N: gathered forward, factorised backward (cfg 3 / 4)", dict(n_classes=22),
     dict(num_proposals=300, nodes_lo=4, nodes_hi=24, edge_factor=1.2, n_classes=22, augmented=True), "fp32"),
    ("n_blocks 4: layers below the concat", dict(n_classes=17, n_blocks=4, n_blocks_out=2),
     dict(num_proposals=90, nodes_lo=6, nodes_hi=18, edges_per_proposal=50), "fp32"),
    ("bf16 storage of the per-edge tensors", dict(n_classes=17, n_blocks=3, n_blocks_out=2),
     dict(num_proposals=150, nodes_lo=10, nodes_hi=20, edges_per_proposal=70), "bf16"),
    ("P >= 1024: the bf16x6 classifier GEMMs", dict(n_classes=17),
     dict(num_proposals=1300, nodes_lo=3, nodes_hi=9, edge_factor=1.5), "fp32"),
])
def test_plan_step_is_bit_identical_to_the_python_schedule(name, optkw, gkw, precision):
    yv = _yv()
    ta, tb = _pair(yv, optkw, 11, precision)
    batches = []
    for s in (1, 2):
        d, sl = yv.synth_batch(2, 40 + s, **gkw)
        for k in ("x", "edge", "e_attr", "bbox_idx", "bbox", "labels"):
            d[k] = d[k].cuda()
        batches.append((d, sl))
    _run(yv, ta, tb, batches)


def test_plan_step_one_stream_and_prepared_graph_batches():
    """engine.SIDE_STREAM off (one stream: the same launches in program order) and batches that carry a prepared graph
    (collate_to_device(csr=True)): still bit-identical to the Python schedule on the same batches"""
    yv = _yv()
    from yolat_vectorgraphicsrecognition_amd import engine
    items = [yv.synth_graph(num_proposals=60 + 9 * i, nodes_lo=5, nodes_hi=16, edges_per_proposal=40, seed=70 + i) for i in range(3)]
    ta, tb = _pair(yv, dict(n_classes=17), 5)
    old = engine.SIDE_STREAM
    engine.SIDE_STREAM = False
    try:
        _run(yv, ta, tb, [yv.collate_to_device(items, csr=True)], steps=2)
    finally:
        engine.SIDE_STREAM = old
    ta, tb = _pair(yv, dict(n_classes=17), 6)
    _run(yv, ta, tb, [yv.collate_to_device(items, csr=True), yv.collate_to_device(items[:2])], steps=3)


def test_plan_declines_what_it_does_not_cover_and_the_python_schedule_takes_the_step():
    yv = _yv()
    from yolat_vectorgraphicsrecognition_amd import engine
    # (a) fewer edges than nodes: the gathered backward of the first edge Linear is not in the plan
    opt = yv.Opt(n_classes=17)
    tr = yv.Trainer(gu.fill_state_(yv.SparseCADGCN(opt), 3).cuda(), opt)
    d, sl = yv.synth_batch(1, 9, num_proposals=80, nodes_lo=6, nodes_hi=12, edges_per_proposal=4)
    assert d.edge.shape[0] < d.x.shape[0]
    assert torch.isfinite(tr.step(d, sl)) and tr.plan_steps == 0
    # (b) dropout in prediction_cls.1 (torch_nn.py:67-68)
    opt = yv.Opt(n_classes=17, dropout=0.3)
    tr = yv.Trainer(gu.fill_state_(yv.SparseCADGCN(opt), 3).cuda(), opt)
    d, sl = yv.synth_batch(1, 10, num_proposals=80, nodes_lo=6, nodes_hi=12, edges_per_proposal=30)
    assert torch.isfinite(tr.step(d, sl)) and tr.plan_steps == 0
    # (c) a flipped schedule flag
    opt = yv.Opt(n_classes=17)
    tr = yv.Trainer(gu.fill_state_(yv.SparseCADGCN(opt), 3).cuda(), opt)
    engine.FUSED_FUSION_TRAIN = False
    try:
        assert torch.isfinite(tr.step(d, sl)) and tr.plan_steps == 0
    finally:
        engine.FUSED_FUSION_TRAIN = True
    assert torch.isfinite(tr.step(d, sl)) and tr.plan_steps == 1


def test_plan_step_reports_malformed_batches_before_the_update():
    """the first step checks the input-validity word (ids in range, bbox_idx sorted) BEFORE Adam applies the update"""
    yv = _yv()
    opt = yv.Opt(n_classes=17)
    tr = yv.Trainer(gu.fill_state_(yv.SparseCADGCN(opt), 3).cuda(), opt)
    d, sl = yv.synth_batch(1, 12, num_proposals=80, nodes_lo=6, nodes_hi=12, edges_per_proposal=30)
    d.edge = d.edge.clone()
    d.edge[3, 1] = d.x.shape[0] + 7
    before = tr.flat.param.clone()
    with pytest.raises(IndexError):
        tr.step(d, sl)
    assert torch.equal(tr.flat.param, before)


def test_plan_host_time_per_step_is_a_fraction_of_the_python_schedule():
    """what the plan is for: the host side of a step (enqueue only, GPU idle-waited outside the timed region)"""
    import time
    yv = _yv()
    from yolat_vectorgraphicsrecognition_amd import trainer as T
    opt = yv.Opt(n_classes=17)
    d, sl = yv.synth_batch(2, 3, num_proposals=400, nodes_lo=4, nodes_hi=30, edge_factor=1.2, augmented=True)
    for k in ("x", "edge", "e_attr", "bbox_idx", "bbox", "labels"):
        d[k] = d[k].cuda()
    host = {}
    for flag in (True, False):
        T.TRAIN_PLAN = flag
        try:
            tr = yv.Trainer(gu.fill_state_(yv.SparseCADGCN(opt), 3).cuda(), opt)
            for _ in range(3):
                tr.step(d, sl)
            torch.cuda.synchronize()
            ts = []
            for _ in range(10):
                t0 = time.perf_counter()
                tr.step(d, sl)
                ts.append(time.perf_counter() - t0)
                torch.cuda.synchronize()
            ts.sort()
            host[flag] = ts[len(ts) // 2]
        finally:
            T.TRAIN_PLAN = True
    assert host[True] < 0.5 * host[False], host


@pytest.mark.parametrize("cfg,precision", [("3", "fp32"), ("4", "fp32"), ("5", "fp32"), ("5", "bf16")])
def test_plan_step_is_bit_identical_at_the_baseline_configurations_full_size(cfg, precision):
    """BASELINE.json configs[2] / [3] / [4] at FULL size (cfg 3: 4 x 2000 proposals, N ~ 175 k; cfg 4: 32 Diagrams-style
    graphs; cfg 5: N = 200 k / E = 1.2 M / P = 8000, n_blocks 4): two steps through yolat_train_step against two steps of the
    Python schedule — loss, flat gradient, parameters, Adam moments, BatchNorm buffers bit for bit.  (The Python schedule's
    gradients are pinned to the float64 oracle at these sizes by tests/test_gpu_configs.py.)"""
    yv = _yv()
    data, slices, optkw, _ = yv.config(cfg)
    for k in ("x", "edge", "e_attr", "bbox_idx", "bbox", "labels"):
        data[k] = data[k].cuda()
    ta, tb = _pair(yv, optkw, 21, precision)
    _run(yv, ta, tb, [(data, slices)], steps=2)
