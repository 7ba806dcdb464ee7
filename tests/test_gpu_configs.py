"""GPU parity tests (-m gpu) at BASELINE.json's full configuration sizes, and of the state
caches the HIP kernels bypass (round-2 review items):

  cfg 1  Floorplans-sized graph (P=2000, N~44k, E~53k ragged, ~1.2 edges per node): eval forward vs the CPU oracle
  cfg 3  4 collated cfg-1-style graphs (N~175k): one training step — loss and every gradient vs the fp32 oracle
  cfg 4  Diagrams-style batch (32 graphs, K=22): Trainer.step, loss / gradients vs the oracle on the same batch
  cfg 5  N=200k / E=1.2M / n_blocks=4: training step — determinism, finiteness, block-diagonal consistency
"""
import os

import numpy as np
import pytest
import torch

import golden_util as gu
from oracle import oracle_torch as orc

pytestmark = pytest.mark.gpu

RTOL_FWD = 1e-4


def _yv():
    import yolat_vectorgraphicsrecognition_amd as yv
    return yv


def _model(yv, optkw, seed):
    return gu.fill_state_(yv.SparseCADGCN(yv.Opt(**optkw)), seed).cuda()


def _elementwise(got, want, rtol, name):
    got, want = got.detach().cpu().double(), want.detach().cpu().double()
    scale = float(want.abs().max())
    bad = (got - want).abs() > rtol * want.abs() + 0.1 * rtol * scale
    assert not bool(bad.any()), "%s: %d of %d elements outside |d| <= %g*|want| + %g*scale (worst %.3e, scale %.3e)" % (
        name, int(bad.sum()), bad.numel(), rtol, 0.1 * rtol, float((got - want).abs().max()), scale)


def _oracle_train(ref, optkw, data, dtype=torch.float32):
    yv = _yv()
    d = yv.Data(x=data.x.to(dtype), pos=data.pos)
    for k in ("edge", "bbox_idx", "bbox", "labels"):
        d[k] = data[k]
    d.e_attr = data.e_attr.to(dtype)
    out = ref(d, None)
    loss = orc.DetectionLoss(orc.Opt(**optkw))(out, d)["loss"]
    loss.backward()
    return out[0].detach(), loss.detach()


class _KinkReLU(torch.autograd.Function):
    """ReLU whose sub-gradient switches at `thr` instead of 0 (the value is the ordinary ReLU)."""

    @staticmethod
    def forward(ctx, z, thr):
        ctx.save_for_backward(z > thr)
        return z.clamp_min(0)

    @staticmethod
    def backward(ctx, g):
        (m,) = ctx.saved_tensors
        return g * m, None


def _kink_band(make_ref, optkw, data, tau=4e-6):
    """|grad(ReLU' switching at +tau) - grad(ReLU' switching at -tau)| per parameter, from the fp64 oracle.

    A pre-activation closer to 0 than the fp32 forward rounding (BatchNorm outputs are O(1); 4e-6 is ~32 ulp) gets its
    ReLU derivative from rounding, not from the math: either value is a correct sub-gradient.  Where few rows feed a sum
    (a [P,1024] block with P of a few hundred, a node with one in-edge that holds a proposal's arg-max) ONE such element
    moves a bias gradient by percents — measured on seed 6 of the random-shape test: the fp64 oracle has |z| = 2.6e-8 in
    head.gconv.nn.5, and flipping that single element reproduces the device's deviation to 3 digits (nn.4.bias 2.31e-2,
    nn.3.weight 1.49e-2 relative).  The band is the oracle's own statement of that ambiguity."""
    grads = []
    for thr in (tau, -tau):
        ref = make_ref().double()
        ref.train()
        for m in ref.modules():
            if isinstance(m, torch.nn.ReLU):
                m.forward = (lambda z, thr=thr: _KinkReLU.apply(z, thr))
        _oracle_train(ref, optkw, data, torch.float64)
        grads.append({n: p.grad for n, p in ref.named_parameters()})
    return {n: (grads[0][n] - grads[1][n]).abs() for n in grads[0]}


def _grad_check(model_grads, ref, rtol, name, band=None):
    rp = dict(ref.named_parameters())
    gmax = max(float(p.grad.abs().max()) for p in ref.parameters())
    for n, g in model_grads.items():
        a, b = g.cpu().double(), rp[n].grad.double()
        scale = float(b.abs().max())
        if band is not None:
            # outside the sub-gradient band (twice its width: flips need not be additive) the usual tolerance holds
            err = float(((a - b).abs() - 2.0 * band[n]).clamp_min(0).max())
        else:
            err = float((a - b).abs().max())
        # floor = a fraction of the LARGEST gradient: the network is discontinuous (per-proposal arg-max, ReLU kinks); a
        # near-tie that fp32 rounding resolves the other way — which one changes with every summation order — moves a
        # small-gradient tensor by a discrete amount of that order (same rule as tests/test_gpu_model.py)
        assert err <= rtol * scale + 2e-3 * gmax, "%s %s: err %.3e scale %.3e" % (name, n, err, scale)


def test_cfg1_floorplans_sized_eval_forward_matches_oracle():
    """BASELINE.json configs[0] at full size: ragged proposals, ~1.2 directed edges per node (the many-node-per-pass
    tiles of the edge kernel), fp32 eval forward vs the op-for-op CPU oracle; determinism; the bf16-storage path."""
    yv = _yv()
    data, slices, optkw, _ = yv.config("1")
    N, E, P = data.x.shape[0], data.edge.shape[0], data.bbox.shape[0]
    assert P == 2000 and 40000 < N < 48000 and 1.15 * N < E < 1.3 * N
    model = _model(yv, optkw, 11)
    ref = gu.fill_state_(orc.SparseCADGCN(orc.Opt(**optkw)), 11)
    model.eval(); ref.eval()
    with torch.no_grad():
        got = model(data, slices)[0]
        want = ref(data, None)[0]
    model.check_last_status()
    assert got.shape == (P, optkw["n_classes"])
    assert float((got.cpu() - want).abs().max()) <= RTOL_FWD * float(want.abs().max())
    _elementwise(got, want, RTOL_FWD, "cfg 1 logits")
    with torch.no_grad():
        data._yolat_stage = None
        again = model(data, slices)[0]
        sched = model.forward_scheduled(data, slices)[0]
    assert torch.equal(got, again)
    _elementwise(sched, want, RTOL_FWD, "cfg 1 logits (scheduled kernels)")
    model.set_eval_precision("bf16")
    with torch.no_grad():
        g16 = model(data, slices)[0].cpu()
    model.set_eval_precision("fp32")
    rms = float(((g16 - want) ** 2).mean().sqrt() / (want ** 2).mean().sqrt())
    assert rms < 1e-2, rms
    assert float((g16 - want).abs().max()) <= 1.5e-2 * float(want.abs().max())
    assert float((g16.argmax(1) == want.argmax(1)).float().mean()) >= 0.99
    # ... and through the one-launch conv stack (P = 2000 takes it by default; forced here for clarity): the same contract
    import os
    old_mode = os.environ.get("YOLAT_CONV_LOCAL")
    os.environ["YOLAT_CONV_LOCAL"] = "2"
    try:
        model.set_eval_precision("bf16")
        with torch.no_grad():
            data._yolat_stage = None
            g16 = model(data, slices)[0].cpu()
        model.set_eval_precision("fp32")
    finally:
        if old_mode is None:
            os.environ.pop("YOLAT_CONV_LOCAL", None)
        else:
            os.environ["YOLAT_CONV_LOCAL"] = old_mode
    assert float(((g16 - want) ** 2).mean().sqrt() / (want ** 2).mean().sqrt()) < 1e-2
    assert float((g16 - want).abs().max()) <= 1.5e-2 * float(want.abs().max())
    assert float((g16.argmax(1) == want.argmax(1)).float().mean()) >= 0.99


def test_cfg3_full_size_train_step_matches_oracle_and_is_deterministic():
    """BASELINE.json configs[2] at full size (4 x 2000 proposals, N~175k): loss / logits / every gradient of one
    training forward+backward vs the fp32 CPU oracle (5e-3 band: at this size the fp32 oracle itself is 2.6e-3 off
    its fp64 twin on the BatchNorm-backward-heavy gradients), run-to-run bit identity, one Trainer step."""
    yv = _yv()
    data, slices, optkw, n_graphs = yv.config("3")
    assert n_graphs == 4 and data.bbox.shape[0] == 8000
    opt = yv.Opt(**optkw)
    model = _model(yv, optkw, 33)
    ref = gu.fill_state_(orc.SparseCADGCN(orc.Opt(**optkw)), 33)
    model.train(); ref.train()
    runs = []
    for _ in range(2):
        data._yolat_stage = None
        model.zero_grad()
        m2 = _model(yv, optkw, 33)          # fresh BatchNorm buffers for the second run
        m2.train()
        out = m2(data, slices)
        loss = yv.DetectionLoss(opt)(out, data)["loss"]
        loss.backward()
        runs.append((out[0].detach().clone(), loss.detach().clone(),
                     {n: p.grad.detach().clone() for n, p in m2.named_parameters()}))
    assert torch.equal(runs[0][0], runs[1][0]) and torch.equal(runs[0][1], runs[1][1])
    for n in runs[0][2]:
        assert torch.equal(runs[0][2][n], runs[1][2][n]), n
    rlogits, rloss = _oracle_train(ref, optkw, data)
    assert abs(float(runs[0][1]) - float(rloss)) <= RTOL_FWD * abs(float(rloss))
    assert float((runs[0][0].cpu() - rlogits).abs().max()) <= 5e-4 * float(rlogits.abs().max())
    # every tensor at its own scale against the float64 oracle (floor = that tensor's fp32-vs-fp64 oracle gap)
    g32 = {n: p.grad.detach().double() for n, p in ref.named_parameters()}
    _, _, g64 = _oracle_grads(optkw, 33, data, torch.float64)
    _grad_check_per_tensor(runs[0][2], g64, g32, 3e-3, "cfg 3")
    # the same step through the trainer (flat buffers + one-kernel Adam): same loss, finite parameters afterwards
    m3 = _model(yv, optkw, 33)
    tr = yv.Trainer(m3, opt, lr=2.5e-4, weight_decay=1e-5)
    data._yolat_stage = None
    l3 = tr.step(data, slices)
    assert torch.equal(l3, runs[0][1])
    assert bool(torch.isfinite(tr.flat.param).all())


def test_cfg4_diagrams_batch_train_step_matches_oracle():
    """BASELINE.json configs[3]'s per-rank workload: 32 Diagrams-style graphs (K = 22 classes), Trainer.step —
    loss and gradients vs the fp32 oracle on the same batch; parameters move, BatchNorm counters advance."""
    yv = _yv()
    data, slices, optkw, n_graphs = yv.config("4")
    assert n_graphs == 32 and optkw["n_classes"] == 22
    opt = yv.Opt(**optkw)
    model = _model(yv, optkw, 44)
    ref = gu.fill_state_(orc.SparseCADGCN(orc.Opt(**optkw)), 44)
    ref.train()
    before = {n: p.detach().clone() for n, p in model.named_parameters()}
    tr = yv.Trainer(model, opt, lr=2.5e-4, weight_decay=1e-5)
    loss = tr.step(data, slices)
    _, rloss = _oracle_train(ref, optkw, data)
    assert abs(float(loss) - float(rloss)) <= RTOL_FWD * abs(float(rloss))
    hip_grads = {n: tr.flat.grad_views[id(p)] for n, p in model.named_parameters()}
    g32 = {n: p.grad.detach().double() for n, p in ref.named_parameters()}
    _, _, g64 = _oracle_grads(optkw, 44, data, torch.float64)
    _grad_check_per_tensor(hip_grads, g64, g32, 3e-3, "cfg 4")
    moved = sum(int(not torch.equal(before[n], p.detach())) for n, p in model.named_parameters())
    assert moved == len(before)
    for n, b in model.named_buffers():
        if n.endswith("num_batches_tracked"):
            assert int(b) == 1, n


def _oracle_grads(optkw, seed, data, dtype):
    ref = gu.fill_state_(orc.SparseCADGCN(orc.Opt(**optkw)), seed).train()
    if dtype == torch.float64:
        ref = ref.double()
    logits, loss = _oracle_train(ref, optkw, data, dtype)
    return logits, loss, {n: p.grad.detach().double() for n, p in ref.named_parameters()}


def _grad_check_per_tensor(model_grads, g64, g32, rtol, name, gap_factor=2.0):
    """Every gradient tensor against the float64 oracle at ITS OWN scale: |d| <= rtol * max|g64_n| + gap_factor * gap_n
    with gap_n = max|g32_n - g64_n|, the distance between the fp32 and the fp64 run of the same CPU oracle on that very
    tensor — what fp32 arithmetic (summation order, near-tie arg-max / ReLU decisions) is worth for it.  No floor taken
    from the largest gradient of the model (the old 2e-3 * gmax): a small-gradient tensor (conv-layer BatchNorm betas,
    lin_r.bias) is held to its own size.  The one exception is a MATHEMATICALLY ZERO gradient — the bias of a Linear in
    front of a BatchNorm: its float64 value is ~1e-17, every fp32 implementation returns the rounding noise of a sum
    over E / N / P rows (measured, gpurun_out/r03a/gap_cfg*.txt: device 3e-9 .. 7e-7, fp32 CPU oracle 1e-9 .. 4e-7 at
    gmax 8e-2) — for those tensors alone the tolerance is the absolute noise level 2e-5 * gmax, a hundredth of the old
    floor.  Round 5: rtol 5e-3 -> 3e-3 and gap_factor 4 -> 2 — with the old pair the largest err / tol of any tensor was
    0.30 (cfg 3), 0.22 (cfg 4), 0.45 (cfg 5: prediction_cls.0.0.weight); YOLAT_TEST_REPORT_GRAD_MARGINS=1 prints them."""
    gmax = max(float(g.abs().max()) for g in g64.values())
    worst = []
    for n, g in model_grads.items():
        a, b = g.detach().cpu().double(), g64[n]
        scale = float(b.abs().max())
        gap = float((g32[n] - b).abs().max())
        err = float((a - b).abs().max())
        tol = rtol * scale + gap_factor * gap
        if scale < 1e-9 * gmax:
            tol += 2e-5 * gmax
        worst.append((err / max(tol, 1e-300), n, err, scale, gap))
    worst.sort(reverse=True)
    if os.environ.get("YOLAT_TEST_REPORT_GRAD_MARGINS"):          # (how much of the tolerance each check uses: tuning aid)
        print("\n%s: per-tensor gradient check, largest err / tol: %s" % (
            name, "; ".join("%s %.2f (err %.2e, scale %.2e, gap %.2e)" % (w[1], w[0], w[2], w[3], w[4]) for w in worst[:4])))
    bad = [w for w in worst if w[0] > 1.0]
    assert not bad, "%s: %s" % (name, "; ".join("%s err %.3e scale %.3e fp32-vs-fp64 gap %.3e" % (w[1], w[2], w[3], w[4])
                                                for w in bad[:6]))
    return worst


def test_cfg5_full_size_eval_forward_matches_oracle():
    """BASELINE.json configs[4] at FULL size (N=200k / E=1.2M / P=8000, n_blocks=4) against the CPU oracle at full size —
    the size at which the persistent wave-specialised edge kernel (bf16x6 layer 2), the bf16x6 node side and the
    rider-less fusion launch are the auto-selected variants.  fp32: per element 1e-4 (arch:106-137); bf16 storage:
    <= 1e-2 rms of the logits, element-wise maximum 1.5e-2, arg-max agreement >= 0.99."""
    yv = _yv()
    data, slices, optkw, _ = yv.config("5")
    N, E, P = data.x.shape[0], data.edge.shape[0], data.bbox.shape[0]
    assert (N, E, P) == (200000, 1200000, 8000) and optkw["n_blocks"] == 4
    model = _model(yv, optkw, 55).eval()
    ref = gu.fill_state_(orc.SparseCADGCN(orc.Opt(**optkw)), 55).eval()
    with torch.no_grad():
        want = ref(data, None)[0]
        got = model(data, slices)[0]
        sched = model.forward_scheduled(data, slices)[0]
    model.check_last_status()
    assert got.shape == (P, optkw["n_classes"])
    _elementwise(got, want, RTOL_FWD, "cfg 5 logits (eval plan)")
    _elementwise(sched, want, RTOL_FWD, "cfg 5 logits (scheduled kernels)")
    model.set_eval_precision("bf16")
    with torch.no_grad():
        g16 = model(data, slices)[0].cpu()
    model.set_eval_precision("fp32")
    # <= 1e-2 (rms, of the logits' scale); the element-wise maximum and the arg-max agreement are held to what
    # round 6 measured with margin (tools/exp/bf16_contract.py, profiles/r06_bf16_contract.txt: max 0.5-1.2e-2, agreement
    # >= 0.9976 over cfg 1 / 2 / 4 / 5 and two weight seeds; every flipped row is a near-tie of the fp32 logits, top-2 gap
    # <= 4e-3 of the scale).  The messages' bf16 rounding in front of the aggregation MFMA of the one-launch conv stack is not
    # the dominant term: the per-layer launches (fp32 messages into the mean) measure the same.
    rms = float(((g16 - want) ** 2).mean().sqrt() / (want ** 2).mean().sqrt())
    assert rms < 1e-2, rms
    assert float((g16 - want).abs().max()) <= 1.5e-2 * float(want.abs().max())
    assert float((g16.argmax(1) == want.argmax(1)).float().mean()) >= 0.99


def test_cfg5_full_size_train_step_matches_oracle():
    """configs[4] training step at full size: loss 1e-4 and every gradient tensor against the float64 CPU oracle at the
    tensor's own scale (train.py:263-284), fp32 storage; the bf16-storage step: loss within 2e-3."""
    yv = _yv()
    data, slices, optkw, _ = yv.config("5")
    opt = yv.Opt(**optkw)
    model = _model(yv, optkw, 55).train()
    out = model(data, slices)
    loss = yv.DetectionLoss(opt)(out, data)["loss"]
    loss.backward()
    grads = {n: p.grad.detach().clone() for n, p in model.named_parameters()}
    l32_logits, l32, g32 = _oracle_grads(optkw, 55, data, torch.float32)
    assert abs(float(loss.detach()) - float(l32)) <= RTOL_FWD * abs(float(l32))
    assert float((out[0].detach().cpu() - l32_logits).abs().max()) <= 5e-4 * float(l32_logits.abs().max())
    _, l64, g64 = _oracle_grads(optkw, 55, data, torch.float64)
    assert abs(float(loss.detach()) - float(l64)) <= RTOL_FWD * abs(float(l64))
    _grad_check_per_tensor(grads, g64, g32, 3e-3, "cfg 5")
    m16 = _model(yv, optkw, 55).train()
    m16.set_train_precision("bf16")
    data._yolat_stage = None
    l16 = yv.DetectionLoss(opt)(m16(data, slices), data)["loss"]
    assert abs(float(l16.detach()) - float(l64)) <= 2e-3 * abs(float(l64))
    # ... and every gradient tensor of the bf16-storage step against the float64 ORACLE at its own scale (the oracle's own
    # sensitivity to a bf16-sized input perturbation + a stated bf16 term: tests/test_gpu_bf16.py)
    from test_gpu_bf16 import bf16_grads_vs_fp64_oracle, _perturb_x
    l16.backward()
    g16 = {n: p.grad.detach().clone() for n, p in m16.named_parameters()}
    x0 = data.x
    data.x = _perturb_x(data)
    try:
        _, _, g64p = _oracle_grads(optkw, 55, data, torch.float64)
    finally:
        data.x = x0
    bf16_grads_vs_fp64_oracle(g16, g64, g64p, "cfg 5, bf16 storage")


def test_cfg5_train_step_is_deterministic_finite_and_block_diagonal():
    """BASELINE.json configs[4] (N=200k / E=1.2M / P=8000, n_blocks=4) training step in fp32: run-to-run bit
    identity, finite loss / gradients, and — with BatchNorm in eval mode, where rows do not interact — the CE of
    the whole graph equals the proposal-weighted mean of the CE of its two block-diagonal halves."""
    yv = _yv()
    data, slices, optkw, _ = yv.config("5")
    N, E, P = data.x.shape[0], data.edge.shape[0], data.bbox.shape[0]
    assert (N, E, P) == (200000, 1200000, 8000) and optkw["n_blocks"] == 4
    opt = yv.Opt(**optkw)
    outs = []
    for _ in range(2):
        model = _model(yv, optkw, 55)
        tr = yv.Trainer(model, opt, lr=2.5e-4, weight_decay=1e-5)
        data._yolat_stage = None
        loss = tr.step(data, slices)
        outs.append((loss.clone(), tr.flat.grad.clone(), tr.flat.param.clone()))
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
    assert torch.equal(outs[0][2], outs[1][2])
    assert bool(torch.isfinite(outs[0][0]).all()) and bool(torch.isfinite(outs[0][1]).all())
    assert float(outs[0][1].abs().max()) > 0
    # block-diagonal consistency through the training kernels' eval mode (forward_scheduled)
    model = _model(yv, optkw, 55).eval()
    crit = yv.DetectionLoss(opt)
    with torch.no_grad():
        whole = model.forward_scheduled(data, slices)
        l_whole = float(crit(whole, data)["loss"])
    half_p = P // 2
    n_half = int((data.bbox_idx < half_p).sum())
    e_mask = data.edge[:, 0] < n_half
    assert bool((e_mask == (data.edge[:, 1] < n_half)).all())      # no edge crosses the cut
    parts = []
    for lo_n, hi_n, lo_p, hi_p, em in ((0, n_half, 0, half_p, e_mask), (n_half, N, half_p, P, ~e_mask)):
        d = yv.Data(x=data.x[lo_n:hi_n].clone(), pos=data.pos[lo_n:hi_n].clone())
        d.edge = data.edge[em] - lo_n
        d.e_attr = data.e_attr[em].clone()
        d.bbox_idx = data.bbox_idx[lo_n:hi_n] - lo_p
        d.bbox = data.bbox[lo_p:hi_p].clone()
        d.labels = data.labels[lo_p:hi_p].clone()
        with torch.no_grad():
            parts.append((float(crit(model.forward_scheduled(d, None), d)["loss"]), hi_p - lo_p))
    l_parts = sum(l * n for l, n in parts) / P
    assert abs(l_whole - l_parts) <= 1e-5 * abs(l_whole)


def test_eval_after_training_sees_updated_batchnorm_and_weights():
    """The HIP kernels update running statistics and parameters through raw pointers (torch `_version` counters do
    not move): an eval forward after Trainer.step — eval plan, scheduled kernels, module-by-module — must use the
    new values every time (folded-coefficient caches are keyed on ops.weight_epoch()).  Reference for each step: a
    freshly constructed model (no caches) loaded with the trained model's state_dict — same kernels, same values,
    so the outputs must be bit-identical; and the first eval forward must agree with the CPU oracle."""
    yv = _yv()
    arrs, optkw = gu.graph_case("medium")
    opt = yv.Opt(**optkw)
    model = _model(yv, optkw, 9)
    ref = gu.fill_state_(orc.SparseCADGCN(orc.Opt(**optkw)), 9).eval()
    data = gu.to_data(arrs, yv.Data)
    tr = yv.Trainer(model, opt, lr=1e-2, weight_decay=1e-5)

    def evals(m):
        m.eval()
        with torch.no_grad():
            return (m(data, None)[0].clone(), m.forward_scheduled(data, None)[0].clone(),
                    m.forward_modular(data, None)[0].clone())

    with torch.no_grad():
        want = ref(gu.to_data(arrs, yv.Data), None)[0]
    prev = evals(model)                               # primes every cache with the initial values
    for got in prev:
        assert float((got.cpu() - want).abs().max()) <= RTOL_FWD * float(want.abs().max())
    for step in range(3):
        tr.step(data)
        cur = evals(model)
        fresh = yv.SparseCADGCN(opt).cuda()
        fresh.load_state_dict({k: v.clone() for k, v in model.state_dict().items()})
        exp = evals(fresh)
        for name, a, b, old in zip(("plan", "scheduled", "modular"), cur, exp, prev):
            assert torch.equal(a, b), "step %d: stale state on the %s path" % (step, name)
            assert not torch.equal(a, old), "step %d: %s output did not change after a training step" % (step, name)
        prev = cur


def test_stage_cache_sees_in_place_edits_of_the_batch():
    yv = _yv()
    arrs, optkw = gu.graph_case("small")
    model = _model(yv, optkw, 4).eval()
    data = gu.to_data(arrs, yv.Data)
    with torch.no_grad():
        a = model(data, None)[0].clone()
        data.x.mul_(0.5)                              # same storage, new contents
        b = model(data, None)[0].clone()
        data.e_attr.add_(0.25)
        c = model(data, None)[0].clone()
        fresh = gu.to_data(arrs, yv.Data)
        fresh.x.mul_(0.5)
        fresh.e_attr.add_(0.25)
        want = model(fresh, None)[0]
    assert not torch.equal(a, b) and not torch.equal(b, c)
    assert torch.equal(c, want)


def test_flat_adam_speaks_torch_adam_state_and_follows_steplr():
    """FlatAdam is a torch.optim.Optimizer: StepLR (train.py:214) drives its lr; torch.optim.Adam state_dicts
    (what reference checkpoints hold, ckpt_util.py:86-104) load into it and vice versa."""
    yv = _yv()
    arrs, optkw = gu.graph_case("small")
    opt = yv.Opt(**optkw)
    data = gu.to_data(arrs, yv.Data)
    # two steps with torch.optim.Adam on the HIP modules, then hand the state to a FlatAdam twin
    m_a = _model(yv, optkw, 6)
    m_a.train()
    adam = torch.optim.Adam(m_a.parameters(), lr=1e-3, weight_decay=1e-5)
    crit = yv.DetectionLoss(opt)
    for _ in range(2):
        adam.zero_grad()
        crit(m_a(data, None), data)["loss"].backward()
        adam.step()
    ckpt = {"epoch": 2, "state_dict": {k: v.clone() for k, v in m_a.state_dict().items()},
            "optimizer_state_dict": adam.state_dict()}
    m_b = _model(yv, optkw, 6)
    tr = yv.Trainer(m_b, opt, lr=1e-3, weight_decay=1e-5)
    yv.load_reference_checkpoint(m_b, ckpt, optimizer=tr.optimizer)
    assert tr.optimizer.step_count == 2
    # third step on both: same parameters afterwards
    adam.zero_grad()
    crit(m_a(data, None), data)["loss"].backward()
    adam.step()
    tr.step(data)
    for (n, pa), (_, pb) in zip(m_a.named_parameters(), m_b.named_parameters()):
        assert float((pa - pb).abs().max()) <= 2e-6 * max(float(pa.abs().max()), 1e-3), n
    # and back: FlatAdam's state_dict loads into torch.optim.Adam
    sd = tr.optimizer.state_dict()
    adam2 = torch.optim.Adam(m_b.parameters(), lr=1e-3, weight_decay=1e-5)
    adam2.load_state_dict(sd)
    assert int(adam2.state[next(iter(m_b.parameters()))]["step"]) == 3
    sched = torch.optim.lr_scheduler.StepLR(tr.optimizer, step_size=1, gamma=0.5)
    tr.step(data)
    sched.step()
    assert abs(tr.optimizer.lr - 5e-4) < 1e-12


def test_cross_entropy_rejects_out_of_range_labels():
    yv = _yv()
    arrs, optkw = gu.graph_case("small")
    opt = yv.Opt(**optkw)
    model = _model(yv, optkw, 2).train()
    data = gu.to_data(arrs, yv.Data)
    out = model(data, None)
    data.labels[0] = optkw["n_classes"]
    with pytest.raises(IndexError):
        yv.DetectionLoss(opt)(out, data)
    data.labels[0] = -100
    with pytest.raises(IndexError):
        yv.DetectionLoss(opt)(out, data)
    # device-resident labels cannot be checked without a sync: the kernel reads no out-of-bounds logit and
    # poisons the loss instead
    logits = torch.randn(300, 17, device="cuda")
    labels = torch.randint(0, 17, (300,), device="cuda")
    labels[7] = 17
    loss = torch.empty(1, device="cuda")
    yv.ops.softmax_ce(logits, labels, loss)
    assert bool(torch.isnan(loss).all())


def test_training_mode_dropout2d_matches_oracle_with_the_same_mask():
    """`--dropout p` puts nn.Dropout2d(p) behind prediction_cls.1 (torch_nn.py:67-68, arch:92).  The kernel draws an
    element-wise Bernoulli(1-p) mask (torch 1.7.1's behaviour for a 2-D input); with that very mask injected into the
    oracle, loss and gradients agree; eval mode ignores dropout; the mask follows torch.manual_seed."""
    yv = _yv()
    arrs, optkw = gu.graph_case("medium")
    optkw = dict(optkw, dropout=0.4)
    opt = yv.Opt(**optkw)
    data = gu.to_data(arrs, yv.Data)
    model = _model(yv, optkw, 8).train()
    assert any(m.__class__.__name__ == "Dropout2d" for m in model.prediction_cls[1])
    captured = {}
    real = yv.ops.dropout_fwd

    def spy(Y, scale, shift, relu, p, seed, Z):
        mask = real(Y, scale, shift, relu, p, seed, Z)
        want = Y * scale + shift if scale is not None else Y
        want = torch.relu(want) if relu else want
        assert torch.equal(Z, want * mask.view(Y.shape).float() * yv.ops.torch.tensor(1.0 / (1.0 - p), device="cuda").float()) \
            or float((Z - want * mask.view(Y.shape).float() / (1 - p)).abs().max()) <= 1e-6 * float(want.abs().max())
        captured["mask"], captured["p"] = mask.view(Y.shape).clone(), p
        return mask

    yv.ops.dropout_fwd = spy
    try:
        torch.manual_seed(123)
        out = model(data, None)
        loss = yv.DetectionLoss(opt)(out, data)["loss"]
        loss.backward()
    finally:
        yv.ops.dropout_fwd = real
    mask, p = captured["mask"], captured["p"]
    frac = float(mask.float().mean())
    assert abs(p - 0.4) < 1e-7 and abs(frac - 0.6) < 0.05, frac
    # the oracle with the same mask
    ref = gu.fill_state_(orc.SparseCADGCN(orc.Opt(**optkw)), 8).train()

    class FixedMask(torch.nn.Module):
        def forward(self, x):
            return x * mask.cpu().float() / (1 - p)

    seq = ref.prediction_cls[1]
    idx = [i for i, m in enumerate(seq) if isinstance(m, torch.nn.Dropout2d)]
    assert len(idx) == 1
    seq[idx[0]] = FixedMask()
    rlogits, rloss = _oracle_train(ref, optkw, data)
    assert abs(float(loss) - float(rloss)) <= RTOL_FWD * abs(float(rloss))
    _grad_check({n: q.grad for n, q in model.named_parameters()}, ref, 2e-3, "dropout")
    # same seed -> same mask; eval mode: no dropout, equals the p = 0 model
    m2 = _model(yv, optkw, 8).train()
    torch.manual_seed(123)
    l2 = yv.DetectionLoss(opt)(m2(data, None), data)["loss"]
    assert torch.equal(l2.detach(), loss.detach())
    fresh = _model(yv, optkw, 8).eval()                   # (the training forward above moved `model`'s running stats)
    plain = _model(yv, dict(optkw, dropout=0.0), 8).eval()
    with torch.no_grad():
        assert torch.equal(fresh(data, None)[0], plain(data, None)[0])


@pytest.mark.parametrize("seed", list(range(10)))
def test_random_shapes_eval_and_train_match_oracle(seed):
    """Seeded random batch shapes (1..5 graphs, 1..120 proposals, 2..40 nodes per proposal, sparse to dense edge sets,
    2..4 blocks, n_blocks_out 1..n_blocks) through the eval plan (bf16x6 fusion / classifier / node-side kernels, both
    edge-kernel families) and one training forward/backward, against the torch oracle: ragged tiles, small proposals,
    graphs whose edge count crosses the factorised-layer thresholds.  Gradients are compared outside the oracle's own
    ReLU sub-gradient band (_kink_band)."""
    yv = _yv()
    rng = np.random.default_rng(1000 + seed)
    n_graphs = int(rng.integers(1, 6))
    nb = int(rng.integers(2, 5))
    optkw = dict(n_classes=int(rng.integers(2, 23)), n_blocks=nb, n_blocks_out=int(rng.integers(1, nb + 1)))
    kw = dict(num_proposals=int(rng.integers(1, 121)), nodes_lo=int(rng.integers(2, 5)), nodes_hi=int(rng.integers(5, 41)),
              n_classes=optkw["n_classes"])
    if rng.random() < 0.5:
        kw["edge_factor"] = float(rng.uniform(0.3, 3.0))
    else:
        kw["edges_per_proposal"] = int(rng.integers(1, 200))
    data, slices = yv.synth_batch(n_graphs, 500 + seed, **kw)
    model = gu.fill_state_(yv.SparseCADGCN(yv.Opt(**optkw)), seed).cuda()
    ref = gu.fill_state_(orc.SparseCADGCN(orc.Opt(**optkw)), seed)
    model.eval(); ref.eval()
    with torch.no_grad():
        got = model(data, slices)[0].cpu()
        want = ref(data, None)[0]
    _elementwise(got, want, 1e-4, "logits seed %d %s %s" % (seed, optkw, kw))
    model.train(); ref.train().double()
    out = model(data, slices)
    loss = yv.DetectionLoss(yv.Opt(**optkw))(out, data)["loss"]
    loss.backward()
    _, rloss = _oracle_train(ref, optkw, data, torch.float64)
    assert abs(float(loss.detach()) - float(rloss)) <= 1e-4 * abs(float(rloss))
    band = _kink_band(lambda: gu.fill_state_(orc.SparseCADGCN(orc.Opt(**optkw)), seed), optkw, data)
    _grad_check({n: p.grad for n, p in model.named_parameters()}, ref, 5e-3, "seed %d" % seed, band=band)
