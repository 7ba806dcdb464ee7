"""Data-parallel training step with two ranks sharing one GPU (-m gpu): the real Trainer.step (HIP forward /
backward, flat-gradient all-reduce, fused Adam) over a gloo process group, checked against a single-process
computation of the averaged gradient.  (The multi-GPU run over RCCL is the driver's; this pins the logic.)"""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _batch(yv, rank):
    return yv.synth_batch(2, 40 + rank, num_proposals=30, nodes_lo=4, nodes_hi=20, edge_factor=1.5, augmented=True)


def _worker(rank, world, port, outdir):
    sys.path.insert(0, os.path.dirname(HERE))
    sys.path.insert(0, HERE)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import yolat_vectorgraphicsrecognition_amd as yv
    import golden_util as gu
    opt = yv.Opt()
    model = gu.fill_state_(yv.SparseCADGCN(opt), 60 + rank).cuda()     # ranks start DIFFERENT: broadcast must fix it
    tr = yv.Trainer(model, opt, lr=1e-3, weight_decay=1e-5)
    data, slices = _batch(yv, rank)
    losses = [float(tr.step(data, slices)) for _ in range(2)]
    torch.cuda.synchronize()
    np.savez(os.path.join(outdir, "rank%d.npz" % rank), param=tr.flat.param.cpu().numpy(), losses=np.array(losses))
    dist.destroy_process_group()


@pytest.mark.parametrize("buckets", ["2", "1"])
def test_dp_train_step_two_ranks_matches_single_process_average(tmp_path, buckets):
    """(buckets: YOLAT_DP_BUCKETS — two async all-reduces, the head bucket issued between the phases of the one-call
    training step, or ONE all-reduce of the whole flat gradient after the backward; the same result)"""
    sys.path.insert(0, os.path.dirname(HERE))
    import yolat_vectorgraphicsrecognition_amd as yv
    import golden_util as gu
    port = _free_port()
    old_env = os.environ.get("YOLAT_DP_BUCKETS")
    os.environ["YOLAT_DP_BUCKETS"] = buckets
    try:
        mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    finally:
        if old_env is None:
            os.environ.pop("YOLAT_DP_BUCKETS", None)
        else:
            os.environ["YOLAT_DP_BUCKETS"] = old_env
    r0, r1 = np.load(tmp_path / "rank0.npz"), np.load(tmp_path / "rank1.npz")
    np.testing.assert_array_equal(r0["param"], r1["param"])           # replicas stay bit-identical
    # single process: same start (rank 0's weights), gradient = mean of the two ranks' gradients, same Adam
    opt = yv.Opt()
    model = gu.fill_state_(yv.SparseCADGCN(opt), 60).cuda()
    flat = yv.FlatParams(model)
    adam = yv.FlatAdam(flat, lr=1e-3, weight_decay=1e-5)
    crit = yv.DetectionLoss(opt)
    batches = [_batch(yv, r) for r in range(2)]
    losses = []
    for _ in range(2):
        model.train()
        gsum = torch.zeros_like(flat.grad)
        bufs = None
        step_losses = []
        for r, (data, slices) in enumerate(batches):
            if r == 1:      # BatchNorm running stats are per replica: rank 0's are the ones compared below
                saved = {n: b.clone() for n, b in model.named_buffers()}
            adam.zero_grad()
            loss = crit(model(data, slices), data)["loss"]
            loss.backward()
            gsum += flat.grad
            step_losses.append(float(loss))
            if r == 1:
                for n, b in model.named_buffers():
                    b.copy_(saved[n])
        flat.grad.copy_(gsum)
        adam.step(grad_scale=0.5)
        losses.append(step_losses)
    np.testing.assert_allclose(r0["losses"], [l[0] for l in losses], rtol=2e-6)
    np.testing.assert_allclose(r1["losses"], [l[1] for l in losses], rtol=2e-6)
    ref = flat.param.cpu().numpy()
    scale = np.abs(ref).max()
    assert np.abs(r0["param"] - ref).max() <= 2e-6 * scale


def _nccl_worker(rank, world, port, outdir):
    sys.path.insert(0, os.path.dirname(HERE))
    sys.path.insert(0, HERE)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", 0))
    import yolat_vectorgraphicsrecognition_amd as yv
    import golden_util as gu
    opt = yv.Opt()
    res = {}
    from yolat_vectorgraphicsrecognition_amd import engine
    # "big": large enough (N ~ 90 k) that the side stream's weight-gradient kernels of the fusion block are still running
    # when the host has already enqueued what follows the backward's head — the situation the join exists for
    batches = {"small": _batch(yv, 0),
               "big": yv.synth_batch(4, 47, num_proposals=1000, nodes_lo=4, nodes_hi=40, edge_factor=1.5, augmented=True)}
    for mode in ("exchange", "local", "premul", "premul_fault", "local_big"):
        model = gu.fill_state_(yv.SparseCADGCN(opt), 61).cuda()
        # premul: the buckets are doubled before their all-reduce, Adam halves them (exact) — the exchange is no longer
        # an identity; premul_fault: the same with the head bucket's exchange issued before its gradients exist
        import contextlib
        fault = engine.fault_early_head_exchange() if mode == "premul_fault" else contextlib.nullcontext()
        with fault:
            tr = yv.Trainer(model, opt, lr=1e-3, weight_decay=1e-5, force_exchange=(not mode.startswith("local")),
                            exchange_premul=(2.0 if mode.startswith("premul") else None))
            assert tr.flat.conv_end > 0                       # the two-bucket branch is the one that runs
            fired = []
            if mode == "exchange":
                real = dist.all_reduce

                def spy(t, *a, **k):
                    fired.append((int(t.numel()), bool(k.get("async_op", False))))
                    return real(t, *a, **k)
                dist.all_reduce = spy
            data, slices = batches["big" if mode.startswith("premul") or mode == "local_big" else "small"]
            losses = []
            for _ in range(3):
                data._yolat_stage = None
                losses.append(float(tr.step(data, slices)))
            torch.cuda.synchronize()
            if mode == "exchange":
                dist.all_reduce = real
                # per step: bucket 1 (fusion + classifier) from inside the backward, bucket 2 (conv layers) after it
                assert len(fired) == 6 and all(a for _, a in fired), fired
                assert fired[0][0] == tr.flat.numel - tr.flat.conv_end and fired[1][0] == tr.flat.conv_end, fired
        res[mode] = (tr.flat.param.cpu().numpy().copy(), np.array(losses))
    np.savez(os.path.join(outdir, "nccl.npz"), p_ex=res["exchange"][0], p_lo=res["local"][0],
             l_ex=res["exchange"][1], l_lo=res["local"][1], p_pm=res["premul"][0], l_pm=res["premul"][1],
             p_pf=res["premul_fault"][0], p_lb=res["local_big"][0], l_lb=res["local_big"][1],
             backend=np.array(dist.get_backend()))
    dist.destroy_process_group()


def test_dp_exchange_over_rccl_in_a_one_rank_group_equals_the_local_step(tmp_path):
    """The exchange branch of Trainer.step (async all-reduce fired from inside the backward, second bucket, wait, Adam
    with the 1/world scale) over the `nccl` backend = librccl, in a process group of ONE rank on the one GPU: the
    collective runs on RCCL's own stream, ordered against the HIP kernels' stream by events — the ordering gloo's
    host-staged path never exercises.  (1) SUM over one rank is the identity: parameters and losses bit-identical to the
    step without the exchange.  (2) The identity cannot see an exchange issued too early, so the same step runs with
    Trainer(exchange_premul=2): every bucket is doubled on the compute stream right before its all-reduce and Adam halves
    it — exact in fp32, still bit-identical, but only if every gradient was in its bucket at that point.  (3) The check
    can fail: with the head bucket's exchange issued before its gradients exist (engine._FAULT_EARLY_HEAD_EXCHANGE: the
    program-order form of a missing wait; a removed stream join alone does not reproduce on demand, see engine.py) the
    gradients land after the doubling and the parameters differ."""
    port = _free_port()
    mp.spawn(_nccl_worker, args=(1, port, str(tmp_path)), nprocs=1, join=True)
    r = np.load(tmp_path / "nccl.npz")
    assert str(r["backend"]) == "nccl"
    np.testing.assert_array_equal(r["l_ex"], r["l_lo"])
    np.testing.assert_array_equal(r["p_ex"], r["p_lo"])
    # a NON-identity exchange (buckets doubled on the compute stream in front of the all-reduce, halved by Adam: exact in
    # fp32): still bit-identical — every gradient was in its bucket when the exchange was issued ...
    np.testing.assert_array_equal(r["l_pm"], r["l_lb"])
    np.testing.assert_array_equal(r["p_pm"], r["p_lb"])
    # ... and the check can FAIL: with the head bucket's exchange issued at the start of the backward
    # (engine._FAULT_EARLY_HEAD_EXCHANGE) its gradients land after the doubling and the parameters differ
    assert not np.array_equal(r["p_pf"], r["p_lb"]), "the exchange-ordering check did not see an exchange issued too early"


def test_bench_line_with_two_ranks_launched_the_way_the_driver_launches_them():
    """bench.py under `python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 ... bench.py --gpus 2` — the driver's
    N > 1 form — with both ranks on the one GPU of this box over gloo (YOLAT_BENCH_DEVICE / YOLAT_BENCH_BACKEND exist for
    exactly this): rank 0 prints ONE JSON line, `value` is the aggregate over the ranks (graph-id sharding, weak scaling:
    every rank steps through its own graph), the timed region is bracketed by the group barrier.  The RCCL run with one
    GPU per rank is the driver's; this pins the control flow (rendezvous on 127.0.0.1, per-rank seeds, MAX over ranks)."""
    import json
    import subprocess
    root = os.path.dirname(HERE)
    env = dict(os.environ, YOLAT_BENCH_DEVICE="0", YOLAT_BENCH_BACKEND="gloo")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                        "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), os.path.join(root, "bench.py"),
                        "--gpus", "2", "--steps", "20", "--warmup", "5", "--no-extras", "--no-cpu-baseline"],
                       env=env, cwd=root, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-1000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 20 and d["warmup"] == 5 and d["scaling"] == "weak"
    assert d["metric"] == "graphs_per_sec" and d["value"] > 0 and d["higher_is_better"] is True
    # aggregate = graphs of both ranks per step / the slowest rank's time
    assert abs(d["value"] - 2.0 / (d["ms_per_step"] * 1e-3)) <= 1e-6 * d["value"]
    # the N > 1 line proves that N ranks took part: a SUM all-reduce of ones, every rank's own ms / step, who they were
    pr = d["participation"]
    assert pr["ranks_seen"] == 2 and pr["world_size"] == 2 and pr["backend"] == "gloo"
    assert len(pr["per_rank_ms_per_step"]) == 2 and all(t > 0 for t in pr["per_rank_ms_per_step"])
    assert max(pr["per_rank_ms_per_step"]) <= d["ms_per_step"] * (1 + 1e-6) + 1e-5       # (the record rounds to 1e-5 ms)
    assert sorted(r_["rank"] for r_ in pr["ranks"]) == [0, 1]


def test_bench_train_dp_record_with_two_ranks_carries_both_exchange_layouts():
    """the N > 1 line's `train_dp` sub-record (cfg-4 training step with the gradient exchange, two ranks over gloo on this one
    GPU): the step goes through yolat_train_step with the exchange between its phases, the one-bucket layout
    (YOLAT_DP_BUCKETS=1) is timed beside the two-bucket one, and the record proves its ranks."""
    import json
    import subprocess
    root = os.path.dirname(HERE)
    env = dict(os.environ, YOLAT_BENCH_DEVICE="0", YOLAT_BENCH_BACKEND="gloo")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT", "YOLAT_DP_BUCKETS"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                        "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), os.path.join(root, "bench.py"),
                        "--gpus", "2", "--steps", "20", "--warmup", "5", "--no-cpu-baseline", "--no-roofline"],
                       env=env, cwd=root, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-1000:]
    td = json.loads(lines[0])["train_dp"]
    assert td["ms_per_step"] > 0 and td["ms_per_step_one_bucket"] > 0 and td["ms_per_step_without_exchange"] > 0
    assert td["steps_through_yolat_train_step"] > 0
    assert td["participation"]["ranks_seen"] == 2 and td["participation"]["backend"] == "gloo"
