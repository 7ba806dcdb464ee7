"""CPU tests of the multi-GPU path (-m "not gpu"): world_size-2 `gloo` processes exercise exactly the
collective code the RCCL run uses (trainer.broadcast_parameters / allreduce_mean_ on the flat
buffers) and the graph-id sharding."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


class _Flat(object):
    def __init__(self, rank):
        g = torch.Generator().manual_seed(100 + rank)
        self.param = torch.randn(1000, generator=g)
        self.grad = torch.randn(1000, generator=g)


class _Model(torch.nn.Module):
    def __init__(self, rank):
        super().__init__()
        self.bn = torch.nn.BatchNorm1d(8)
        with torch.no_grad():
            self.bn.running_mean.fill_(float(rank + 1))


def _worker(rank, world, port, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from yolat_vectorgraphicsrecognition_amd import trainer
    flat, model = _Flat(rank), _Model(rank)
    g_local = flat.grad.clone()
    trainer.broadcast_parameters(flat, model, src=0)
    scale = trainer.allreduce_mean_(flat.grad)
    gathered = [torch.zeros(1000) for _ in range(world)]
    dist.all_gather(gathered, g_local)
    ok = True
    ok &= abs(scale - 1.0 / world) < 1e-12
    ok &= torch.allclose(flat.grad, sum(gathered))
    ok &= torch.equal(flat.param, _Flat(0).param)                     # everyone holds rank 0's parameters
    ok &= float(model.bn.running_mean[0]) == 1.0                      # and rank 0's BatchNorm buffers
    # the two exchange layouts of Trainer.step (YOLAT_DP_BUCKETS): two asynchronous SUM all-reduces of the flat gradient's
    # [conv_end:] and [:conv_end] ranges (head bucket first, as the one-call step issues them between its phases) give what
    # ONE all-reduce of the whole buffer gives
    whole = g_local.clone()
    dist.all_reduce(whole, op=dist.ReduceOp.SUM)
    two = g_local.clone()
    conv_end = 137
    handles = [dist.all_reduce(two[conv_end:], op=dist.ReduceOp.SUM, async_op=True),
               dist.all_reduce(two[:conv_end], op=dist.ReduceOp.SUM, async_op=True)]
    for h in handles:
        h.wait()
    ok &= torch.equal(two, whole)
    os.environ["YOLAT_DP_BUCKETS"] = "1"
    ok &= trainer.dp_buckets() == 1
    os.environ.pop("YOLAT_DP_BUCKETS")
    ok &= trainer.dp_buckets() == 2
    ids = trainer.shard_graph_ids(11, rank, world)
    all_ids = [None] * world
    dist.all_gather_object(all_ids, ids)
    flat_ids = sorted(i for part in all_ids for i in part)
    ok &= flat_ids == list(range(11))
    out[rank] = bool(ok)
    dist.destroy_process_group()


def test_flat_gradient_allreduce_and_broadcast_world2():
    world = 2
    port = _free_port()
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(world, port, out), nprocs=world, join=True)
    assert dict(out) == {0: True, 1: True}


def test_single_process_allreduce_is_identity():
    from yolat_vectorgraphicsrecognition_amd import trainer
    g = torch.arange(5.0)
    assert trainer.allreduce_mean_(g) == 1.0
    assert torch.equal(g, torch.arange(5.0))
    assert trainer.shard_graph_ids(5, 0, 1) == [0, 1, 2, 3, 4]


def _proof_worker(rank, world, port, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(os.path.dirname(os.path.dirname(
        os.path.abspath(__file__))), "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    out[rank] = bench.participation_record(1.5 + rank, world)
    dist.destroy_process_group()


def test_bench_participation_record_world2():
    """bench.py's N > 1 proof (ranks_seen, per-rank ms / step, rank identities) over a two-rank gloo group"""
    world, port = 2, _free_port()
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_proof_worker, args=(world, port, out), nprocs=world, join=True)
    for rank in range(world):
        pr = out[rank]
        assert pr["ranks_seen"] == 2 and pr["world_size"] == 2 and pr["backend"] == "gloo" and pr["rccl_version"] is None
        assert pr["per_rank_ms_per_step"] == [1.5, 2.5]
        assert [r["rank"] for r in pr["ranks"]] == [0, 1]
