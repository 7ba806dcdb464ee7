"""Where does the gradient exchange's cost go at world size 1?  (`train_dp_single_rank_nccl` 2.955 ms with
the exchange against 2.741 ms without, for a collective that takes 0.012 ms alone.)

Same model instance, exchange toggled per leg, 50 timed steps per leg (3 legs interleaved twice), per step:
  * wall time of the step (host clock, stream synchronised at both ends),
  * GPU time of the step (HIP events on the compute stream),
  * host time spent inside dist.all_reduce (enqueue of the collective on RCCL's stream + its event plumbing), inside the
    bucket pre-multiplies and inside work.wait().
usage: python tools/exp/dp1_breakdown.py [steps=50]"""
import os
import sys
import time

import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
import yolat_vectorgraphicsrecognition_amd as yv
import golden_util as gu

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 50
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", str(29500 + os.getpid() % 2000))
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
data, slices, optkw, n_graphs = yv.config("4")
opt = yv.Opt(**optkw)
for k, v in list(data.__dict__.items()):
    if torch.is_tensor(v):
        data.__dict__[k] = v.cuda()
model = gu.fill_state_(yv.SparseCADGCN(opt), 0).cuda()
tr = yv.Trainer(model, opt, lr=2.5e-4, weight_decay=1e-5, force_exchange=True)

host = {"all_reduce": 0.0, "wait": 0.0, "calls": 0}
real_ar = dist.all_reduce


class _Work(object):
    def __init__(self, w):
        self.w = w

    def wait(self):
        t0 = time.perf_counter()
        r = self.w.wait()
        host["wait"] += time.perf_counter() - t0
        return r


def timed_ar(t, *a, **k):
    t0 = time.perf_counter()
    w = real_ar(t, *a, **k)
    host["all_reduce"] += time.perf_counter() - t0
    host["calls"] += 1
    return _Work(w) if w is not None else None


dist.all_reduce = timed_ar


def leg(name, exchange, premul):
    tr.exchange_gradients = exchange
    tr.exchange_premul = premul
    for _ in range(5):
        data._yolat_stage = None
        tr.step(data, slices)
    torch.cuda.synchronize()
    for k in host:
        host[k] = 0
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    gpu = 0.0
    t0 = time.perf_counter()
    for _ in range(steps):
        data._yolat_stage = None
        ev0.record()
        tr.step(data, slices)
        ev1.record()
        ev1.synchronize()
        gpu += ev0.elapsed_time(ev1)
    wall_sync = (time.perf_counter() - t0) / steps * 1e3
    torch.cuda.synchronize()
    # free-running (what bench.py times): no per-step synchronisation
    t0 = time.perf_counter()
    for _ in range(steps):
        data._yolat_stage = None
        tr.step(data, slices)
    torch.cuda.synchronize()
    wall_free = (time.perf_counter() - t0) / steps * 1e3
    return {"leg": name, "ms_free_running": wall_free, "ms_synced_each_step": wall_sync, "gpu_ms_event": gpu / steps,
            "host_ms_in_all_reduce": host["all_reduce"] / (2 * steps) * 1e3, "host_ms_in_wait": host["wait"] / (2 * steps) * 1e3,
            "all_reduce_calls_per_step": host["calls"] / (2.0 * steps)}


rows = []
for rep in range(2):
    rows.append(leg("local", False, None))
    rows.append(leg("exchange", True, None))
    rows.append(leg("exchange+premul2", True, 2.0))
print("%-18s %10s %10s %10s %14s %10s %6s" % ("leg", "free ms", "synced ms", "gpu ms", "host in a-r ms", "wait ms", "calls"))
for r in rows:
    print("%-18s %10.4f %10.4f %10.4f %14.4f %10.4f %6.1f" % (r["leg"], r["ms_free_running"], r["ms_synced_each_step"], r["gpu_ms_event"],
                                                             r["host_ms_in_all_reduce"], r["host_ms_in_wait"], r["all_reduce_calls_per_step"]))
# the collective alone
g = tr.flat.grad
for _ in range(5):
    real_ar(g)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(50):
    real_ar(g)
torch.cuda.synchronize()
print("all_reduce of the flat gradient alone, back to back: %.4f ms per call (host-paced)" % ((time.perf_counter() - t0) / 50 * 1e3))
dist.destroy_process_group()
