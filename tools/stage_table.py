"""Print the per-stage HIP-event table of the eval plan (yolat_profile_*) for a config."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
import yolat_vectorgraphicsrecognition_amd as yv

cfg = sys.argv[1] if len(sys.argv) > 1 else "2"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 100
data, slices, optkw, n_graphs = yv.config(cfg)
model = yv.SparseCADGCN(yv.Opt(**optkw)).cuda().eval()
bench.to_device(data)


def step():
    with torch.no_grad():
        data._yolat_stage = None
        model(data, slices)


for _ in range(20):
    step()
t = bench.plan_profile(step, n)
tot = 0.0
for k, v in sorted(t.items(), key=lambda kv: -kv[1]["ms_total"]):
    us = v["ms_total"] / n * 1e3
    tot += us
    print("%-58s calls/step %4.1f  %8.2f us/step  %8.2f us/call  %7.2f TF/s %8.1f GB/s" % (
        k, v["calls"] / n, us, v["ms_avg"] * 1e3, v["flops"] / (v["ms_avg"] * 1e-3) / 1e12 if v["ms_avg"] else 0,
        v["bytes"] / (v["ms_avg"] * 1e-3) / 1e9 if v["ms_avg"] else 0))
print("sum %.1f us" % tot)
