"""Host enqueue time vs GPU time of a training step (cfg given): is the step host-bound?"""
import os, sys, time
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import torch
import yolat_vectorgraphicsrecognition_amd as yv, golden_util as gu, bench
cfg = sys.argv[1] if len(sys.argv) > 1 else "3"
data, slices, optkw, _ = yv.config(cfg)
opt = yv.Opt(**optkw)
model = gu.fill_state_(yv.SparseCADGCN(opt), 0).cuda()
bench.to_device(data)
tr = yv.Trainer(model, opt, lr=2.5e-4, weight_decay=1e-5)
def step():
    data._yolat_stage = None
    return tr.step(data, slices)
for _ in range(5): step()
torch.cuda.synchronize()
n = 20
t0 = time.perf_counter()
for _ in range(n): step()
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print("cfg %s: host enqueue %.3f ms/step, wall %.3f ms/step" % (cfg, (t1 - t0) / n * 1e3, (t2 - t0) / n * 1e3))
import cProfile, pstats
pr = cProfile.Profile(); pr.enable()
for _ in range(10): step()
pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("tottime").print_stats(18)
