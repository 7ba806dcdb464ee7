import os, sys, time, torch
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import yolat_vectorgraphicsrecognition_amd as yv, golden_util as gu, bench
import cProfile, pstats
data, slices, optkw, _ = yv.config("2")
model = gu.fill_state_(yv.SparseCADGCN(yv.Opt(**optkw)), 0).cuda().eval()
bench.to_device(data)
streams = [torch.cuda.Stream() for _ in range(8)]
def step(i):
    data._yolat_stage = None
    with torch.cuda.stream(streams[i % 8]), torch.no_grad():
        return model(data, slices)[0]
for i in range(50): step(i)
torch.cuda.synchronize()
n = 800
t0 = time.perf_counter()
for i in range(n): step(i)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print("host enqueue %.1f us/step, total %.1f us/step" % ((t1 - t0) / n * 1e6, (t2 - t0) / n * 1e6))
pr = cProfile.Profile(); pr.enable()
for i in range(400): step(i)
pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("cumulative").print_stats(14)
