"""Host time of the two halves of a DeviceLoader hand-over at cfg 2: drawing a batch (DeviceLoader.__next__) and enqueueing the
forward on it."""
import os, sys, time, cProfile, pstats, io
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import yolat_vectorgraphicsrecognition_amd as yv
item, _, optkw, _ = yv.config("2")
for k in ("roots",):
    if hasattr(item, k):
        delattr(item, k)
model = yv.SparseCADGCN(yv.Opt(**optkw)).cuda().eval()
n = 2000
with torch.no_grad():
    b, sl = yv.collate_to_device([item], csr=True)
    for _ in range(20):
        model(b, sl)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        model(b, sl)
    t_enq = (time.perf_counter() - t0) / n
    torch.cuda.synchronize()
    t_tot = (time.perf_counter() - t0) / n
    print("forward on a resident prepared batch: host enqueue %.1f us / call, %.1f us / call with the GPU drained" % (t_enq * 1e6, t_tot * 1e6))
    loader = yv.DeviceLoader(([item] for _ in range(n + 50)), slots=3)
    for _ in range(50):
        next(loader)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n - 10):
        next(loader)
    print("DeviceLoader.__next__ alone: %.1f us / batch" % ((time.perf_counter() - t0) / (n - 10) * 1e6))
    loader.close()
    loader = yv.DeviceLoader(([item] for _ in range(600)), slots=3)
    pr = cProfile.Profile()
    pr.enable()
    for b2, s2 in loader:
        model(b2, s2)
    pr.disable()
    torch.cuda.synchronize()
    loader.close()
    st = io.StringIO()
    pstats.Stats(pr, stream=st).sort_stats("tottime").print_stats(18)
    print(st.getvalue()[:3500])
