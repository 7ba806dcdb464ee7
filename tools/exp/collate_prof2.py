"""Where the host time of collate_to_device(csr=True) + one forward goes (cfg 2, one item)."""
import os, sys, time, cProfile, pstats
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
import torch
import golden_util as gu
import yolat_vectorgraphicsrecognition_amd as yv
item, _, optkw, _ = yv.config("2")
model = gu.fill_state_(yv.SparseCADGCN(yv.Opt(**optkw)), 0).cuda().eval()
for csr in (False, True):
    for _ in range(10):
        b, sl = yv.collate_to_device([item], csr=csr)
        with torch.no_grad():
            model(b, sl)
    torch.cuda.synchronize()
    n = 300
    t0 = time.perf_counter()
    for _ in range(n):
        b, sl = yv.collate_to_device([item], csr=csr)
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    for _ in range(n):
        with torch.no_grad():
            model(b, sl)
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    for _ in range(n):
        b, sl = yv.collate_to_device([item], csr=csr)
        with torch.no_grad():
            model(b, sl)
    torch.cuda.synchronize()
    t3 = time.perf_counter()
    print("csr=%s: collate alone %.1f us, forward alone %.1f us, both %.1f us" % (csr, (t1 - t0) / n * 1e6, (t2 - t1) / n * 1e6,
                                                                             (t3 - t2) / n * 1e6))
pr = cProfile.Profile()
pr.enable()
for _ in range(300):
    b, sl = yv.collate_to_device([item], csr=True)
    with torch.no_grad():
        model(b, sl)
torch.cuda.synchronize()
pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(22)
