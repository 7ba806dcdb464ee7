import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import yolat_vectorgraphicsrecognition_amd as yv
item, _, optkw, _ = yv.config("2")
for k in ("roots",):
    if hasattr(item, k): delattr(item, k)
model = yv.SparseCADGCN(yv.Opt(**optkw)).cuda().eval()
for csr in (True, False):
    with torch.no_grad():
        n = 1000
        ld = yv.DeviceLoader(([item] for _ in range(n + 20)), slots=3, csr=csr)
        for _ in range(20): next(ld)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(n - 5): next(ld)
        print("csr=%s next() alone: %.1f us per batch" % (csr, (time.perf_counter() - t0) / (n - 5) * 1e6))
        ld.close()
        ld = yv.DeviceLoader(([item] for _ in range(n)), slots=3, csr=csr)
        tn = tf = 0.0; it = iter(ld); t_all = time.perf_counter()
        while True:
            t0 = time.perf_counter()
            try: b, s = next(it)
            except StopIteration: break
            t1 = time.perf_counter(); model(b, s); t2 = time.perf_counter()
            tn += t1 - t0; tf += t2 - t1
        torch.cuda.synchronize()
        print("csr=%s loop: next %.1f us, forward enqueue %.1f us, total %.1f us per batch" % (csr, tn / n * 1e6, tf / n * 1e6, (time.perf_counter() - t_all) / n * 1e6))
        ld.close()
