import os, sys, time, torch
sys.path.insert(0, "/root/repo")
import yolat_vectorgraphicsrecognition_amd as yv
item, _, optkw, _ = yv.config("2")
for k in ("roots",):
    if hasattr(item, k): delattr(item, k)
model = yv.SparseCADGCN(yv.Opt(**optkw)).cuda().eval()
with torch.no_grad():
    b, sl = yv.collate_to_device([item], csr=True)
    def run(n, ev=False, copy=False):
        evs = [torch.cuda.Event() for _ in range(4)]
        cs = torch.cuda.Stream()
        src = torch.empty(1400000, dtype=torch.uint8).pin_memory()
        dst = [torch.empty(1400000, dtype=torch.uint8, device="cuda") for _ in range(3)]
        for i in range(n):
            if copy:
                with torch.cuda.stream(cs):
                    dst[i % 3].copy_(src, non_blocking=True)
            model(b, sl)
            if ev:
                evs[i % 4].record()
        torch.cuda.synchronize()
    for name, kw in (("plain", {}), ("event record per forward", {"ev": True}), ("1.4 MB H2D per forward on a side stream", {"copy": True}), ("both", {"ev": True, "copy": True})):
        run(50, **kw)
        r = []
        for _ in range(3):
            t = time.perf_counter(); run(400, **kw); r.append((time.perf_counter() - t) / 400 * 1e6)
        print("%-45s %.1f us per forward" % (name, sorted(r)[1]))
