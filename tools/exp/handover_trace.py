"""one leg of the hand-over comparison for a kernel trace: python tools/exp/handover_trace.py resident|loader [n]"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import yolat_vectorgraphicsrecognition_amd as yv
mode = sys.argv[1]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 400
item, _, optkw, _ = yv.config("2")
for k in ("roots",):
    if hasattr(item, k): delattr(item, k)
model = yv.SparseCADGCN(yv.Opt(**optkw)).cuda().eval()
with torch.no_grad():
    b, sl = yv.collate_to_device([item], csr=True)
    for _ in range(20):
        model(b, sl)
    torch.cuda.synchronize()
    t = time.perf_counter()
    if mode == "resident":
        for _ in range(n):
            model(b, sl)
    else:
        ld = yv.DeviceLoader(([item] for _ in range(n)), slots=3)
        for bb, ss in ld:
            model(bb, ss)
    torch.cuda.synchronize()
    print(mode, "%.1f us per batch" % ((time.perf_counter() - t) / n * 1e6))
