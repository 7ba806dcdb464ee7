"""Host time of the two calls of the hand-over loop (perf_counter around each, no synchronisation inside the loop)."""
import os, sys, time, torch
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tests"))
import yolat_vectorgraphicsrecognition_amd as yv
import golden_util as gu
cpu_item, _, optkw, _ = yv.config("2")
if hasattr(cpu_item, "roots"):
    delattr(cpu_item, "roots")
model = gu.fill_state_(yv.SparseCADGCN(yv.Opt(**optkw)), 0).cuda().eval()
pc = time.perf_counter
for rep in range(2):
    tc = tf = 0.0
    torch.cuda.synchronize()
    t0 = pc()
    for _ in range(300):
        a = pc()
        b, sl = yv.collate_to_device([cpu_item], csr=True)
        c = pc()
        with torch.no_grad():
            model(b, sl)
        d = pc()
        tc += c - a; tf += d - c
    t1 = pc()
    torch.cuda.synchronize()
    t2 = pc()
    print("host: collate %.1f us, forward call %.1f us, loop %.1f us/iter; drain after loop %.1f us; total %.1f us/iter"
          % (tc / 300 * 1e6, tf / 300 * 1e6, (t1 - t0) / 300 * 1e6, (t2 - t1) * 1e6, (t2 - t0) / 300 * 1e6))
