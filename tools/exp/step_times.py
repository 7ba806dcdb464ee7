"""Per-step wall times of the cfg-3 training step / cfg-2 forward right after process start (is the first ~100 ms slow?)."""
import os, sys, time, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo")); sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "tests"))
import golden_util as gu
import yolat_vectorgraphicsrecognition_amd as yv
mode = sys.argv[1] if len(sys.argv) > 1 else "train"
cfg = "3" if mode == "train" else "2"
data, slices, optkw, n = yv.config(cfg)
opt = yv.Opt(**optkw)
model = gu.fill_state_(yv.SparseCADGCN(opt), 0).cuda()
for k in ("x", "edge", "e_attr", "bbox_idx", "bbox", "labels"):
    setattr(data, k, getattr(data, k).cuda())
if mode == "train":
    tr = yv.Trainer(model, opt, lr=2.5e-4, weight_decay=1e-5)
    def step():
        data._yolat_stage = None
        tr.step(data, slices)
else:
    model.eval()
    def step():
        data._yolat_stage = None
        with torch.no_grad():
            model(data, slices)
ts = []
t_start = time.perf_counter()
for i in range(80 if mode == "train" else 2000):
    t0 = time.perf_counter(); step(); torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
blk = 10 if mode == "train" else 250
print(mode, "per-block mean ms:", [round(sum(ts[i:i + blk]) / blk, 4) for i in range(0, len(ts), blk)], "mem_reserved_MB", torch.cuda.memory_reserved() >> 20)
