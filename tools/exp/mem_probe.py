import os, sys, gc, torch
R = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import golden_util as gu
import yolat_vectorgraphicsrecognition_amd as yv
cfg = sys.argv[1] if len(sys.argv) > 1 else "4"
data, slices, optkw, n = yv.config(cfg)
opt = yv.Opt(**optkw)
model = gu.fill_state_(yv.SparseCADGCN(opt), 0).cuda()
tr = yv.Trainer(model, opt, lr=2.5e-4, weight_decay=1e-5)
for k in ("x", "edge", "e_attr", "bbox_idx", "bbox", "labels"):
    setattr(data, k, getattr(data, k).cuda())
for i in range(40):
    data._yolat_stage = None
    l = tr.step(data, slices)
    torch.cuda.synchronize()
    print(i, "alloc %d MB reserved %d MB gc %s" % (torch.cuda.memory_allocated() >> 20, torch.cuda.memory_reserved() >> 20, gc.get_count()), flush=True)
    if i == 29:
        n = gc.collect(); torch.cuda.synchronize()
        print("gc.collect() freed %d objects -> alloc %d MB" % (n, torch.cuda.memory_allocated() >> 20), flush=True)
