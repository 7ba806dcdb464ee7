"""Soak: 300 training steps (cfg 4 batch), then 300 predict() calls and 300 H2D-inclusive forwards; memory and finiteness."""
import os, sys, time, torch
R = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import golden_util as gu
import yolat_vectorgraphicsrecognition_amd as yv
data, slices, optkw, n = yv.config("4")
opt = yv.Opt(**optkw)
model = gu.fill_state_(yv.SparseCADGCN(opt), 0).cuda()
tr = yv.Trainer(model, opt, lr=2.5e-4, weight_decay=1e-5)
for k in ("x", "edge", "e_attr", "bbox_idx", "bbox", "labels"):
    setattr(data, k, getattr(data, k).cuda())
losses = []
for i in range(300):
    data._yolat_stage = None
    l = tr.step(data, slices)
    if i % 50 == 0 or i == 299:
        losses.append(float(l)); print("train", i, losses[-1], torch.cuda.memory_reserved() >> 20, "MB", flush=True)
assert all(x == x and abs(x) < 1e6 for x in losses) and losses[-1] < losses[0]
model.eval()
cpu_item, sl, _, _ = yv.config("2")
for i in range(300):
    b, s2 = yv.collate_to_device([cpu_item])
    with torch.no_grad():
        out = model.__class__.forward  # noqa
    with torch.no_grad():
        o = model(b, s2)[0] if optkw["n_classes"] == 17 else None
    if i % 100 == 0:
        print("h2d fwd", i, torch.cuda.memory_reserved() >> 20, "MB", flush=True)
torch.cuda.synchronize(); print("soak ok")
