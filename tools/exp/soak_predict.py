"""300 predict() calls (two-pass inference with device-side sub-batch extraction) + NMS: memory level and result stability."""
import os, sys, gc, torch
R = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import golden_util as gu
import yolat_vectorgraphicsrecognition_amd as yv
data, slices = gu.predict_case(yv.synth_batch)
model = gu.fill_state_(yv.SparseCADGCN(yv.Opt(**gu.PREDICT_OPT)), 7).cuda().eval()
ref = None
gc.collect(); gc.disable()
for i in range(300):
    with torch.no_grad():
        cls, bbox, _, sb, sib, _ = model.predict(data, slices)
    if ref is None:
        ref = cls.clone()
    if i % 50 == 0 or i == 299:
        torch.cuda.synchronize()
        print(i, "alloc", torch.cuda.memory_allocated() >> 10, "KB reserved", torch.cuda.memory_reserved() >> 20, "MB equal", bool(torch.equal(cls, ref)), flush=True)
print("predict soak ok")
