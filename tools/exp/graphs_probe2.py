import os, sys, torch
R = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import golden_util as gu
import yolat_vectorgraphicsrecognition_amd as yv
data, slices, optkw, n = yv.config("2")
model = gu.fill_state_(yv.SparseCADGCN(yv.Opt(**optkw)), 0).cuda().eval()
for k in ("x", "edge", "e_attr", "bbox_idx", "bbox", "labels"):
    setattr(data, k, getattr(data, k).cuda())
model.use_hip_graphs(True)
nsteps = int(sys.argv[1]) if len(sys.argv) > 1 else 500
ref = None
for i in range(nsteps):
    data._yolat_stage = None
    with torch.no_grad():
        out = model(data, slices)[0]
    if os.environ.get('SYNC_EVERY') and i % int(os.environ['SYNC_EVERY']) == 0:
        torch.cuda.synchronize()
    if i % 10 == 0:
        torch.cuda.synchronize()
        plan = next(iter(model.__dict__.get("_yolat_plans", {}).values()), None)
        ng = len(plan._graphs) if plan is not None else -1
        if ref is None: ref = out.clone()
        if i == 50 or i == 0:
            def rng(name, t):
                print("  %-10s %#x .. %#x (%d B)" % (name, t.data_ptr(), t.data_ptr() + t.numel() * t.element_size(), t.numel() * t.element_size()), flush=True)
            rng("ws", plan._ws); rng("status", plan._status)
            for k in ("x", "edge", "e_attr", "bbox_idx"): rng(k, getattr(data, k))
            for key, ent in plan._graphs.items():
                if ent: rng("static_out", ent[1])
            print("  reserved MB", torch.cuda.memory_reserved() >> 20, flush=True)
        print(i, "graphs", ng, "max diff", float((out - ref).abs().max()), flush=True)
torch.cuda.synchronize(); print("done")
