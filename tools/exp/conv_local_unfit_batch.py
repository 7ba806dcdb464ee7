"""What a batch WITHOUT the proposal-local property costs: cfg 5 with one 200-node proposal, per-layer path vs the default (conv_local raises its flag, its workgroups stop at their next tile, the gated per-layer launches run)."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import yolat_vectorgraphicsrecognition_amd as yv
data, slices, optkw, _ = yv.config("5")
model = yv.SparseCADGCN(yv.Opt(**optkw)).cuda().eval().set_eval_precision("bf16")
# one 200-node proposal in the middle: merge 8 proposals
bb = data.bbox_idx.clone()
bb[100000:100200] = bb[100000]
_, inv = torch.unique_consecutive(bb, return_inverse=True)
bad = yv.Data(**{k: v for k, v in data.__dict__.items() if not k.startswith("_")})
bad.bbox_idx = inv
P2 = int(inv.max()) + 1
bad.bbox = data.bbox[:P2].clone(); bad.stat_feats = data.stat_feats[:P2].clone(); bad.labels = data.labels[:P2].clone()
def t(d, mode):
    os.environ["YOLAT_CONV_LOCAL"] = str(mode)
    x, edge, ea, b = d.x.cuda(), d.edge.cuda(), d.e_attr.cuda(), d.bbox_idx.cuda()
    P = int(d.bbox.shape[0])
    with torch.no_grad():
        model(d, None); plan = model._yolat_plan
        for _ in range(5): out = plan.run(x, edge, ea, b, P)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(50): out = plan.run(x, edge, ea, b, P)
        torch.cuda.synchronize()
    return (time.perf_counter() - t0) / 50 * 1e3, out.clone()
a0, o0 = t(bad, 0); a1, o1 = t(bad, 1); g1, _ = t(data, 1)
print("batch with one 200-node proposal: per-layer path %.4f ms | default (conv_local raises the flag, exits early, fall-back runs) %.4f ms | identical %s | good batch %.4f ms" % (a0, a1, bool(torch.equal(o0, o1)), g1))
