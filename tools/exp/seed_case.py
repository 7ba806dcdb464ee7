import sys, os, numpy as np, torch
sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "tests"))
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import golden_util as gu
from oracle import oracle_torch as orc
import yolat_vectorgraphicsrecognition_amd as yv
seed = int(sys.argv[1]) if len(sys.argv) > 1 else 6
rng = np.random.default_rng(1000 + seed)
n_graphs = int(rng.integers(1, 6)); nb = int(rng.integers(2, 5))
optkw = dict(n_classes=int(rng.integers(2, 23)), n_blocks=nb, n_blocks_out=int(rng.integers(1, nb + 1)))
kw = dict(num_proposals=int(rng.integers(1, 121)), nodes_lo=int(rng.integers(2, 5)), nodes_hi=int(rng.integers(5, 41)), n_classes=optkw["n_classes"])
if rng.random() < 0.5: kw["edge_factor"] = float(rng.uniform(0.3, 3.0))
else: kw["edges_per_proposal"] = int(rng.integers(1, 200))
if os.environ.get("NBO"):
    optkw["n_blocks_out"] = int(os.environ["NBO"])
if os.environ.get("NB"):
    optkw["n_blocks"] = int(os.environ["NB"])
data, slices = yv.synth_batch(n_graphs, 500 + seed, **kw)
ref = gu.fill_state_(orc.SparseCADGCN(orc.Opt(**optkw)), seed).double(); ref.train()
d = yv.Data(x=data.x.double(), pos=data.pos)
for k in ("edge", "bbox_idx", "bbox", "labels"): d[k] = data[k]
d.e_attr = data.e_attr.double()
out = ref(d, None); loss = orc.DetectionLoss(orc.Opt(**optkw))(out, d)["loss"]; loss.backward()
model = gu.fill_state_(yv.SparseCADGCN(yv.Opt(**optkw)), seed).cuda(); model.train()
o = model(data, slices); l = yv.DetectionLoss(yv.Opt(**optkw))(o, data)["loss"]; l.backward()
rp = dict(ref.named_parameters())
worst = sorted(((float((p.grad.cpu().double() - rp[n].grad).abs().max()) / max(float(rp[n].grad.abs().max()), 1e-12), n) for n, p in model.named_parameters() if float(rp[n].grad.abs().max()) > 1e-6), reverse=True)[:3]
print(os.environ.get("TAGV", ""), "loss err %.2e" % abs(float(l.detach()) - float(loss.detach())), worst)
if os.environ.get("ALLP"):
    for n, p in model.named_parameters():
        b = rp[n].grad
        print("%-52s rel %.2e  scale %.2e" % (n, float((p.grad.cpu().double() - b).abs().max()) / max(float(b.abs().max()), 1e-30), float(b.abs().max())))
    print("N", data.x.shape[0], "E", data.edge.shape[0], "P", data.bbox.shape[0], optkw, kw)
    # proposals with identical feature rows? (exact ties in the pass-through max pooling)
    bb = data.bbox_idx.numpy()
    cnt = np.bincount(bb, minlength=data.bbox.shape[0])
    print("nodes per proposal: min %d max %d; proposals with 0 nodes: %d" % (cnt.min(), cnt.max(), int((cnt == 0).sum())))
