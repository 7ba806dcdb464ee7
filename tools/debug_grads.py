"""Dev tool: per-parameter gradient error of the HIP path vs the CPU oracle (fp32 and fp64)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
import golden_util as gu
from oracle import oracle_torch as orc
import yolat_vectorgraphicsrecognition_amd as yv

kinds = sys.argv[1:] or ["medium"]
for kind in kinds:
    seed = {"tiny": 101, "small": 102, "medium": 103, "deep": 104}[kind]
    arrs, optkw = gu.graph_case(kind)
    ref = gu.fill_state_(orc.SparseCADGCN(orc.Opt(**optkw)), seed).double()
    d = gu.to_data(arrs, yv.Data); d.x = d.x.double(); d.e_attr = d.e_attr.double()
    ref.train()
    out = ref(d, None); loss = orc.DetectionLoss(orc.Opt(**optkw))(out, d)["loss"]; loss.backward()
    model = gu.fill_state_(yv.SparseCADGCN(yv.Opt(**optkw)), seed).cuda()
    model.train()
    data = gu.to_data(arrs, yv.Data)
    o = model(data, None); l = yv.DetectionLoss(yv.Opt(**optkw))(o, data)["loss"]; l.backward()
    print(kind, "loss", float(l), float(loss), "logits err", float((o[0].cpu().double() - out[0]).abs().max()))
    rp = dict(ref.named_parameters())
    for n, p in model.named_parameters():
        a, b = p.grad.cpu().double(), rp[n].grad
        print("  %-52s scale %.2e  relerr %.2e" % (n, float(b.abs().max()), float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))))
