import sys, os
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import torch
import test_gpu_bf16 as T
import golden_util as gu
from oracle import oracle_torch as orc
yv = T._yv()
arrs, optkw = gu.graph_case("deep")
data = gu.to_data(arrs, yv.Data)
def oracle(x):
    ref = gu.fill_state_(orc.SparseCADGCN(orc.Opt(**optkw)), 77).double().train()
    d = gu.to_data(arrs, yv.Data)
    d.x = x.double(); d.e_attr = d.e_attr.double()
    out = ref(d, None)
    loss = orc.DetectionLoss(orc.Opt(**optkw))(out, d)["loss"]
    loss.backward()
    return float(loss.detach()), {n: p.grad.detach().double() for n, p in ref.named_parameters()}
l64, g64 = oracle(data.x)
_, g64p = oracle(T._perturb_x(data))
l16, g16, _ = T._train_once(yv, optkw, data, 77, "bf16")
rows = []
for n, ref in g64.items():
    a = g16[n].detach().cpu().double()
    err, sens, rms = T._rms(a - ref), T._rms(g64p[n] - ref), T._rms(ref)
    if rms > 0 and float(ref.abs().max()) >= 1e-9 * max(float(v.abs().max()) for v in g64.values()):
        rows.append(((err - 2e-2 * rms) / max(sens, 1e-300), n, err, sens, rms))
rows.sort(reverse=True)
for r in rows[:5]:
    print("factor needed %.2f  %s  err %.3e sens %.3e rms %.3e" % r)
