"""Which Python lines issue the small device copies / fills of a cfg-3 train step (torch.profiler with stacks)."""
import os, sys, torch
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tests"))
import yolat_vectorgraphicsrecognition_amd as yv
import golden_util as gu
data, slices, optkw, _ = yv.config("3")
opt = yv.Opt(**optkw)
model = gu.fill_state_(yv.SparseCADGCN(opt), 0).cuda()
import bench
bench.to_device(data)
tr = yv.Trainer(model, opt, lr=2.5e-4, weight_decay=1e-5)
for _ in range(3):
    data._yolat_stage = None
    tr.step(data, slices)
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU], with_stack=True, record_shapes=True) as prof:
    data._yolat_stage = None
    tr.step(data, slices)
    torch.cuda.synchronize()
seen = {}
for ev in prof.events():
    if ev.name in ("aten::copy_", "aten::fill_", "aten::zero_", "aten::clone", "aten::contiguous", "aten::zeros", "aten::_to_copy"):
        st = [s for s in ev.stack if "yolat_vectorgraphicsrecognition_amd" in s or "bench.py" in s][:2]
        key = (ev.name, tuple(st), str(ev.input_shapes)[:60])
        seen[key] = seen.get(key, 0) + 1
for (name, st, shp), n in sorted(seen.items(), key=lambda kv: -kv[1]):
    print(n, name, shp, " <- ", " | ".join(s.split("yolat_vectorgraphicsrecognition_amd/")[-1] for s in st))
