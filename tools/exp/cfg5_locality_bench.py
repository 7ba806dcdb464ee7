"""cfg-5 bf16 forward with the batch's locality decided before enqueue (plan.LOCALITY_CACHE) against the gated form:
ms per forward (resident inputs, HIP events) and the C-side stage table of each.
usage: python tools/exp/cfg5_locality_bench.py [cfg=5] [reps=50]"""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import yolat_vectorgraphicsrecognition_amd as yv
from yolat_vectorgraphicsrecognition_amd import plan as plan_mod
from yolat_vectorgraphicsrecognition_amd._lib import lib

cfg = sys.argv[1] if len(sys.argv) > 1 else "5"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 50
data, slices, optkw, _ = yv.config(cfg)
torch.manual_seed(0)
model = yv.SparseCADGCN(yv.Opt(**optkw)).cuda().eval().set_eval_precision("bf16")
for k in ("x", "edge", "e_attr", "bbox_idx", "bbox"):
    setattr(data, k, getattr(data, k).cuda())


def one():
    data._yolat_stage = None
    with torch.no_grad():
        return model(data, slices)[0]


def stages(n=20):
    lib.yolat_profile_reset()
    lib.yolat_profile_enable(1)
    for _ in range(n):
        one()
    torch.cuda.synchronize()
    lib.yolat_profile_enable(0)
    name = ctypes.create_string_buffer(128)
    ms, calls, fl, by = ctypes.c_float(), ctypes.c_int(), ctypes.c_double(), ctypes.c_double()
    out = []
    for i in range(lib.yolat_profile_count()):
        lib.yolat_profile_get(i, name, 128, ctypes.byref(ms), ctypes.byref(calls), ctypes.byref(fl), ctypes.byref(by))
        out.append((name.value.decode(), ms.value / max(calls.value, 1) * 1e3))
    lib.yolat_profile_reset()
    return out


def timed():
    for _ in range(10):
        one()
    torch.cuda.synchronize()
    best = []
    for _ in range(5):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(reps):
            out = one()
        e.record()
        torch.cuda.synchronize()
        best.append(s.elapsed_time(e) / reps)
    best.sort()
    return best[len(best) // 2], out


res = {}
for tag, on in (("gated (found out on the device)", False), ("locality decided before enqueue", True), ("gated again", False),
                ("decided again", True)):
    plan_mod.LOCALITY_CACHE = on
    t, out = timed()
    res[tag] = out.clone()
    print("%-36s %.4f ms per forward" % (tag, t))
    for n, us in stages():
        print("      %-100s %8.1f us" % (n[:100], us))
print("bit-identical logits:", bool(torch.equal(res["gated (found out on the device)"], res["locality decided before enqueue"])))
model._yolat_plan.check_status()
