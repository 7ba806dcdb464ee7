"""Time the one-launch proposal-local conv stack (csrc/conv_local.hip) on a cfg-sized graph, standalone and inside the bf16
forward, next to the per-layer launches it replaces.
usage: python tools/exp/conv_local_bench.py [cfg=5] [reps=50]"""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import yolat_vectorgraphicsrecognition_amd as yv
from yolat_vectorgraphicsrecognition_amd import ops
from yolat_vectorgraphicsrecognition_amd._lib import lib, check, GraphCsr

cfg = sys.argv[1] if len(sys.argv) > 1 else "5"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 50
data, slices, optkw, _ = yv.config(cfg)
torch.manual_seed(0)
model = yv.SparseCADGCN(yv.Opt(**optkw)).cuda().eval().set_eval_precision("bf16")
dev = torch.device("cuda")


def fwd_ms(mode):
    os.environ["YOLAT_CONV_LOCAL"] = str(mode)
    lib.yolat_conv_local_tune(int(os.environ.get("CL_NW", "0")), 0, 0, None)
    with torch.no_grad():
        for _ in range(5):
            model(data, slices)
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        # inputs resident: time the plan alone
        plan = model._yolat_plan
        x, edge, ea, bb = (data.x.cuda(), data.edge.cuda(), data.e_attr.cuda(), data.bbox_idx.cuda())
        P = int(data.bbox.shape[0])
        for _ in range(5):
            plan.run(x, edge, ea, bb, P)
        torch.cuda.synchronize()
        s.record()
        for _ in range(reps):
            out = plan.run(x, edge, ea, bb, P)
        e.record()
        torch.cuda.synchronize()
    return s.elapsed_time(e) / reps, out


t0, out0 = fwd_ms(0)
t1, out1 = fwd_ms(1)
t2, out2 = fwd_ms(2)
t3, out3 = fwd_ms(3)
print('forward without the gated fall-back launches (measurement only): %.4f ms' % t3)
d = (out1.double() - out0.double())
print("cfg %s forward bf16: per-layer %.4f ms | local (auto) %.4f ms | local (forced) %.4f ms | max diff %.3e of scale %.3e, same as forced %s"
      % (cfg, t0, t1, t2, float(d.abs().max()), float(out0.abs().max()), bool(torch.equal(out1, out2))))

# standalone op
plan = model._yolat_plan
h, base = plan._desc_h, plan._desc
N, P = int(data.x.shape[0]), int(data.bbox.shape[0])
g = ops.build_graph(data.edge.cuda(), data.e_attr.cuda(), data.bbox_idx.cuda(), N, P)
D, F = base.C * base.n_blocks_out, base.F
feats = torch.zeros((N, D), dtype=torch.bfloat16, device=dev)
Z = torch.zeros((P, 2 * (F + D)), dtype=torch.float32, device=dev)
flag = torch.zeros(1, dtype=torch.int32, device=dev)
x = data.x.cuda().contiguous()
gc = GraphCsr(*g.device_pointers())
st = torch.cuda.current_stream().cuda_stream


def run():
    check(lib.yolat_conv_stack_local_bf16(ctypes.byref(h), h.conv_local, x.data_ptr(), x.stride(0), ctypes.byref(gc), N, g.E, P,
                                          feats.data_ptr(), D, Z.data_ptr(), Z.stride(0), flag.data_ptr(), st))


NW = int(os.environ.get("CL_NW", "0"))
abls = ["0"] if len(sys.argv) <= 4 else sys.argv[4].split(",")
for g0, abl in [(g, ab) for g in (["auto"] if len(sys.argv) <= 3 else sys.argv[3].split(","))
                for ab in abls]:
    lib.yolat_conv_local_tune(NW, 0 if g0 == "auto" else int(g0), int(abl), None)
    for _ in range(5):
        run()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        run()
    e.record()
    torch.cuda.synchronize()
    t = s.elapsed_time(e) / reps * 1e3
    by = 4.0 * N * 5 + 24.0 * g.E + 4.0 * N + 2.0 * N * D + 4.0 * P * (F + 2 * D)
    print("conv_local standalone N=%d E=%d P=%d L=%d G0=%s abl=%s: %.1f us  flag %d  compulsory %.0f MB -> %.2f TB/s  finite %s"
          % (N, g.E, P, base.n_blocks, g0 or "auto", abl, t, int(flag.item()), by / 1e6, by / t / 1e6,
             bool(torch.isfinite(feats.float()).all())))

# ---- in-kernel phase stamps (s_memtime, shader cycles) of the first tiles of every workgroup
if os.environ.get("CL_STAMPS", "1") == "1":
    import numpy as np
    nwg = 4096
    stamps = torch.zeros(nwg * 64, dtype=torch.int64, device=dev)
    lib.yolat_conv_local_tune(NW, 0, 0, stamps.data_ptr())
    run()
    torch.cuda.synchronize()
    lib.yolat_conv_local_tune(0, 0, 0, None)
    st_ = stamps.cpu().numpy().reshape(nwg, 64)
    used = st_[:, 0] != 0
    st_ = st_[used]
    L = base.n_blocks
    per_tile = 3 + 5 * L
    names = ["tile load", "stream set-up"]
    for l in range(L):
        names += ["L%d node phase" % l, "L%d node barrier" % l, "L%d mean-pool + steps" % l, "L%d finalize+barrier" % l, "L%d outputs" % l]
    for tile in range(2):
        seg = st_[:, tile * per_tile:(tile + 1) * per_tile + 1].astype(np.float64)
        ok = (seg != 0).all(1)
        seg = seg[ok]
        d = np.diff(seg, axis=1)
        print("tile %d: %d workgroups, total %.0f cycles (to next tile start)" % (tile, len(seg), (seg[:, -1] - seg[:, 0]).mean()))
        for i, nme in enumerate(names + ["zero Z + barrier"]):
            if i < d.shape[1]:
                print("   %-24s mean %7.0f  p10 %7.0f  p90 %7.0f" % (nme, d[:, i].mean(), np.percentile(d[:, i], 10), np.percentile(d[:, i], 90)))
    print("workgroup start spread: %.0f cycles; end-to-end (first start .. last stamp) %.0f cycles" %
          (st_[:, 0].max() - st_[:, 0].min(), st_.max() - st_[:, 0].min()))
