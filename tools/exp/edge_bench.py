"""Time yolat_edge_uv_mlp2_mean_eval (factorised edge MLP + mean) on a cfg-sized graph and print a checksum of its
output, so variants selected by YOLAT_EDGE_V2 / YOLAT_EDGE_WGS (one process each) can be compared bit for bit.
usage: python tools/exp/edge_bench.py [cfg=5] [reps=20]"""
import hashlib
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import yolat_vectorgraphicsrecognition_amd as yv
from yolat_vectorgraphicsrecognition_amd._lib import lib, check

cfg = sys.argv[1] if len(sys.argv) > 1 else "5"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
data, _, _, _ = yv.config(cfg)
g = yv.ops.build_graph(data.edge.cuda(), data.e_attr.cuda(), data.bbox_idx.cuda(), int(data.x.shape[0]),
                       int(data.bbox.shape[0]))
E, N, C = g.E, g.N, 64
gen = torch.Generator().manual_seed(0)
UV = torch.randn(N, 2 * C, generator=gen).cuda()
W2 = (torch.randn(C, C, generator=gen) / 8).cuda()
wc4 = torch.randn(C, 4, generator=gen).cuda()
vec = [torch.randn(C, generator=gen).cuda() for _ in range(6)]
root = torch.randn(N, C, generator=gen).cuda()
f_out = root.clone()
st = torch.cuda.current_stream().cuda_stream


FOLD = os.environ.get("FOLD", "0") == "1"


def run():
    if FOLD:
        check(lib.yolat_edge_uv_mlp2_mean_eval(UV.data_ptr(), 2 * C, g.src.data_ptr(), g.dst.data_ptr(), g.attr.data_ptr(),
                                               g.row_ptr.data_ptr(), N, E, wc4.data_ptr(), None, None, None, W2.data_ptr(),
                                               None, vec[4].data_ptr(), vec[5].data_ptr(), C, f_out.data_ptr(), C, st))
        return
    check(lib.yolat_edge_uv_mlp2_mean_eval(UV.data_ptr(), 2 * C, g.src.data_ptr(), g.dst.data_ptr(), g.attr.data_ptr(),
                                           g.row_ptr.data_ptr(), N, E, wc4.data_ptr(), vec[0].data_ptr(),
                                           vec[1].data_ptr(), vec[2].data_ptr(), W2.data_ptr(), vec[3].data_ptr(),
                                           vec[4].data_ptr(), vec[5].data_ptr(), C, f_out.data_ptr(), C, st))


run()
torch.cuda.synchronize()
digest = hashlib.sha1(f_out.cpu().numpy().tobytes()).hexdigest()[:16]
ref_path = "/tmp/edge_ref_%s_%d.pt" % (cfg, FOLD)
err = ""
if os.path.exists(ref_path):
    ref = torch.load(ref_path)
    d = (f_out.cpu().double() - ref.double() - 0).abs()
    # f_out = root + mean: compare the aggregated part against its own scale
    agg_ref = (ref.double() - root.cpu().double())
    err = " | vs ref: max|d| %.3e, scale %.3e, rel %.2e" % (float(d.max()), float(agg_ref.abs().max()),
                                                           float(d.max() / agg_ref.abs().max()))
else:
    torch.save(f_out.cpu(), ref_path)
for _ in range(3):
    run()
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
s.record()
for _ in range(reps):
    run()
e.record()
torch.cuda.synchronize()
t = s.elapsed_time(e) / reps * 1e3
print("FOLD=%d " % FOLD + "cfg %s N=%d E=%d  WS=%s WGS=%s NPT=%s : %.1f us  sha1 %s  finite %s" %
      (cfg, N, E, os.environ.get("YOLAT_EDGE_WS", "-"), os.environ.get("YOLAT_EDGE_WGS", "-"),
       os.environ.get("YOLAT_EDGE_NPT", "-"), t, digest, bool(torch.isfinite(f_out).all())) + " X6=%s" %
      os.environ.get("YOLAT_EDGE_X6", "-") + err)
