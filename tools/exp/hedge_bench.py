"""Time the bf16 edge stage (yolat_edge_uv_mlp2_mean_eval_bf16) on a cfg-sized graph: node tiles vs chained MFMA waves.
usage: python tools/exp/hedge_bench.py [cfg=5] [reps=20] [variant=2]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import yolat_vectorgraphicsrecognition_amd as yv
from yolat_vectorgraphicsrecognition_amd._lib import lib, check

cfg = sys.argv[1] if len(sys.argv) > 1 else "5"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
variant = int(sys.argv[3]) if len(sys.argv) > 3 else 2
data, _, _, _ = yv.config(cfg)
g = yv.ops.build_graph(data.edge.cuda(), data.e_attr.cuda(), data.bbox_idx.cuda(), int(data.x.shape[0]),
                       int(data.bbox.shape[0]))
E, N, C = g.E, g.N, 64
gen = torch.Generator().manual_seed(0)
UV = torch.randn(N, 2 * C, generator=gen).to(torch.bfloat16).cuda()
W2f = (torch.randn(C, C, generator=gen) / 8).to(torch.bfloat16).cuda()
wc4 = torch.randn(C, 4, generator=gen).cuda()
s1 = (torch.rand(C, generator=gen) + 0.5).cuda()
t2f = torch.randn(C, generator=gen).cuda()
root = torch.randn(N, C, generator=gen).cuda()
out = torch.zeros(N, C, dtype=torch.bfloat16, device="cuda")
st = torch.cuda.current_stream().cuda_stream


def run(v):
    check(lib.yolat_edge_uv_mlp2_mean_eval_bf16(UV.data_ptr(), 128, g.src.data_ptr(), g.dst.data_ptr(), g.attr.data_ptr(),
                                                g.row_ptr.data_ptr(), N, E, wc4.data_ptr(), s1.data_ptr(), W2f.data_ptr(),
                                                t2f.data_ptr(), root.data_ptr(), C, out.data_ptr(), C, v, st))


for v in ([1, 2] if variant == 0 else [variant]):
    for _ in range(3):
        run(v)
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        run(v)
    e.record()
    torch.cuda.synchronize()
    t = s.elapsed_time(e) / reps * 1e3
    by = N * 128 * 2.0 + E * 24.0 + N * 64 * 6.0 + 4.0 * N
    print("cfg %s N=%d E=%d variant %d WGS=%s: %.1f us   compulsory %.0f MB -> %.2f TB/s   finite %s" %
          (cfg, N, E, v, os.environ.get("YOLAT_HCHAIN_WGS", "-"), t, by / 1e6, by / t / 1e6,
           bool(torch.isfinite(out.float()).all())))
