import os, sys, numpy as np, torch
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import test_gpu_ops as T
yv = T._yv()
from yolat_vectorgraphicsrecognition_amd._lib import lib, check
N, E = 70, 300
src, dst, _, attr = T._edge_case(N, E, 64, 13 * N + E, ldx=64)
dst[:150] = N // 3
g = yv.ops.build_graph(T.dev(np.stack([src, dst], 1)), T.dev(attr), None, N, 1)
tg = torch.Generator().manual_seed(N + 7 * E)
UV = torch.randn(N, 128, generator=tg).cuda(); W2 = (torch.randn(64, 64, generator=tg) / 8).cuda()
wc4 = (torch.randn(64, 4, generator=tg) * 0.3).cuda()
b1, b2 = (torch.randn(64, generator=tg) * 0.1).cuda(), (torch.randn(64, generator=tg) * 0.1).cuda()
p1 = ((torch.rand(64, generator=tg) - 0.2).cuda(), (torch.randn(64, generator=tg) * 0.2).cuda())
p2 = ((torch.rand(64, generator=tg) - 0.2).cuda(), (torch.randn(64, generator=tg) * 0.2).cuda())
base = torch.randn(N, 128, generator=tg).cuda()
st = torch.cuda.current_stream().cuda_stream
H2 = torch.empty(E, 64).cuda()
check(lib.yolat_edge_uv_mlp2_eval(UV.data_ptr(), 128, g.src.data_ptr(), g.dst.data_ptr(), g.attr.data_ptr(), E, wc4.data_ptr(), b1.data_ptr(), p1[0].data_ptr(), p1[1].data_ptr(), W2.data_ptr(), b2.data_ptr(), p2[0].data_ptr(), p2[1].data_ptr(), 64, H2.data_ptr(), 64, st))
want = base.clone(); yv.ops.csr_mean_fwd(H2, g, want[:, 64:], accumulate=True)
got = base.clone()
check(lib.yolat_edge_uv_mlp2_mean_eval(UV.data_ptr(), 128, g.src.data_ptr(), g.dst.data_ptr(), g.attr.data_ptr(), g.row_ptr.data_ptr(), N, E, wc4.data_ptr(), b1.data_ptr(), p1[0].data_ptr(), p1[1].data_ptr(), W2.data_ptr(), b2.data_ptr(), p2[0].data_ptr(), p2[1].data_ptr(), 64, got[:, 64:].data_ptr(), 128, st))
d = (got - want).abs()
rows = torch.nonzero(d.max(1)[0] > 0).flatten().tolist()
rp = g.row_ptr.cpu().numpy()
print("rows differing:", rows, "max", float(d.max()))
for r in rows[:6]:
    print(r, "deg", rp[r + 1] - rp[r], "range", rp[r], rp[r + 1], "tile", r // 13, "maxdiff", float(d[r].max()), "ncols", int((d[r] > 0).sum()))
