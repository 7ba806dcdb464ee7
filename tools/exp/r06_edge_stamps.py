"""Phase timeline of k_edge_uv_mlp2_mean<1> (cfg 2's two edge launches), round 6: a debug build of edge.hip
(-DYOLAT_EDGE_STAMPS, tools/exp/r06_edge_stamps.sh) stamps the 100 MHz wall clock at the phase borders of every node-tile
workgroup; [0] = the launch that also runs the next layer's node side, [1] = the last layer's launch."""
import ctypes
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tests"))
import numpy as np  # noqa: E402
import torch  # noqa: E402

import bench  # noqa: E402
import golden_util as gu  # noqa: E402
import yolat_vectorgraphicsrecognition_amd as yv  # noqa: E402
from yolat_vectorgraphicsrecognition_amd._lib import lib  # noqa: E402

cfg = sys.argv[1] if len(sys.argv) > 1 else "2"
data, slices, optkw, _ = yv.config(cfg)
model = gu.fill_state_(yv.SparseCADGCN(yv.Opt(**optkw)), 0).cuda().eval()
bench.to_device(data)
for _ in range(30):
    data._yolat_stage = None
    with torch.no_grad():
        model(data, slices)
torch.cuda.synchronize()
n = 2 * 4096 * 16
buf = (ctypes.c_longlong * n)()
fn = lib.yolat_debug_edge_stamps
fn.argtypes = [ctypes.c_void_p, ctypes.c_int]
assert fn(buf, n) == 0
st = np.frombuffer(buf, dtype=np.int64).reshape(2, 4096, 16)
names = ["index / f_out / W2 loads issued, W2 -> LDS (0->1)", "barrier (1->2)", "UV gather + layer 1 -> LDS (2->3)", "barrier (3->4)",
         "32 fp32 MFMAs (4->5)", "barrier, messages -> LDS, barrier (5->6)", "per-node sums (6->7)", "barrier(s) / further passes (7->8)",
         "f_out store (8->9)", "next layer's node side (9->10)"]
for w, label in ((0, "launch with the next layer's node side"), (1, "last layer's launch")):
    s = st[w]
    s = s[s[:, 0] > 0]
    if not len(s):
        continue
    t0 = s[:, 0].min()
    print("== %s: %d node-tile workgroups, passes per workgroup %s, span %.2f us" % (
        label, len(s), np.unique(s[:, 11]), (s[:, 10].max() - t0) / 100.0))
    print("   start after launch begin: median %.2f  p90 %.2f  max %.2f us" % (
        np.median(s[:, 0] - t0) / 100.0, np.percentile(s[:, 0] - t0, 90) / 100.0, (s[:, 0] - t0).max() / 100.0))
    for k, nm in enumerate(names):
        d = (s[:, k + 1] - s[:, k]) / 100.0
        print("   %-58s median %.2f  p90 %.2f us" % (nm, np.median(d), np.percentile(d, 90)))
    print("   whole workgroup: median %.2f  max %.2f us" % (np.median(s[:, 10] - s[:, 0]) / 100.0, (s[:, 10] - s[:, 0]).max() / 100.0))
