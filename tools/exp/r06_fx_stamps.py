"""Where a workgroup of k_fusion_rows_x6<128> spends its time (round 6): a debug build of the library (fusion_x6.hip compiled
with -DYOLAT_FX_STAMPS, linked as tools/exp/libyolat_hip_stamps.so — see tools/exp/r06_fx_stamps.sh) stamps the wall clock
(100 MHz) at the phase borders of every workgroup; this script runs the eval forward of a config, reads the stamps of the
last forward and prints the timeline.   usage: YOLAT_LIB_PATH=tools/exp/libyolat_hip_stamps.so python tools/exp/r06_fx_stamps.py 2"""
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


def main():
    cfg = sys.argv[1] if len(sys.argv) > 1 else "2"
    h8 = len(sys.argv) > 2 and sys.argv[2] == "h8"      # the bf16 kernel (fusion_h8.hip, -DYOLAT_H8_STAMPS): tiles 4..7
    data, slices, optkw, _ = yv.config(cfg)
    model = gu.fill_state_(yv.SparseCADGCN(yv.Opt(**optkw)), 0).cuda().eval()
    if h8:
        model.set_eval_precision("bf16")
    bench.to_device(data)

    def one():
        data._yolat_stage = None
        with torch.no_grad():
            return model(data, slices)[0]
    for _ in range(30):
        one()
    torch.cuda.synchronize()
    n = 4096 * 32
    buf = (ctypes.c_longlong * n)()
    zero = (ctypes.c_longlong * n)()
    fn = lib.yolat_debug_h8_stamps if h8 else lib.yolat_debug_fx_stamps
    fn.argtypes = [ctypes.c_void_p, ctypes.c_int]
    rc = fn(buf, n)
    assert rc == 0, rc
    st = np.frombuffer(buf, dtype=np.int64).reshape(4096, 32).copy()
    live = st[:, 0] > 0
    st = st[live]
    t0 = st[:, 0].min()
    big = st[:, 22] == 0
    print("workgroups with stamps: %d (big problem %d, small %d); launch span %.2f us (first start -> last end)" % (
        len(st), big.sum(), (~big).sum(), (st[:, 20].max() - t0) / 100.0))
    for name, sel in (("big", big), ("small", ~big)):
        s = st[sel]
        if not len(s):
            continue
        ngl = s[:, 21]
        us = lambda a: a / 100.0
        print("== %s problem: %d workgroups, column tiles per workgroup: %s" % (name, len(s), np.unique(ngl)))
        print("   start after launch begin  : median %.2f  p90 %.2f  max %.2f us" % (
            us(np.median(s[:, 0] - t0)), us(np.percentile(s[:, 0] - t0, 90)), us((s[:, 0] - t0).max())))
        print("   A load + split (0->1)     : median %.2f  max %.2f us" % (us(np.median(s[:, 1] - s[:, 0])), us((s[:, 1] - s[:, 0]).max())))
        print("   runs + W0 issue (1->2)    : median %.2f us" % us(np.median(s[:, 2] - s[:, 1])))
        print("   barrier + W0 -> LDS (2->3): median %.2f us" % us(np.median(s[:, 3] - s[:, 2])))
        for j in range(int(min(4, ngl.max()))):
            m = ngl > (j + 4 if h8 else j)
            if not m.any():
                continue
            prev = (s[m, 28] if h8 else s[m, 3]) if j == 0 else s[m, 7 + 4 * (j - 1)]
            if h8:
                print("   tile %d: drain %.2f |" % (j + 4, us(np.median(s[m, 24 + j] - prev))), end="")
                prev = s[m, 24 + j]
            print("   tile %d: MFMAs %.2f | store_w + vmcnt(0) %.2f | epilogue %.2f | barrier %.2f us   (median over %d)" % (
                j, us(np.median(s[m, 4 + 4 * j] - prev)), us(np.median(s[m, 5 + 4 * j] - s[m, 4 + 4 * j])),
                us(np.median(s[m, 6 + 4 * j] - s[m, 5 + 4 * j])), us(np.median(s[m, 7 + 4 * j] - s[m, 6 + 4 * j])), m.sum()))
        print("   whole workgroup (0->end)  : median %.2f  max %.2f us;  end after launch begin: median %.2f  max %.2f us" % (
            us(np.median(s[:, 20] - s[:, 0])), us((s[:, 20] - s[:, 0]).max()), us(np.median(s[:, 20] - t0)), us((s[:, 20] - t0).max())))
        if not h8 and s[:, 25].max() > 0:
            e = s[:, 25:30]
            print("   last tile's run walk (wave 0): rows 0-3 %.2f | 4-7 %.2f | 8-11 %.2f | 12-15 %.2f us (median)" % tuple(
                us(np.median(e[:, i + 1] - e[:, i])) for i in range(4)))
        xcc = s[:, 23]
        print("   workgroups per XCC: %s" % np.bincount(xcc.astype(int) & 15)[:8])


if __name__ == "__main__":
    main()
