"""A/B of fusion-stage variants inside one process (round 6): eval forward of a config under several values of an
environment switch the launcher reads at launch time; prints ms per forward, the fusion stage's HIP-event time and whether
the logits are bit-equal to the first variant's.

    python tools/exp/r06_fusion_ab.py YOLAT_H8_AB n,w,v 5:bf16 2:bf16 1:bf16
"""
import os
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tests"))
import torch  # noqa: E402

import bench  # noqa: E402
import golden_util as gu  # noqa: E402
import yolat_vectorgraphicsrecognition_amd as yv  # noqa: E402


def main():
    var, values = sys.argv[1], sys.argv[2].split(",")
    cases = sys.argv[3:] or ["5:bf16"]
    for case in cases:
        cfg, precision = case.split(":")
        data, slices, optkw, n_graphs = yv.config(cfg)
        model = gu.fill_state_(yv.SparseCADGCN(yv.Opt(**optkw)), 0).cuda().eval()
        model.set_eval_precision(precision)
        bench.to_device(data)

        def one():
            data._yolat_stage = None
            with torch.no_grad():
                return model(data, slices)[0]

        ref = None
        for rnd in range(2):
            for v in values:
                if v == "-":
                    os.environ.pop(var, None)
                else:
                    os.environ[var] = v
                for _ in range(10):
                    one()
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(100):
                    out = one()
                torch.cuda.synchronize()
                ms = (time.perf_counter() - t0) / 100 * 1e3
                table = bench.plan_profile(one, 20)
                fus = [(k, x["ms_avg"] * 1e3) for k, x in table.items() if k.startswith("fusion")]
                out = out.float().clone()
                if ref is None:
                    ref = out
                same = bool(torch.equal(out, ref))
                print("cfg %s %s  %s=%-3s  %.4f ms/forward  fusion %s us  bit-equal to first: %s  max|d| %.3g" % (
                    cfg, precision, var, v, ms, ",".join("%.1f" % u for _, u in fus), same,
                    float((out - ref).abs().max())), flush=True)
        del model


if __name__ == "__main__":
    main()
