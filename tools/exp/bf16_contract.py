"""The bf16-storage eval forward against the fp32 HIP forward (itself pinned to 1e-4 of the CPU oracle at full size) on the
BASELINE configurations: max / rms error of the logits relative to their scale and arg-max agreement, for the one-launch
conv stack (messages rounded to bf16 in front of the aggregation MFMA) and for the per-layer launches (fp32 messages into the
mean) — is the extra rounding the dominant error term?   usage: python tools/exp/bf16_contract.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
import golden_util as gu
import yolat_vectorgraphicsrecognition_amd as yv


def rel(got, want):
    got, want = got.double(), want.double()
    sc = float(want.abs().max())
    return (float((got - want).abs().max()) / sc, float((got - want).pow(2).mean().sqrt() / want.pow(2).mean().sqrt()),
            float((got.argmax(1) == want.argmax(1)).double().mean()))


for cfg in ("1", "2", "4", "5"):
    data, slices, optkw, _ = yv.config(cfg)
    for seed in (55, 9):
        model = gu.fill_state_(yv.SparseCADGCN(yv.Opt(**optkw)), seed).cuda().eval()
        for k in ("x", "edge", "e_attr", "bbox_idx", "bbox"):
            data[k] = data[k].cuda()
        with torch.no_grad():
            want = model(data, slices)[0].clone()
            model.set_eval_precision("bf16")
            out = {}
            for mode, tag in (("2", "one-launch stack (bf16 messages)"), ("0", "per-layer launches (fp32 messages)")):
                os.environ["YOLAT_CONV_LOCAL"] = mode
                out[tag] = model(data, slices)[0].clone()
        os.environ.pop("YOLAT_CONV_LOCAL", None)
        for tag, got in out.items():
            mx, rms, agree = rel(got, want)
            print("cfg %s seed %2d  %-36s max %.3e  rms %.3e  arg-max agreement %.4f" % (cfg, seed, tag, mx, rms, agree))
        a, b = list(out.values())
        mx, rms, agree = rel(a, b)
        print("cfg %s seed %2d  %-36s max %.3e  rms %.3e  arg-max agreement %.4f" % (cfg, seed, "stack vs per-layer", mx, rms, agree))
        # how close are the two largest logits where the arg-max flips?  (a flip between two classes 1e-3 apart is a tie)
        got = a
        flip = (got.argmax(1) != want.argmax(1))
        if bool(flip.any()):
            top2 = want[flip].topk(2, dim=1).values
            gap = (top2[:, 0] - top2[:, 1]) / want.abs().max()
            print("          flips: %d of %d; fp32 top-2 gap of the flipped rows / scale: median %.2e  max %.2e"
                  % (int(flip.sum()), got.shape[0], float(gap.median()), float(gap.max())))
