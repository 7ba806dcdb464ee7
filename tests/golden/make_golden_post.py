"""Generates tests/golden/postprocess.npz.  DEV-TIME ONLY: needs /root/reference (absent on the GPU box); never
imported by tests.  Run from the repo root:  python tests/golden/make_golden_post.py

 1. The reference's OWN ``non_max_suppression`` (cad_recognition/train.py:34-121) is compiled from its source text
    where it lies (train.py itself cannot be imported: torch_geometric, torchvision, sklearn ... are absent) and run
    with a stand-in ``torchvision`` whose ``ops.nms`` is oracle_np.nms, the restatement of torchvision's published
    kernel.  So the candidate selection / class offsets / limits are pinned by the reference's code; only the
    third-party nms is restated.
 2. The reference's utils/det_util.py is imported as is (tqdm / matplotlib exist here) and its
    ``get_batch_statistics`` / ``ap_per_class`` run on those detections against synthetic targets.
 3. A tiny hand-checkable nms case with exactly representable coordinates.
"""
import ast
import importlib.util
import os
import sys
import time
import types

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
REF = "/root/reference"
sys.path.insert(0, REPO)
from oracle import oracle_np as onp          # noqa: E402


def reference_nms_function():
    path = os.path.join(REF, "cad_recognition", "train.py")
    src = open(path).read()
    node = next(n for n in ast.parse(src).body if isinstance(n, ast.FunctionDef) and n.name == "non_max_suppression")
    seg = ast.get_source_segment(src, node)
    tv = types.ModuleType("torchvision")
    tv.ops = types.SimpleNamespace(
        nms=lambda b, s, t: torch.from_numpy(onp.nms(b.numpy(), s.numpy(), float(t))))
    ns = {"torch": torch, "time": time, "torchvision": tv}
    exec(compile(seg, path, "exec"), ns)
    return ns["non_max_suppression"]


def synth_prediction(rng, n, nc, size=800.0):
    """[1, n, 5 + nc]: clustered boxes (so that NMS has work to do), obj conf, softmax class conf."""
    centers = rng.random((max(n // 6, 1), 2)) * size
    c = centers[rng.integers(0, len(centers), size=n)] + rng.normal(0, 6.0, size=(n, 2))
    wh = 20 + rng.random((n, 2)) * 60
    box = np.concatenate([c - wh / 2, c + wh / 2], 1)
    obj = rng.random((n, 1))
    logits = rng.normal(0, 2.0, size=(n, nc))
    cls = np.exp(logits) / np.exp(logits).sum(1, keepdims=True)
    return np.concatenate([box, obj, cls], 1).astype(np.float32)[None]


def main():
    out = {}
    ref_nms = reference_nms_function()
    rng = np.random.default_rng(7)
    cases = {"a": dict(n=300, nc=16, conf=0.0, iou=0.5, agnostic=False),     # the evaluation loop's call, train.py:448
             "b": dict(n=500, nc=21, conf=0.05, iou=0.45, agnostic=False),
             "c": dict(n=200, nc=1, conf=0.25, iou=0.45, agnostic=True)}     # single class: best-class branch
    for name, cs in cases.items():
        pred = synth_prediction(rng, cs["n"], cs["nc"])
        got = ref_nms(torch.from_numpy(pred.copy()), conf_thres=cs["conf"], iou_thres=cs["iou"], agnostic=cs["agnostic"])
        out["nms_%s/pred" % name] = pred
        out["nms_%s/args" % name] = np.array([cs["conf"], cs["iou"], float(cs["agnostic"])], dtype=np.float64)
        out["nms_%s/out" % name] = got[0].numpy()
        print("case %s: %d candidates -> %d detections" % (name, cs["n"] * cs["nc"], got[0].shape[0]))
    # class filter + a-priori labels branch
    pred = synth_prediction(rng, 120, 8)
    lab = torch.tensor([[2.0, 10, 10, 60, 70], [5.0, 300, 310, 380, 390]])
    got = ref_nms(torch.from_numpy(pred.copy()), conf_thres=0.1, iou_thres=0.5, classes=[2, 5], labels=[lab])
    out["nms_d/pred"], out["nms_d/labels"], out["nms_d/out"] = pred, lab.numpy(), got[0].numpy()

    # ---- det_util metrics on case a's detections
    spec = importlib.util.spec_from_file_location("ref_det_util", os.path.join(REF, "utils", "det_util.py"))
    du = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(du)
    det = torch.from_numpy(out["nms_a/out"])
    # targets: some detections' boxes jittered (true positives), some unrelated boxes
    k = 25
    pick = rng.choice(det.shape[0], size=k, replace=False)
    tb = det[pick, :4].numpy() + rng.normal(0, 2.0, size=(k, 4)).astype(np.float32)
    tl = det[pick, 5].numpy()
    extra = np.concatenate([rng.random((6, 2)) * 700, rng.random((6, 2)) * 50 + 20], 1)
    extra[:, 2:] += extra[:, :2]
    tb = np.concatenate([tb, extra]).astype(np.float32)
    tl = np.concatenate([tl, rng.integers(0, 16, size=6)]).astype(np.float32)
    targets = torch.from_numpy(np.concatenate([np.zeros((len(tb), 1), np.float32), tl[:, None], tb], 1))
    out["metrics/targets"] = targets.numpy()
    tps, confs, labs = [], None, None
    for th in (0.5, 0.75):
        m = du.get_batch_statistics([det], targets, iou_threshold=th)
        out["metrics/tp_%g" % th] = m[0][0]
        tps.append(m[0][0])
        confs, labs = m[0][1].numpy(), m[0][2].numpy()
    p, r, ap, f1, cls = du.ap_per_class(tps[0], confs, labs, tl)
    out["metrics/p"], out["metrics/r"], out["metrics/ap"], out["metrics/f1"], out["metrics/cls"] = p, r, ap, f1, cls
    iou = du.bbox_iou(det[:1, :4], torch.from_numpy(tb))
    out["metrics/bbox_iou_row0"] = iou.numpy()
    print("metrics: %d TP@0.5, mAP %.4f over %d classes" % (int(tps[0].sum()), float(ap.mean()), len(cls)))

    # ---- tiny hand-checkable nms: integer coordinates, IoU values 1/3 (2 of 6 vs union 6... ) spelled out
    boxes = np.array([[0, 0, 4, 4],        # A score .9
                      [0, 0, 4, 3],        # B IoU(A,B) = 12/16 = .75      -> dropped at .5
                      [2, 2, 6, 6],        # C IoU(A,C) = 4/28 = .143      -> kept
                      [2, 2, 6, 5],        # D IoU(C,D) = 12/16 = .75      -> dropped (by C)
                      [10, 10, 12, 12],    # E disjoint                    -> kept
                      [0, 0, 4, 2]],       # F IoU(A,F) = 8/16 = .5 exactly -> kept (strict >)
                     dtype=np.float32)
    scores = np.array([0.9, 0.8, 0.7, 0.6, 0.5, 0.4], dtype=np.float32)
    out["tiny/boxes"], out["tiny/scores"] = boxes, scores
    out["tiny/keep_0.5"] = onp.nms(boxes, scores, 0.5)
    assert out["tiny/keep_0.5"].tolist() == [0, 2, 4, 5]
    np.savez_compressed(os.path.join(os.path.dirname(os.path.abspath(__file__)), "postprocess.npz"), **out)
    print("postprocess: %d arrays" % len(out))


if __name__ == "__main__":
    main()
