"""Times the REFERENCE'S OWN host functions of SURVEY.md section 8 f.3 / f.4 in the build container and writes
tests/golden/reference_timings.json — the figures `bench.py`'s `proposals` / `nms` legs quote next to this repo's times
(the reference never travels to the GPU box; only these numbers do).  DEV-TIME ONLY: needs /root/reference.
Run from the repo root:  python tests/golden/time_reference.py

  _get_proposal          Datasets/graph_dict3.py:309-789, compiled from its source text (make_golden_proposals.reference_method)
                         on the synthetic per-SVG dict `proposals_util.TIMING_CASE` (Floorplans-sized: 877 proposals, 9.8 k proposal nodes)
  non_max_suppression    cad_recognition/train.py:34-121, compiled from its source text (make_golden_post.reference_nms_function)
                         with torchvision.ops.nms restated by oracle_np.nms, on 10 000 candidates (625 boxes x 16 classes,
                         conf_thres 0: the evaluation loop's call, train.py:448)
Both single-threaded Python, as the reference runs them (one DataLoader worker / the evaluation loop)."""
import json
import os
import sys
import time
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tests"))
sys.path.insert(0, HERE)
import proposals_util as pu                      # noqa: E402
import make_golden_proposals as mgp              # noqa: E402
import make_golden_post as mgpost                # noqa: E402


def best_of(fn, reps):
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter()
        out = fn()
        ts.append(time.perf_counter() - t0)
    ts.sort()
    return ts[len(ts) // 2], out


def main():
    torch.set_num_threads(1)
    get, _ = mgp.reference_method()
    gd, gt_bbox, gt_labels, step, n_classes = pu.synth_graph_dict(**pu.TIMING_CASE)
    me = types.SimpleNamespace(do_mixup=False, n_classes=n_classes, normalize_bbox=True)
    t_prop, res = best_of(lambda: get(me, gd, gt_bbox, gt_labels, bbox_sampling_step=step), 3)
    n_prop = int(np.asarray(res[9]).shape[0])
    ref_nms = mgpost.reference_nms_function()
    pred = mgpost.synth_prediction(np.random.default_rng(pu.NMS_TIMING["seed"]), pu.NMS_TIMING["n"], pu.NMS_TIMING["nc"])
    t_nms, det = best_of(lambda: ref_nms(torch.from_numpy(pred.copy()), conf_thres=0.0, iou_thres=0.5), 3)
    out = {
        "host": {"cpus": os.cpu_count(), "threads_used": 1, "note": "build container; single-threaded Python as the reference runs it"},
        "get_proposal": {"case": pu.TIMING_CASE, "proposals": n_prop, "nodes": int(res[0].shape[0]), "edges": int(res[3].shape[0]),
                         "seconds_per_svg": t_prop, "reference": "Datasets/graph_dict3.py:309-789"},
        "non_max_suppression": {"case": pu.NMS_TIMING, "candidates": pu.NMS_TIMING["n"] * pu.NMS_TIMING["nc"],
                                "detections": int(det[0].shape[0]), "seconds_per_call": t_nms,
                                "reference": "cad_recognition/train.py:34-121 (torchvision.ops.nms restated by oracle_np.nms)"},
    }
    with open(os.path.join(HERE, "reference_timings.json"), "w") as f:
        json.dump(out, f, indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
