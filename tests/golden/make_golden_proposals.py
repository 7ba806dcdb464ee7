"""Generates tests/golden/proposals.npz from the REFERENCE'S OWN `SESYDFloorPlan._get_proposal`
(/root/reference/Datasets/graph_dict3.py:309-789) — run in the build container only; the fixture travels.

The method (and `idxTree`, :24-27) is compiled from the reference's source text where it lies: graph_dict3.py cannot be
imported (torch_geometric, cv2, svgpathtools absent).  Names it uses resolve to the real things: `np`, `random`, and
`bbox_iou_ios_cpu` / `intersect_bb_idx` imported from the reference's own utils/det_util.py.  It is bound to a bare
object carrying the three attributes it reads (do_mixup, n_classes, normalize_bbox = True); for the mixup cases the
reference's own `mixup` (:791-907) is compiled the same way and bound to that object.

Inputs: three synthetic per-SVG graph dicts in the on-disk schema of utils/svg_utils/build_graph_bbox.py:351-370
(lattice-like points with repeated coordinates, control points that get dropped, parallel edges, super edges) + ground
truth boxes.  Stored: the inputs, and the reference's outputs as CANONICALLY ORDERED per-proposal records (the reference
emits proposals in Python-set order, :557) — see `records()` / `canonical()`; tests/test_proposals.py applies the same
canonicalisation to this repo's output.
"""
import ast
import os
import random
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tests"))
import proposals_util as pu  # noqa: E402


def reference_method():
    sys.path.insert(0, REF)
    from utils.det_util import bbox_iou_ios_cpu, intersect_bb_idx
    path = os.path.join(REF, "Datasets", "graph_dict3.py")
    src = open(path).read()
    tree = ast.parse(src)
    tree_cls = next(n for n in tree.body if isinstance(n, ast.ClassDef) and n.name == "idxTree")
    ds_cls = next(n for n in tree.body if isinstance(n, ast.ClassDef) and n.name == "SESYDFloorPlan")
    fn = next(n for n in ds_cls.body if isinstance(n, ast.FunctionDef) and n.name == "_get_proposal")
    assert (fn.lineno, fn.end_lineno) == (309, 789), (fn.lineno, fn.end_lineno)
    mx = next(n for n in ds_cls.body if isinstance(n, ast.FunctionDef) and n.name == "mixup")
    assert (mx.lineno, mx.end_lineno) == (791, 907), (mx.lineno, mx.end_lineno)
    ns = {"np": np, "random": random, "bbox_iou_ios_cpu": bbox_iou_ios_cpu, "intersect_bb_idx": intersect_bb_idx}
    exec(compile(ast.Module(body=[tree_cls, fn, mx], type_ignores=[]), path, "exec"), ns)
    return ns["_get_proposal"], ns["mixup"]


def main():
    get, mix = reference_method()
    out = {}
    for name, kw in pu.CASES.items():
        gd, gt_bbox, gt_labels, step, n_classes = pu.synth_graph_dict(**kw)
        me = types.SimpleNamespace(do_mixup=False, n_classes=n_classes, normalize_bbox=True)
        res = get(me, gd, gt_bbox, gt_labels, bbox_sampling_step=step)
        recs = pu.canonical(pu.records(res))
        pu.pack_inputs(out, name, gd, gt_bbox, gt_labels, step, n_classes)
        pu.pack_records(out, name, recs, res)
        print(name, "proposals:", len(recs), "nodes:", res[0].shape[0], "edges:", res[3].shape[0],
              "components:", len(res[13]))
    # the mixup augmentation (:354-355 -> :791-907): the same cases with do_mixup on, Python's and numpy's GLOBAL generators
    # seeded the way the test seeds them (pu.MIXUP_SEEDS); ground truth widened by one box over the frame the synthetic
    # components are put into ([0, 2.1]^2), as a real annotation set covers its drawing
    for name in pu.MIXUP_SEEDS:
        gd, gt_bbox, gt_labels, step, n_classes = pu.synth_graph_dict(**pu.MIXUP_CASES[name])
        pu.pack_inputs(out, name + "_mixup", gd, gt_bbox, gt_labels, step, n_classes)
        me = types.SimpleNamespace(do_mixup=True, n_classes=n_classes, normalize_bbox=True)
        me.mixup = types.MethodType(mix, me)
        seed = pu.MIXUP_SEEDS[name]
        random.seed(seed)
        np.random.seed(seed)
        res = get(me, gd, gt_bbox, gt_labels, bbox_sampling_step=step)
        recs = pu.canonical(pu.records(res))
        pu.pack_records(out, name + "_mixup", recs, res)
        print(name + "_mixup", "proposals:", len(recs), "nodes:", res[0].shape[0], "edges:", res[3].shape[0],
              "components:", len(res[13]))
    out["provenance"] = np.array("outputs of the reference's own SESYDFloorPlan._get_proposal "
                                 "(Datasets/graph_dict3.py:309-789, compiled from its source text), canonically ordered")
    np.savez_compressed(os.path.join(HERE, "proposals.npz"), **out)


if __name__ == "__main__":
    main()
