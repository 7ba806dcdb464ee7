"""Shared by tests/golden/make_golden_proposals.py (reference side) and tests/test_proposals.py (this repo's side):
synthetic per-SVG graph dicts, and the canonical form in which two `_get_proposal` outputs are compared.

Canonical form.  `_get_proposal` returns concatenated arrays whose proposal order is the iteration order of a Python
set (graph_dict3.py:557).  `records()` cuts the 14-tuple back into per-proposal records through the proposal tree
(every proposal is the root or a child of exactly one tree, :756-781); `canonical()` sorts the records by
(box, node positions, local edge list) — a total order on distinct proposals."""
import numpy as np

CASES = {
    "floorplan_like": dict(seed=11, n_cc=4, pts=(9, 16), lattice=5, step=10, n_classes=17, control=0, extra_edges=4),
    "diagram_like": dict(seed=12, n_cc=6, pts=(5, 9), lattice=4, step=5, n_classes=22, control=3, extra_edges=2),
    "dense_lattice": dict(seed=13, n_cc=2, pts=(20, 28), lattice=6, step=10, n_classes=17, control=5, extra_edges=10),
}


# the timed case of bench.py's `proposals` leg and of tests/golden/time_reference.py (the reference's own _get_proposal on
# the same dict): nine dense components, 877 proposals / 9.8 k proposal nodes — the size of a Floorplans drawing
TIMING_CASE = dict(seed=31, n_cc=9, pts=(22, 30), lattice=6, step=10, n_classes=17, control=4, extra_edges=8)
# ... and of the `nms` leg: 625 boxes x 16 classes = 10 000 candidates at conf_thres 0 (train.py:448)
NMS_TIMING = dict(seed=17, n=625, nc=16)

# seeds of Python's `random` and numpy's global generator for the do_mixup runs (reference side and this repo's side)
MIXUP_SEEDS = {"diagram_like": 202, "floorplan_mini": 101}
MIXUP_CASES = {
    "diagram_like": CASES["diagram_like"],
    "floorplan_mini": dict(seed=21, n_cc=3, pts=(7, 10), lattice=4, step=10, n_classes=17, control=2, extra_edges=3),
}


def synth_graph_dict(seed, n_cc, pts, lattice, step, n_classes, control, extra_edges):
    """Components on disjoint patches of the unit square; points sit on a jittered-free lattice (many repeated x / y
    values: the distinct-value grid is much smaller than the point count), edges = a random spanning path plus extra
    random and duplicated pairs, a few control points (dropped by _get_proposal) mixed into the node list."""
    rng = np.random.default_rng(seed)
    pos, is_control, cc, edges, sedges = [], [], [], [], []
    gt_bbox, gt_labels = [], []
    for c in range(n_cc):
        n = int(rng.integers(pts[0], pts[1] + 1))
        ox, oy = (c % 3) * 0.33 + 0.02, (c // 3) * 0.33 + 0.02
        cells = rng.choice(lattice * lattice, size=n, replace=False)
        p = np.stack([ox + (cells % lattice) * (0.28 / (lattice - 1)), oy + (cells // lattice) * (0.28 / (lattice - 1))], 1)
        if len(set(p[:, 0])) < 2 or len(set(p[:, 1])) < 2:
            p[0] = [ox, oy]
            p[1] = [ox + 0.28, oy + 0.28]
        ids = []
        for q in p:
            while control and rng.random() < 0.15:
                pos.append(rng.random(2)); is_control.append(1); control -= 1
            ids.append(len(pos)); pos.append(q); is_control.append(0)
        cc.append(ids)
        order = rng.permutation(n)
        for a, b in zip(order[:-1], order[1:]):
            edges.append((ids[a], ids[b]) if rng.random() < 0.5 else (ids[b], ids[a]))
        for _ in range(extra_edges):
            a, b = rng.choice(n, size=2, replace=False)
            edges.append((ids[a], ids[b]))
        edges.append(edges[-1])                                # a parallel duplicate
        for _ in range(3):
            a, b = rng.choice(n, size=2, replace=False)
            sedges.append((ids[a], ids[b]))
        gt_bbox.append([p[:, 0].min() - 0.01, p[:, 1].min() - 0.01, p[:, 0].max() + 0.01, p[:, 1].max() + 0.01])
        gt_labels.append(int(rng.integers(0, n_classes - 1)))
        sub = p[: max(3, n // 2)]
        gt_bbox.append([sub[:, 0].min(), sub[:, 1].min(), sub[:, 0].max() + 1e-3, sub[:, 1].max() + 1e-3])
        gt_labels.append(int(rng.integers(0, n_classes - 1)))
    pos = np.array(pos, dtype=np.float64)
    E, S = len(edges), len(sedges)
    gd = {"cc": cc, "pos": {"spatial": pos},
          "edge": {"shape": np.array(edges, dtype=np.int64), "super": np.array(sedges, dtype=np.int64)},
          "edge_attr": {"shape": rng.standard_normal((E, 6)), "super": rng.standard_normal((S, 6))},
          "attr": {"is_super": np.zeros((pos.shape[0], 1)), "is_control": np.array(is_control, dtype=np.int64)[:, None]},
          "img_width": 1000, "img_height": 1000}
    return gd, np.array(gt_bbox), np.array(gt_labels), step, n_classes


def records(res):
    (pos, is_super, is_control, edge, edge_super, e_attr, e_attr_super, labels, bbox_idx, bbox, bbox_targets, stat_feats,
     has_obj, roots) = res
    recs = []
    for ci, root in enumerate(roots):
        for role, t in [("root", root)] + [("child", ch) for ch in root.children]:
            p0, p1 = t.value["idx_pos"]
            e0, e1 = t.value["idx_edge"]
            s0, s1 = t.value["idx_edge_super"]
            b = int(t.value["idx_bbox"])
            assert (np.asarray(bbox_idx[p0:p1]) == b).all()
            recs.append({"cc": ci, "role": role, "pos": np.asarray(pos[p0:p1], dtype=np.float64),
                         "is_super": np.asarray(is_super[p0:p1], dtype=np.float64),
                         "edge": np.asarray(edge[e0:e1], dtype=np.int64) - p0,
                         "e_attr": np.asarray(e_attr[e0:e1], dtype=np.float64),
                         "edge_super": np.asarray(edge_super[s0:s1], dtype=np.int64).reshape(-1, 2) - p0,
                         "e_attr_super": np.asarray(e_attr_super[s0:s1], dtype=np.float64),
                         "label": int(labels[b]), "has_obj": int(has_obj[b]),
                         "bbox": np.asarray(bbox[b], dtype=np.float64),
                         "bbox_target": np.asarray(bbox_targets[b], dtype=np.float64),
                         "stat": np.asarray(stat_feats[b], dtype=np.float64)})
    assert len(recs) == len(labels) == np.asarray(bbox).shape[0]
    return recs


def canonical(recs):
    return sorted(recs, key=lambda r: (r["cc"], tuple(r["bbox"]), r["pos"].tobytes(), r["edge"].tobytes()))


_KEYS2 = ["pos", "is_super", "edge", "e_attr", "edge_super", "e_attr_super"]


def pack_records(out, name, recs, res):
    for k in _KEYS2:
        width = max([r[k].shape[1] for r in recs if r[k].ndim == 2 and r[k].shape[0]] + [2 if "edge" == k[:4] else 1])
        out["%s/rec/%s" % (name, k)] = np.concatenate([r[k].reshape(r[k].shape[0], width) for r in recs], 0)
        out["%s/rec/%s_ptr" % (name, k)] = np.cumsum([0] + [r[k].shape[0] for r in recs])
    for k in ("label", "has_obj", "cc"):
        out["%s/rec/%s" % (name, k)] = np.array([r[k] for r in recs], dtype=np.int64)
    out["%s/rec/is_root" % name] = np.array([r["role"] == "root" for r in recs])
    for k in ("bbox", "bbox_target", "stat"):
        out["%s/rec/%s" % (name, k)] = np.stack([r[k] for r in recs])
    out["%s/is_control_out" % name] = np.asarray(res[2])


def unpack_records(z, name):
    n = z["%s/rec/label" % name].shape[0]
    recs = []
    for i in range(n):
        r = {}
        for k in _KEYS2:
            p = z["%s/rec/%s_ptr" % (name, k)]
            r[k] = z["%s/rec/%s" % (name, k)][p[i]:p[i + 1]]
        for k in ("label", "has_obj", "cc"):
            r[k] = int(z["%s/rec/%s" % (name, k)][i])
        r["role"] = "root" if bool(z["%s/rec/is_root" % name][i]) else "child"
        for k in ("bbox", "bbox_target", "stat"):
            r[k] = z["%s/rec/%s" % (name, k)][i]
        recs.append(r)
    return recs


def pack_inputs(out, name, gd, gt_bbox, gt_labels, step, n_classes):
    out["%s/in/pos" % name] = gd["pos"]["spatial"]
    out["%s/in/edge" % name] = gd["edge"]["shape"]
    out["%s/in/edge_super" % name] = gd["edge"]["super"]
    out["%s/in/e_attr" % name] = gd["edge_attr"]["shape"]
    out["%s/in/e_attr_super" % name] = gd["edge_attr"]["super"]
    out["%s/in/is_super" % name] = gd["attr"]["is_super"]
    out["%s/in/is_control" % name] = gd["attr"]["is_control"]
    out["%s/in/cc_ptr" % name] = np.cumsum([0] + [len(c) for c in gd["cc"]])
    out["%s/in/cc_idx" % name] = np.concatenate([np.asarray(c, dtype=np.int64) for c in gd["cc"]])
    out["%s/in/gt_bbox" % name] = gt_bbox
    out["%s/in/gt_labels" % name] = gt_labels
    out["%s/in/step" % name] = np.int64(step)
    out["%s/in/n_classes" % name] = np.int64(n_classes)


def unpack_inputs(z, name):
    ptr, idx = z["%s/in/cc_ptr" % name], z["%s/in/cc_idx" % name]
    gd = {"cc": [[int(v) for v in idx[ptr[i]:ptr[i + 1]]] for i in range(len(ptr) - 1)],
          "pos": {"spatial": z["%s/in/pos" % name]},
          "edge": {"shape": z["%s/in/edge" % name], "super": z["%s/in/edge_super" % name]},
          "edge_attr": {"shape": z["%s/in/e_attr" % name], "super": z["%s/in/e_attr_super" % name]},
          "attr": {"is_super": z["%s/in/is_super" % name], "is_control": z["%s/in/is_control" % name]},
          "img_width": 1000, "img_height": 1000}
    return gd, z["%s/in/gt_bbox" % name], z["%s/in/gt_labels" % name], int(z["%s/in/step" % name]), \
        int(z["%s/in/n_classes" % name])
