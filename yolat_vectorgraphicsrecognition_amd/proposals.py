"""Box-proposal generation — the dataset-side producer of the hot path's input.

``get_proposal(graph_dict, gt_bbox, gt_labels, bbox_sampling_step, n_classes)`` mirrors
``SESYDFloorPlan._get_proposal`` (/root/reference/Datasets/graph_dict3.py:309-789, with the optional random ``mixup``
augmentation of :354-355, :791-907): same arguments, same 14-tuple

    pos, is_super, is_control, edge, edge_super, e_attr, e_attr_super, labels, bbox_idx, bbox, bbox_targets,
    stat_feats, has_obj, roots

The combinatorial core — distinct-coordinate grid, sampling-grid windows, point set per window, de-duplication, edge
pick-up, the rejection tests — is native host code behind the C ABI (csrc/proposals.hip, ``yolat_proposals_build``);
what is left here is per-proposal float64 numpy arithmetic written exactly as the reference writes it (IoU labels
:624-640, angle statistics :644-705, box normalisation :714-722) and the bookkeeping of the proposal tree (:756-781).

ORDER.  The reference walks ``list(set(sub_clusters))`` (:557): CPython hash order, not reproducible across runs of
different interpreters.  Here the proposals of a component come in lexicographic order of their sorted node-id
tuples.  Everything per proposal is identical; ``bbox_idx``, the edge offsets and the root (first largest-area
proposal) follow that order.  tests/test_proposals.py compares against fixtures produced by the reference's own method
as canonically ordered sets.
"""
import ctypes

import numpy as np

from ._lib import lib, check, YolatLibraryError
from .data import idxTree


def bbox_iou_ios_cpu(box1, box2):
    """utils/det_util.py:311-341 (x1y1x2y2)."""
    b1_x1, b1_y1, b1_x2, b1_y2 = box1[:, 0], box1[:, 1], box1[:, 2], box1[:, 3]
    b2_x1, b2_y1, b2_x2, b2_y2 = box2[:, 0], box2[:, 1], box2[:, 2], box2[:, 3]
    ix1, iy1 = np.maximum(b1_x1, b2_x1), np.maximum(b1_y1, b2_y1)
    ix2, iy2 = np.minimum(b1_x2, b2_x2), np.minimum(b1_y2, b2_y2)
    inter = np.maximum(ix2 - ix1, 0) * np.maximum(iy2 - iy1, 0)
    a1 = (b1_x2 - b1_x1) * (b1_y2 - b1_y1)
    a2 = (b2_x2 - b2_x1) * (b2_y2 - b2_y1)
    return inter / (a1 + a2 - inter + 1e-16), inter / a2


def intersect_bb_idx(box1, box2):
    """utils/det_util.py:343-362."""
    ix1, iy1 = np.maximum(box1[:, 0], box2[:, 0]), np.maximum(box1[:, 1], box2[:, 1])
    ix2, iy2 = np.minimum(box1[:, 2], box2[:, 2]), np.minimum(box1[:, 3], box2[:, 3])
    return np.where((ix2 > ix1) & (iy2 > iy1))[0]


def _i64(a):
    return np.ascontiguousarray(a, dtype=np.int64)


def _ptr(a):
    return a.ctypes.data_as(ctypes.c_void_p) if a is not None and a.size else None


def proposal_windows(pos, cc, edge, edge_super, bbox_sampling_step, keep_handle=False):
    """keep_handle: the dict also carries the native handle ("handle", to be freed by the caller with
    lib.yolat_proposals_free) and the contiguous inputs it was built from — what yolat_proposals_assemble works on.
    The native core on already renumbered inputs.  pos [n,2] float64; cc: list of lists of node ids;
    edge / edge_super [.,2] int64.  Returns a dict of int64 CSR arrays (node / edge / super-edge members of every
    proposal), `cc_of`, `bbox` [count,4] float64 and the per-component window statistics."""
    pos = np.ascontiguousarray(pos, dtype=np.float64)
    edge = _i64(np.asarray(edge).reshape(-1, 2))
    edge_super = _i64(np.asarray(edge_super).reshape(-1, 2))
    cc_ptr = np.zeros(len(cc) + 1, dtype=np.int64)
    np.cumsum([len(c) for c in cc], out=cc_ptr[1:])
    cc_idx = _i64(np.concatenate([np.asarray(c, dtype=np.int64) for c in cc])) if len(cc) else np.zeros(0, np.int64)
    handle = ctypes.c_void_p()
    rc = lib.yolat_proposals_build(_ptr(pos), pos.shape[0], _ptr(cc_ptr), _ptr(cc_idx), len(cc), _ptr(edge),
                                   edge.shape[0], _ptr(edge_super), edge_super.shape[0], float(bbox_sampling_step),
                                   ctypes.byref(handle))
    if rc == -2:        # YOLAT_E_UNSUPPORTED: numpy.arange(min, max, 0) in the reference
        raise ZeroDivisionError("a connected component has zero width or height (graph_dict3.py:462-463)")
    check(rc, "yolat_proposals_build")
    try:
        n = int(lib.yolat_proposals_count(handle))
        tot = [int(lib.yolat_proposals_total(handle, w)) for w in range(3)]
        out = {"node_ptr": np.zeros(n + 1, np.int64), "node_idx": np.zeros(tot[0], np.int64),
               "edge_ptr": np.zeros(n + 1, np.int64), "edge_idx": np.zeros(tot[1], np.int64),
               "sedge_ptr": np.zeros(n + 1, np.int64), "sedge_idx": np.zeros(tot[2], np.int64),
               "cc_of": np.zeros(n, np.int64), "bbox": np.zeros((n, 4), np.float64),
               "windows_per_cc": np.zeros(len(cc), np.int64), "distinct_per_cc": np.zeros(len(cc), np.int64)}
        check(lib.yolat_proposals_get(handle, _ptr(out["node_ptr"]), _ptr(out["node_idx"]), _ptr(out["edge_ptr"]),
                                      _ptr(out["edge_idx"]), _ptr(out["sedge_ptr"]), _ptr(out["sedge_idx"]),
                                      _ptr(out["cc_of"]), _ptr(out["bbox"])), "yolat_proposals_get")
        check(lib.yolat_proposals_window_counts(handle, _ptr(out["windows_per_cc"]), _ptr(out["distinct_per_cc"])),
              "yolat_proposals_window_counts")
    except Exception:
        lib.yolat_proposals_free(handle)
        raise
    if keep_handle:
        out["handle"], out["_pos"], out["_edge"], out["_edge_super"] = handle, pos, edge, edge_super
    else:
        lib.yolat_proposals_free(handle)
    return out


def mixup(cc, pos, edge, edge_super, e_attr, e_attr_super, is_super):
    """The reference's `mixup` augmentation, graph_dict3.py:791-907 (called at :354-355 when the dataset is built with
    do_mixup): for EVERY component i a partner j is drawn (Python's global `random`), both are brought into the unit square
    (same scale on both axes: the larger extent), the partner is pushed to the right of / below the first by a random offset
    (numpy's global RNG), and the pair becomes one NEW component appended behind the existing nodes — its shape edges are
    the two components' edges renumbered, its super edges theirs plus the complete bipartite set between the two (attributes:
    zeros, 6 columns).  The draws are made with the same calls in the same order as the reference makes them (per component:
    random.choice(range(n)), random.choice([True, False]), two np.random.random()), so that a caller who seeds the two
    global generators the way a reference run does gets the reference's arrays bit for bit (tests/test_proposals.py).
    Like the reference, a component without a shape edge or without a super edge cannot be mixed (np.stack of nothing):
    ValueError."""
    import random
    n_nodes = pos.shape[0]
    owner = np.zeros(n_nodes, dtype=np.int64)
    for ci, cluster in enumerate(cc):
        owner[np.asarray(cluster, dtype=np.int64)] = ci
    edge = np.asarray(edge)
    edge_super = np.asarray(edge_super)
    e_owner = owner[edge[:, 0]] if edge.size else np.zeros(0, np.int64)
    s_owner = owner[edge_super[:, 0]] if edge_super.size else np.zeros(0, np.int64)
    edges_of = [np.nonzero(e_owner == ci)[0] for ci in range(len(cc))]          # original order inside a component
    sedges_of = [np.nonzero(s_owner == ci)[0] for ci in range(len(cc))]

    def unit(p):
        lo = np.array([p[:, 0].min(0), p[:, 1].min(0)])
        ex, ey = p[:, 0].max(0) - lo[0], p[:, 1].max(0) - lo[1]
        d = ex if ex > ey else ey
        return (p - lo) / np.array([d, d])

    def renumber(rows, old_ids, new_ids):
        table = {}
        for o, nw in zip(old_ids, new_ids):
            table[o] = nw
        return np.array([[table[a], table[b]] for a, b in rows])

    offset = n_nodes
    add_cc, add_pos, add_edge, add_sedge, add_attr, add_sattr, add_super = [], [], [], [], [], [], []
    for ci in range(len(cc)):
        cj = random.choice(range(len(cc)))
        first, second = cc[ci], cc[cj]
        if edges_of[ci].size == 0 or edges_of[cj].size == 0 or sedges_of[ci].size == 0 or sedges_of[cj].size == 0:
            raise ValueError("need at least one array to stack")        # np.stack([]) in the reference, :838-845
        p0, p1 = unit(pos[first]), unit(pos[second])
        if random.choice([True, False]):
            tx = 1 + np.random.random() * 0.1
            ty = np.random.random()
        else:
            tx = np.random.random()
            ty = 1 + 0.1 * np.random.random()
        p1[:, 0] += tx
        p1[:, 1] += ty
        ids0 = offset + np.arange(len(first))
        ids1 = offset + len(first) + np.arange(len(second))
        cross = [[a, b] for a in ids0 for b in ids1]
        e0, e1 = renumber(edge[edges_of[ci]], first, ids0), renumber(edge[edges_of[cj]], second, ids1)
        s0, s1 = renumber(edge_super[sedges_of[ci]], first, ids0), renumber(edge_super[sedges_of[cj]], second, ids1)
        add_pos.append(np.concatenate([p0, p1], axis=0))
        add_super.append(np.concatenate([is_super[first], is_super[second]], axis=0))
        add_cc.append(list(ids0) + list(ids1))
        add_edge.append(np.concatenate([e0, e1], axis=0))
        add_sedge.append(np.concatenate([s0, s1, cross], axis=0))
        add_attr.append(np.concatenate([e_attr[edges_of[ci]], e_attr[edges_of[cj]]], axis=0))
        add_sattr.append(np.zeros((s0.shape[0] + s1.shape[0] + len(cross), 6)))
        offset += len(first) + len(second)
    return (cc + add_cc, np.concatenate([pos] + add_pos, axis=0), np.concatenate([edge] + add_edge, axis=0),
            np.concatenate([edge_super] + add_sedge, axis=0), np.concatenate([e_attr] + add_attr, axis=0),
            np.concatenate([e_attr_super] + add_sattr, axis=0), np.concatenate([is_super] + add_super, axis=0))


def _assemble_native(w, cc, pos, is_super, e_attr, e_attr_super, gt_bbox, gt_labels, n_classes, normalize_bbox, stat_feats):
    """graph_dict3.py:573-789 on the native member lists: the per-proposal assembly is ONE call (yolat_proposals_assemble:
    local re-indexing, IoU / IoS labels, the 13 statistics, normalisation); the proposal tree (:756-781) is built here."""
    count = int(w["cc_of"].shape[0])
    pos_c, edge_c, sedge_c = w["_pos"], w["_edge"], w["_edge_super"]
    gt_bbox = np.ascontiguousarray(np.asarray(gt_bbox, dtype=np.float64).reshape(-1, 4))
    gt_lab = _i64(np.asarray(gt_labels).reshape(-1))
    # :573-577 — every component must overlap a ground-truth box
    valid_ptr, valid_idx = [0], []
    for cluster in cc:
        pc = pos[cluster, :]
        bbox_cc = np.array([pc[:, 0].min(0), pc[:, 1].min(0), pc[:, 0].max(0), pc[:, 1].max(0)])[None, :]
        valid = intersect_bb_idx(bbox_cc, gt_bbox)
        if valid.shape[0] == 0:
            raise SystemExit("cc has no intersect gt bbox")
        valid_idx.append(valid)
        valid_ptr.append(valid_ptr[-1] + valid.shape[0])
    valid_ptr = _i64(np.asarray(valid_ptr))
    valid_idx = _i64(np.concatenate(valid_idx)) if valid_idx else np.zeros(0, np.int64)
    is_super = np.ascontiguousarray(np.asarray(is_super, dtype=np.float64).reshape(pos_c.shape[0], -1))
    e_attr = np.ascontiguousarray(np.asarray(e_attr, dtype=np.float64).reshape(edge_c.shape[0], -1)) if edge_c.shape[0] \
        else np.zeros((0, np.asarray(e_attr).shape[-1] if np.asarray(e_attr).ndim == 2 else 0))
    e_attr_super = np.ascontiguousarray(np.asarray(e_attr_super, dtype=np.float64).reshape(sedge_c.shape[0], -1)) \
        if sedge_c.shape[0] else np.zeros((0, np.asarray(e_attr_super).shape[-1] if np.asarray(e_attr_super).ndim == 2 else 0))
    sw, aw, asw = is_super.shape[1], e_attr.shape[1], e_attr_super.shape[1]
    nn_, ne_, ns_ = int(w["node_ptr"][-1]), int(w["edge_ptr"][-1]), int(w["sedge_ptr"][-1])
    new_pos = np.zeros((nn_, 2)); new_is_super = np.zeros((nn_, sw))
    new_edge = np.zeros((ne_, 2), np.int64); new_e_attr = np.zeros((ne_, aw))
    new_edge_super = np.zeros((ns_, 2), np.int64); new_e_attr_super = np.zeros((ns_, asw))
    labels = np.zeros(count, np.int64); has_obj = np.zeros(count, np.int64); bbox_idx = np.zeros(nn_, np.int64)
    bbox_targets = np.zeros((count, 4)); stat = np.zeros((count, 13))
    rc = lib.yolat_proposals_assemble(w["handle"], _ptr(pos_c), _ptr(is_super), sw, _ptr(edge_c), _ptr(e_attr), aw,
                                      _ptr(sedge_c), _ptr(e_attr_super), asw, _ptr(gt_bbox), _ptr(gt_lab), _ptr(valid_ptr),
                                      _ptr(valid_idx), int(n_classes), 1 if normalize_bbox else 0, 1 if stat_feats else 0,
                                      _ptr(new_pos), _ptr(new_is_super), _ptr(new_edge), _ptr(new_e_attr),
                                      _ptr(new_edge_super), _ptr(new_e_attr_super), _ptr(labels), _ptr(has_obj),
                                      _ptr(bbox_idx), _ptr(bbox_targets), _ptr(stat))
    check(rc, "yolat_proposals_assemble")
    # :756-781 — per component: the largest-area proposal is the root, the others its children
    roots = []
    bb = np.array(w["bbox"]).reshape(-1, 4)
    sp, se, ss = w["node_ptr"], w["edge_ptr"], w["sedge_ptr"]
    for c in range(len(cc)):
        members = np.where(w["cc_of"] == c)[0]
        if members.size == 0:
            raise ValueError("component %d produced no proposal (np.argmax of an empty array in the reference)" % c)
        area = (bb[members, 2] - bb[members, 0]) * (bb[members, 3] - bb[members, 1])
        top = int(np.argmax(area))
        nodes = []
        for m in members:
            t = idxTree()
            t.value["idx_pos"] = (int(sp[m]), int(sp[m + 1]))
            t.value["idx_edge"] = (int(se[m]), int(se[m + 1]))
            t.value["idx_edge_super"] = (int(ss[m]), int(ss[m + 1]))
            t.value["idx_bbox"] = int(m)
            nodes.append(t)
        root = nodes[top]
        root.children = [t for i, t in enumerate(nodes) if i != top]
        roots.append(root)
    return (new_pos, new_is_super, np.zeros((new_pos.shape[0], 1)), new_edge, new_edge_super, new_e_attr, new_e_attr_super,
            labels.tolist(), bbox_idx, bb, bbox_targets, stat, has_obj.tolist(), roots)


def get_proposal(graph_dict, gt_bbox, gt_labels, bbox_sampling_step=-1, n_classes=17, normalize_bbox=True, do_mixup=False,
                 stat_feats=True, native=True):
    """graph_dict3.py:309-789; do_mixup: the augmentation of :354-355 / :791-907 (`mixup` above; off in the published
    recipe, README.md:47,52).  stat_feats=False skips the 13 per-proposal statistics of :644-705 (an O(sum deg^2) Python
    loop over neighbour pairs) and returns zeros in their place: the model never reads them (arch:87 `dim_stat = 0`, :112
    copies them to the device unused); the default keeps the reference's output.  graph_dict: the pickled per-SVG dict of
    utils/svg_utils/build_graph_bbox.py:351-370 ('cc', 'pos'/'spatial', 'edge'/{'shape','super'},
    'edge_attr'/{'shape','super'}, 'attr'/{'is_super','is_control'}, 'img_width', 'img_height')."""
    cc = graph_dict["cc"]
    pos = np.asarray(graph_dict["pos"]["spatial"])
    edge = np.asarray(graph_dict["edge"]["shape"])
    edge_super = np.asarray(graph_dict["edge"]["super"])
    e_attr = np.asarray(graph_dict["edge_attr"]["shape"])
    e_attr_super = np.asarray(graph_dict["edge_attr"]["super"])
    is_super = np.asarray(graph_dict["attr"]["is_super"])
    is_control = np.asarray(graph_dict["attr"]["is_control"])
    gt_bbox = np.asarray(gt_bbox)
    # :329-356 — control points are dropped, everything is renumbered
    not_control = (is_control == 0)[:, 0]
    o2n = np.cumsum(not_control) - 1
    if edge.size and (~not_control[edge.reshape(-1)]).any():
        raise KeyError("an edge references a control point")          # o2n[e[0]] in the reference
    # (the reference looks every id up in its o2n dict, graph_dict3.py:337-356: a control-point id anywhere raises KeyError)
    if edge_super.size and (~not_control[edge_super.reshape(-1)]).any():
        raise KeyError("a super edge references a control point")
    for cluster in cc:
        if len(cluster) and (~not_control[np.asarray(cluster, dtype=np.int64)]).any():
            raise KeyError("a connected component contains a control point")
    edge = o2n[edge.reshape(-1, 2)] if edge.size else np.zeros((0, 2), np.int64)
    edge_super = o2n[edge_super.reshape(-1, 2)] if edge_super.size else np.zeros((0, 2), np.int64)
    cc = [[int(o2n[i]) for i in cluster] for cluster in cc]
    pos = pos[not_control]
    is_super = is_super[not_control]
    if do_mixup:
        cc, pos, edge, edge_super, e_attr, e_attr_super, is_super = mixup(cc, pos, edge, edge_super, e_attr, e_attr_super,
                                                                          is_super)

    w = proposal_windows(pos, cc, edge, edge_super, bbox_sampling_step, keep_handle=native)
    count = w["cc_of"].shape[0]
    if native:
        try:
            return _assemble_native(w, cc, pos, is_super, e_attr, e_attr_super, gt_bbox, gt_labels, n_classes, normalize_bbox,
                                    stat_feats)
        finally:
            lib.yolat_proposals_free(w["handle"])
    # :573-577 — every component must overlap a ground-truth box
    valid_of_cc = []
    for c, cluster in enumerate(cc):
        pc = pos[cluster, :]
        bbox_cc = np.array([pc[:, 0].min(0), pc[:, 1].min(0), pc[:, 0].max(0), pc[:, 1].max(0)])[None, :]
        valid = intersect_bb_idx(bbox_cc, gt_bbox)
        if valid.shape[0] == 0:
            raise SystemExit("cc has no intersect gt bbox")
        valid_of_cc.append(valid)

    new_pos, new_is_super, new_edge, new_edge_super, new_e_attr, new_e_attr_super = [], [], [], [], [], []
    labels, has_objs, bbox_idx, new_bbox, bbox_targets, stat_feats_out = [], [], [], [], [], []
    slice_pos, slice_edge, slice_super = [0], [0], [0]
    offset = 0
    for p in range(count):
        idxs = w["node_idx"][w["node_ptr"][p]:w["node_ptr"][p + 1]]
        eids = w["edge_idx"][w["edge_ptr"][p]:w["edge_ptr"][p + 1]]
        sids = w["sedge_idx"][w["sedge_ptr"][p]:w["sedge_ptr"][p + 1]]
        local = {int(g): i for i, g in enumerate(idxs)}
        pos_bbox = pos[idxs, :]
        edge_bbox = np.array([[local[int(a)] + offset, local[int(b)] + offset] for a, b in edge[eids]])
        e_attr_bbox = e_attr[eids]
        edge_super_bbox = np.array([[local[int(a)] + offset, local[int(b)] + offset] for a, b in edge_super[sids]])
        e_attr_super_bbox = e_attr_super[sids]
        min_x, min_y, max_x, max_y = w["bbox"][p]
        # :624-640
        valid = valid_of_cc[int(w["cc_of"][p])]
        proposal = np.array([min_x, min_y, max_x, max_y])[None, :]
        iou, ios = bbox_iou_ios_cpu(proposal, gt_bbox[valid, :])
        idx_gt = np.argmax(iou)
        if iou[idx_gt] > 0.7:
            label = gt_labels[valid[idx_gt]]
            bbox_target = gt_bbox[valid[idx_gt]][None, :]
        else:
            label = n_classes - 1
            bbox_target = np.zeros((1, 4))
        has_obj = 1 if ios[idx_gt] > 0.7 else 0
        # :644-705 — angle statistics over pairs of neighbours of every node
        k = pos_bbox.shape[0]
        if not stat_feats:
            if normalize_bbox:
                pos_bbox = (pos_bbox - [min_x, min_y]) / [max_x - min_x, max_y - min_y]
            slice_pos.append(slice_pos[-1] + k)
            slice_edge.append(slice_edge[-1] + edge_bbox.shape[0])
            slice_super.append(slice_super[-1] + edge_super_bbox.shape[0])
            new_pos.append(pos_bbox)
            new_is_super.append(is_super[idxs, :])
            new_edge.append(edge_bbox)
            if edge_super_bbox.shape[0] > 0:
                new_edge_super.append(edge_super_bbox)
            new_e_attr.append(e_attr_bbox)
            new_e_attr_super.append(e_attr_super_bbox)
            labels.append(label)
            has_objs.append(has_obj)
            bbox_idx += [p] * k
            offset += k
            new_bbox.append([min_x, min_y, max_x, max_y])
            bbox_targets.append(bbox_target)
            stat_feats_out.append(np.zeros((1, 13)))
            continue
        adj = [set() for _ in range(k)]
        for a, b in edge_bbox:
            adj[a - offset].add(b - offset)
            adj[b - offset].add(a - offset)
        n_less, n_90, n_more, angles = 0, 0, 0, []
        for anchor, neighbors in enumerate(adj):
            neighbors = list(neighbors)
            for i in range(len(neighbors)):
                for j in range(i + 1, len(neighbors)):
                    v0 = pos_bbox[neighbors[i]] - pos_bbox[anchor]
                    v1 = pos_bbox[neighbors[j]] - pos_bbox[anchor]
                    dot = v0[0] * v1[0] + v0[1] * v1[1]
                    if dot <= -1e-2:
                        n_more += 1
                    elif dot >= 1e-2:
                        n_less += 1
                    elif np.abs(dot) < 1e-2:
                        n_90 += 1
                    angles.append(dot)
        assert angles, "the native rejection test keeps only proposals with an angle pair"
        angles = np.array(angles)
        width, height = max_x - min_x, max_y - min_y
        stat_feat = np.array([k, edge_bbox.shape[0], n_90, n_less, n_more, width, height, np.mean(angles),
                              np.max(angles), np.min(angles), np.std(angles), np.mean(e_attr_bbox[:, -1]),
                              np.std(e_attr_bbox[:, -1])])[None, :]
        if normalize_bbox:
            pos_bbox = (pos_bbox - [min_x, min_y]) / [max_x - min_x, max_y - min_y]
        slice_pos.append(slice_pos[-1] + k)
        slice_edge.append(slice_edge[-1] + edge_bbox.shape[0])
        slice_super.append(slice_super[-1] + edge_super_bbox.shape[0])
        new_pos.append(pos_bbox)
        new_is_super.append(is_super[idxs, :])
        new_edge.append(edge_bbox)
        if edge_super_bbox.shape[0] > 0:
            new_edge_super.append(edge_super_bbox)
        new_e_attr.append(e_attr_bbox)
        new_e_attr_super.append(e_attr_super_bbox)
        labels.append(label)
        has_objs.append(has_obj)
        bbox_idx += [p] * k
        offset += k
        new_bbox.append([min_x, min_y, max_x, max_y])
        bbox_targets.append(bbox_target)
        stat_feats_out.append(stat_feat)

    # :756-781 — per component: the largest-area proposal is the root, the others its children
    roots = []
    bb = np.array(new_bbox).reshape(-1, 4)
    for c in range(len(cc)):
        members = np.where(w["cc_of"] == c)[0]
        if members.size == 0:
            raise ValueError("component %d produced no proposal (np.argmax of an empty array in the reference)" % c)
        area = (bb[members, 2] - bb[members, 0]) * (bb[members, 3] - bb[members, 1])
        top = int(np.argmax(area))
        nodes = []
        for m in members:
            t = idxTree()
            t.value["idx_pos"] = (slice_pos[m], slice_pos[m + 1])
            t.value["idx_edge"] = (slice_edge[m], slice_edge[m + 1])
            t.value["idx_edge_super"] = (slice_super[m], slice_super[m + 1])
            t.value["idx_bbox"] = int(m)
            nodes.append(t)
        root = nodes[top]
        root.children = [t for i, t in enumerate(nodes) if i != top]
        roots.append(root)

    pos_out = np.concatenate(new_pos, axis=0)
    return (pos_out, np.concatenate(new_is_super, axis=0), np.zeros((pos_out.shape[0], 1)),
            np.concatenate(new_edge, axis=0), np.concatenate(new_edge_super, axis=0),
            np.concatenate(new_e_attr, axis=0), np.concatenate(new_e_attr_super, axis=0), labels,
            np.array(bbox_idx), bb, np.concatenate(bbox_targets, axis=0), np.concatenate(stat_feats_out, axis=0),
            has_objs, roots)
