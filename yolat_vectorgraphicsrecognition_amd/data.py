"""Host-side batch container, block-diagonal batching and synthetic Bézier-graph generators.

Mirrors what the reference's callers hand to ``SparseCADGCN.forward(data, slices)``:

* ``Data`` — attribute bag standing in for ``torch_geometric.data.Data`` (absent from the image),
  with the tiny surface the reference touches: ``.keys``, ``__cat_dim__``, item get/set
  (cad_recognition/train.py:126-157, architecture3cc_rpn_gp_iter2.py:180).
* ``collate`` / ``fixup_offsets`` — cad_recognition/train.py:123-171 and :238-258.
* ``synth_graph`` — seeded synthetic graphs that honour the structural invariants of
  ``Datasets/graph_dict3.py`` (SURVEY.md §8 d, App. F): nodes grouped by proposal with sorted
  ``bbox_idx`` (:732); edges only inside a proposal, stored once, arbitrary direction (:594-600);
  ``x = [0,0,0,px,py]`` with per-proposal min-max-normalised positions (:714,966-969);
  ``e_attr`` 4-d control-point offsets, exactly 0 for straight segments
  (Datasets/bezier_parser.py:62-71).
"""
from itertools import product

import numpy as np
import torch


class Data(object):
    """Minimal ``torch_geometric.data.Data`` look-alike (attribute bag)."""

    def __init__(self, x=None, pos=None, **kwargs):
        self.x = x
        self.pos = pos
        for k, v in kwargs.items():
            setattr(self, k, v)

    @property
    def keys(self):
        return [k for k in self.__dict__.keys() if not k.startswith("_") and self.__dict__[k] is not None]

    def __getitem__(self, key):
        return getattr(self, key, None)

    def __setitem__(self, key, value):
        setattr(self, key, value)

    def __contains__(self, key):
        return key in self.keys

    def __cat_dim__(self, key, value):
        return 0

    @property
    def num_nodes(self):
        return None if self.x is None else self.x.size(0)

    def to_dict(self):
        return {k: self[k] for k in self.keys}


def collate(data_list):
    """List of per-image ``Data`` -> (batched ``Data``, ``slices``) — train.py:123-171."""
    keys = data_list[0].keys
    data = data_list[0].__class__()
    for key in keys:
        data[key] = []
    slices = {key: [0] for key in keys}
    for item, key in product(data_list, keys):
        v = item[key]
        data[key].append(v)
        if isinstance(v, torch.Tensor) and v.dim() > 0:
            s = slices[key][-1] + v.size(0)
        elif isinstance(v, list):
            s = slices[key][-1] + len(v)
        else:
            s = slices[key][-1] + 1
        slices[key].append(s)
    for key in keys:
        item = data_list[0][key]
        if isinstance(item, torch.Tensor) and len(data_list) > 1:
            data[key] = torch.cat(data[key], dim=0) if item.dim() > 0 else torch.stack(data[key])
        elif isinstance(item, torch.Tensor):
            data[key] = data[key][0]
        elif isinstance(item, (int, float)):
            data[key] = torch.tensor(data[key])
        elif isinstance(item, list):
            flat = []
            for it in data[key]:
                flat += it
            data[key] = flat
        slices[key] = torch.tensor(slices[key], dtype=torch.long)
    return data, slices


def fixup_offsets(data, slices):
    """In-place index fix-up of a collated batch — train.py:238-258: every key containing
    'edge' += node offset of its image (``slices['pos']``); 'bbox_idx' += proposal offset
    (``slices['labels']``).  Integer, bit-exact."""
    pos_slice = slices["pos"]
    for key in slices:
        if "edge" in key:
            s = slices[key]
            o = getattr(data, key)
            if not isinstance(o, torch.Tensor):
                continue
            for i in range(len(s) - 1):
                o[int(s[i]):int(s[i + 1])] += pos_slice[i]
        elif "bbox_idx" in key:
            s = slices[key]
            o = getattr(data, key)
            off = slices["labels"]
            for i in range(len(s) - 1):
                o[int(s[i]):int(s[i + 1])] += off[i]
    return data


# --------------------------------------------------------------------------------------
# collate + offset fix-up + host->device in one go (train.py:123-171,238-258 and the six synchronous
# .cuda() copies of architecture3cc_rpn_gp_iter2.py:107-115)
# --------------------------------------------------------------------------------------

_DEVICE_KEYS = ("x", "pos", "edge", "e_attr", "bbox_idx", "bbox", "stat_feats", "labels")
_PINNED = {}


def collate_to_device(data_list, device="cuda"):
    """List of per-image ``Data`` (CPU tensors) -> (batched ``Data`` of CUDA tensors, ``slices``).

    Same result as ``collate`` + ``fixup_offsets`` + ``.cuda()`` of every tensor, but: the items' arrays are
    packed into ONE pinned staging buffer, moved with ONE asynchronous H2D copy, the batched tensors are
    views of that single device buffer, and the per-image index offsets are added by one device kernel
    (``yolat_fixup_offsets``) instead of Python loops over the images.  Non-tensor keys (``roots`` ...)
    and ``slices`` are assembled on the host exactly like ``collate`` does."""
    from . import ops
    from ._lib import lib, check
    keys = data_list[0].keys
    tkeys = [k for k in keys if isinstance(data_list[0][k], torch.Tensor) and data_list[0][k].dim() > 0 and k in _DEVICE_KEYS]
    rest = [k for k in keys if k not in tkeys]
    B = len(data_list)
    # host-side assembly of everything that is not a device tensor, and of the slices
    host_items = [data_list[0].__class__(**{k: it[k] for k in rest}) for it in data_list] if rest else None
    slices = {}
    for k in tkeys:
        sizes = [int(it[k].shape[0]) for it in data_list]
        slices[k] = torch.tensor(np.concatenate([[0], np.cumsum(sizes)]), dtype=torch.long)
    batch = data_list[0].__class__()
    if rest:
        hb, hs = collate(host_items)
        for k in rest:
            batch[k] = hb[k]
            slices[k] = hs[k]
        node_slices = slices["pos"] if "pos" in slices else slices["x"]
        for k in rest:                 # host-resident index tensors get the reference's fix-up too (train.py:244)
            if "edge" in k and isinstance(batch[k], torch.Tensor) and batch[k].dtype == torch.long:
                for i in range(B):
                    batch[k][int(slices[k][i]):int(slices[k][i + 1])] += node_slices[i]
    # layout of the packed buffer (256-byte aligned fields) + the 4 small index tables of the fix-up
    tables = {"edge_ptr": slices["edge"], "node_ptr": slices["pos"] if "pos" in slices else slices["x"],
              "node_off": (slices["pos"] if "pos" in slices else slices["x"])[:-1],
              "prop_off": slices["labels"][:-1] if "labels" in slices else slices["bbox"][:-1]}
    fields, off = [], 0
    for k in tkeys:
        first = data_list[0][k]
        shape = (int(slices[k][-1]),) + tuple(first.shape[1:])
        nbytes = int(np.prod(shape)) * first.element_size()
        fields.append((k, first.dtype, shape, off, nbytes))
        off = (off + nbytes + 255) // 256 * 256
    for k, t in tables.items():
        nbytes = t.numel() * 8
        fields.append((k, torch.int64, (t.numel(),), off, nbytes))
        off = (off + nbytes + 255) // 256 * 256
    total = max(off, 256)
    # two pinned staging buffers used alternately; a buffer is only re-packed after the H2D copy that last read
    # it has completed (event recorded right after that copy)
    slot = _PINNED["next"] = 1 - _PINNED.get("next", 1)
    pin, ev = _PINNED.get(("buf", slot)), _PINNED.get(("ev", slot))
    if ev is not None:
        ev.synchronize()
    if pin is None or pin.numel() < total:
        pin = _PINNED[("buf", slot)] = torch.empty(int(total * 1.5), dtype=torch.uint8).pin_memory()
    dbuf = torch.empty(total, dtype=torch.uint8, device=device)
    pin_np = pin.numpy()                       # plain memcpy per item (torch.cat / copy_ wake a thread pool per call)
    for k, dtype, shape, o, nbytes in fields:
        if not nbytes:
            continue
        if k in tables:
            pin_np[o:o + nbytes] = tables[k].numpy().view(np.uint8).reshape(-1)
            continue
        pos = o
        for it in data_list:
            src = it[k].contiguous().numpy().view(np.uint8).reshape(-1)
            pin_np[pos:pos + src.size] = src
            pos += src.size
    dbuf.copy_(pin[:total], non_blocking=True)                  # the one H2D copy
    ev = _PINNED.get(("ev", slot))
    if ev is None:
        ev = _PINNED[("ev", slot)] = torch.cuda.Event()
    ev.record()
    dv = {}
    for k, dtype, shape, o, nbytes in fields:
        dv[k] = dbuf[o:o + nbytes].view(dtype).view(shape)
    for k in tkeys:
        batch[k] = dv[k]
    E, N = dv["edge"].shape[0], dv["bbox_idx"].shape[0]
    check(lib.yolat_fixup_offsets(dv["edge"].data_ptr(), E, dv["edge_ptr"].data_ptr(), dv["bbox_idx"].data_ptr(), N,
                                  dv["node_ptr"].data_ptr(), dv["node_off"].data_ptr(), dv["prop_off"].data_ptr(), B,
                                  ops._stream()), "yolat_fixup_offsets")
    batch._device_buffer = dbuf
    return batch, slices


# --------------------------------------------------------------------------------------
# two-pass inference support: root / children proposal tree and sub-batch extraction
# (architecture3cc_rpn_gp_iter2.py:139-242; tree nodes: Datasets/graph_dict3.py:24-27,745-767)
# --------------------------------------------------------------------------------------

class idxTree(object):
    """Node of the proposal tree a dataset item carries in ``data.roots``: ``value`` holds the
    item-local ranges ``idx_pos``, ``idx_edge``, ``idx_edge_super`` (start, end) and ``idx_bbox``."""

    def __init__(self):
        self.children = []
        self.value = {}


def _expand_ranges(starts, ends):
    """Concatenation of ``range(s, e)`` for every (s, e) pair, as one int64 array."""
    starts = np.asarray(starts, dtype=np.int64)
    lens = np.asarray(ends, dtype=np.int64) - starts
    if np.any(lens < 0):
        raise ValueError("idx range with end < start")
    total = int(lens.sum())
    if total == 0:
        return np.zeros(0, dtype=np.int64)
    first = np.cumsum(lens) - lens
    return np.repeat(starts - first, lens) + np.arange(total, dtype=np.int64)


def select_tree_ranges(data, slices, has_object=None):
    """First (``has_object is None``) or second pass selection of arch:153-164 / :277-296 as index RANGES.
    Returns (pos_s, pos_e, edge_s, edge_e, slice_bbox, slice_image_bbox): int64 arrays of global
    [start, end) node / edge ranges per selected tree node, the global proposal rows (python list) and the
    per-image offsets into slice_bbox.  O(#selected tree nodes) host work; nothing of size O(nodes)."""
    roots = data.roots
    slice_root = [int(v) for v in slices["roots"]]
    pos_s, pos_e, edge_s, edge_e, slice_bbox, image_off = [], [], [], [], [], [0]
    count = 0
    for i in range(len(slice_root) - 1):
        off_pos, off_edge, off_bbox = int(slices["pos"][i]), int(slices["edge"][i]), int(slices["bbox"][i])
        int(slices["edge_super"][i])               # the reference reads it (KeyError if absent)
        for root in roots[slice_root[i]:slice_root[i + 1]]:
            if has_object is None:
                nodes = (root,)
            else:
                nodes = root.children if has_object[count] else ()
                count += 1
            for nd in nodes:
                v = nd.value
                pos_s.append(v["idx_pos"][0] + off_pos)
                pos_e.append(v["idx_pos"][1] + off_pos)
                edge_s.append(v["idx_edge"][0] + off_edge)
                edge_e.append(v["idx_edge"][1] + off_edge)
                slice_bbox.append(int(v["idx_bbox"] + off_bbox))
        image_off.append(len(slice_bbox))
    as64 = lambda a: np.asarray(a, dtype=np.int64)
    return as64(pos_s), as64(pos_e), as64(edge_s), as64(edge_e), slice_bbox, image_off


def select_tree_nodes(data, slices, has_object=None):
    """`select_tree_ranges` with the ranges expanded on the host: (slice_pos, slice_edge, slice_bbox,
    slice_image_bbox) exactly as the reference's Python lists (used by the CPU cross-checks)."""
    ps, pe, es, ee, slice_bbox, image_off = select_tree_ranges(data, slices, has_object)
    return _expand_ranges(ps, pe), _expand_ranges(es, ee), slice_bbox, image_off


def build_subset(data, slice_pos, slice_edge, slice_bbox):
    """``build_data`` of arch:166-242 without the per-element Python loops: node re-numbering through
    a lookup table, run-length renumbering of ``bbox_idx``.  Integer results identical to the loops."""
    N = data.x.shape[0]
    sp = torch.from_numpy(slice_pos)
    se = torch.from_numpy(slice_edge)
    o2n = np.full(N, -1, dtype=np.int64)
    o2n[slice_pos] = np.arange(len(slice_pos), dtype=np.int64)    # later duplicates win, like the dict
    new = data.__class__(x=data.x[sp], pos=data.pos[sp])
    old_edge = data.edge[se].numpy()
    new_edge = o2n[old_edge] if len(old_edge) else np.zeros((0, 2), dtype=np.int64)
    if (new_edge < 0).any():
        bad = old_edge[(new_edge < 0)][0]
        raise KeyError(int(bad))                                    # the reference's o2n[...] KeyError
    new.edge = torch.from_numpy(new_edge.reshape(-1, 2))
    new.e_attr = data.e_attr[se]
    sb = torch.tensor(slice_bbox, dtype=torch.long)
    new.bbox = data.bbox[sb]
    new.stat_feats = data.stat_feats[sb]
    old_idx = data.bbox_idx[sp].numpy()
    if len(old_idx):
        change = np.concatenate([[0], (old_idx[1:] != old_idx[:-1]).astype(np.int64)])
        new.bbox_idx = torch.from_numpy(np.cumsum(change))
    else:
        new.bbox_idx = torch.zeros(0, dtype=torch.long)
    return new


def interleave_root_child(slice_p, slice_c, out_p, out_c):
    """``interleaf_pc`` of arch:317-328: per image, the root rows followed by the child rows."""
    out, s = [], [0]
    for i in range(len(slice_c) - 1):
        out.append(out_p[slice_p[i]:slice_p[i + 1]])
        out.append(out_c[slice_c[i]:slice_c[i + 1]])
        s.append(s[-1] + slice_p[i + 1] - slice_p[i] + slice_c[i + 1] - slice_c[i])
    return out, s


def synth_roots(item, seed, cluster_lo=1, cluster_hi=4):
    """Proposal tree for a ``synth_graph`` item (needs its edges grouped by proposal, which
    synth_graph guarantees): consecutive proposals form clusters of U{lo..hi}; the member with the
    largest box area is the root, the others its children (graph_dict3.py:740-767)."""
    rng = np.random.default_rng(seed)
    bbox_idx = item.bbox_idx.numpy()
    P = int(item.bbox.shape[0])
    node_ptr = np.searchsorted(bbox_idx, np.arange(P + 1))
    owner = bbox_idx[item.edge[:, 0].numpy()] if item.edge.shape[0] else np.zeros(0, dtype=np.int64)
    if len(owner) and np.any(owner[1:] < owner[:-1]):
        raise ValueError("edges are not grouped by proposal")
    edge_ptr = np.searchsorted(owner, np.arange(P + 1))
    box = item.bbox.numpy()
    area = (box[:, 2] - box[:, 0]) * (box[:, 3] - box[:, 1])
    roots, p = [], 0
    while p < P:
        k = min(int(rng.integers(cluster_lo, cluster_hi + 1)), P - p)
        members = list(range(p, p + k))
        top = p + int(np.argmax(area[p:p + k]))
        nodes = {}
        for m in members:
            t = idxTree()
            t.value["idx_pos"] = (int(node_ptr[m]), int(node_ptr[m + 1]))
            t.value["idx_edge"] = (int(edge_ptr[m]), int(edge_ptr[m + 1]))
            t.value["idx_edge_super"] = (0, 0)
            t.value["idx_bbox"] = m
            nodes[m] = t
        root = nodes[top]
        root.children = [nodes[m] for m in members if m != top]
        roots.append(root)
        p += k
    return roots


# --------------------------------------------------------------------------------------
# synthetic graphs (SURVEY.md §8 d)
# --------------------------------------------------------------------------------------

def synth_graph(num_proposals, nodes_lo, nodes_hi, edge_factor=1.2, n_classes=17, seed=0,
                edges_per_proposal=None, augmented=False, with_roots=False):
    """One synthetic "image" (a dataset item) as a ``Data``.

    nodes per proposal ~ U{nodes_lo..nodes_hi}; directed edges per proposal =
    ``edges_per_proposal`` or ceil(edge_factor * n_p), uniform inside the proposal, src != dst,
    stored once; 85 % of e_attr rows exactly zero, rest N(0, 0.05^2).
    """
    rng = np.random.default_rng(seed)
    P = int(num_proposals)
    n_p = rng.integers(nodes_lo, nodes_hi + 1, size=P)
    N = int(n_p.sum())
    starts = np.concatenate([[0], np.cumsum(n_p)])[:-1]
    bbox_idx = np.repeat(np.arange(P), n_p)

    pos = rng.random((N, 2)).astype(np.float32)
    # per-proposal min-max normalisation touching 0 and 1 on both axes (graph_dict3.py:714)
    for p in range(P):
        s, e = starts[p], starts[p] + n_p[p]
        seg = pos[s:e]
        lo, hi = seg.min(0), seg.max(0)
        span = np.where(hi > lo, hi - lo, 1.0)
        pos[s:e] = (seg - lo) / span
    if augmented:  # graph_dict3.py:236-258,283-298: positions roam ~[-0.5, 2.1]
        pos = (pos - 0.5) * rng.uniform(0.4, 1.6) + 0.5 + rng.uniform(-0.1, 0.1, size=(1, 2))
        pos = pos.astype(np.float32)
    x = np.zeros((N, 5), dtype=np.float32)
    x[:, 3:5] = pos

    if edges_per_proposal is None:
        e_p = np.ceil(edge_factor * n_p).astype(np.int64)
    else:
        e_p = np.full(P, int(edges_per_proposal), dtype=np.int64)
    E = int(e_p.sum())
    owner = np.repeat(np.arange(P), e_p)
    n_own = n_p[owner]
    a = rng.integers(0, 1 << 30, size=E) % n_own
    b = rng.integers(0, 1 << 30, size=E) % (n_own - 1)
    b = np.where(b >= a, b + 1, b)              # src != dst
    edge = np.stack([starts[owner] + a, starts[owner] + b], axis=1).astype(np.int64)

    e_attr = (rng.standard_normal((E, 4)) * 0.05).astype(np.float32)
    e_attr[rng.random(E) < 0.85] = 0.0

    labels = rng.integers(0, n_classes, size=P).astype(np.int64)
    bbox = rng.random((P, 4)).astype(np.float32)
    bbox[:, 2:] = bbox[:, :2] + 0.05 + bbox[:, 2:] * 0.2
    stat = rng.random((P, 13)).astype(np.float32)

    d = Data(x=torch.from_numpy(x), pos=torch.from_numpy(pos))
    d.edge = torch.from_numpy(edge)
    d.e_attr = torch.from_numpy(e_attr)
    d.bbox_idx = torch.from_numpy(bbox_idx.astype(np.int64))
    d.bbox = torch.from_numpy(bbox)
    d.stat_feats = torch.from_numpy(stat)
    d.labels = torch.from_numpy(labels)
    d.is_super = torch.zeros(N, dtype=torch.bool)
    if with_roots:      # what predict() needs on top of forward(): the proposal tree + (empty) super edges
        d.edge_super = torch.zeros((0, 2), dtype=torch.long)
        d.e_attr_super = torch.zeros((0, 4), dtype=torch.float32)
        d.roots = synth_roots(d, seed + 7919)
    return d


def synth_batch(n_graphs, seed, **kw):
    """``n_graphs`` synthetic items collated + offset-fixed like train.py does."""
    items = [synth_graph(seed=seed * 1000 + i, **kw) for i in range(n_graphs)]
    data, slices = collate(items)
    fixup_offsets(data, slices)
    return data, slices


# Named configurations of BASELINE.md / SURVEY.md §8(d)
def config(name, rank=0):
    """Returns (data, slices, opt_kwargs, n_graphs) for cfg '1'..'5' (synthetic, seeded)."""
    name = str(name)
    if name == "1":      # Floorplans-sized, batch 1
        d, s = synth_batch(1, 1, num_proposals=2000, nodes_lo=4, nodes_hi=40, edge_factor=1.2)
        return d, s, dict(n_classes=17, n_blocks=2, n_blocks_out=2), 1
    if name == "2":      # N=10k / E=40k / P=400
        d, s = synth_batch(1, 2, num_proposals=400, nodes_lo=25, nodes_hi=25, edges_per_proposal=100)
        return d, s, dict(n_classes=17, n_blocks=2, n_blocks_out=2), 1
    if name == "3":      # 4 cfg-1 graphs collated, train step
        d, s = synth_batch(4, 3, num_proposals=2000, nodes_lo=4, nodes_hi=40, edge_factor=1.2,
                           augmented=True)
        return d, s, dict(n_classes=17, n_blocks=2, n_blocks_out=2), 4
    if name == "4":      # Diagrams-style, 32 graphs / step / GPU
        d, s = synth_batch(32, 4 + rank, num_proposals=300, nodes_lo=4, nodes_hi=24,
                           edge_factor=1.2, n_classes=22, augmented=True)
        return d, s, dict(n_classes=22, n_blocks=2, n_blocks_out=2), 32
    if name == "5":      # N=200k / E=1.2M / P=8000, n_blocks=4
        d, s = synth_batch(1, 5 + rank, num_proposals=8000, nodes_lo=25, nodes_hi=25,
                           edges_per_proposal=150)
        return d, s, dict(n_classes=17, n_blocks=4, n_blocks_out=2), 1
    raise ValueError("unknown config %r" % name)
