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

import ctypes
import os

import struct

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

ctypes_i64 = ctypes.c_int64
_DEVICE_KEYS = ("x", "pos", "edge", "e_attr", "bbox_idx", "bbox", "stat_feats", "labels")
_CSR_SKIP = ("edge", "e_attr", "bbox_idx")       # replaced by the prepared graph in csr mode
_PINNED = {}


def _source_key(item, names):
    """(data_ptr, version counter, shape) of the tensors a cache entry was computed from: an in-place edit (augmentation,
    normalisation) or a re-assignment of any of them invalidates the entry — the same rule `_stage` applies to the
    non-csr path."""
    d = item.__dict__          # (the attribute bag itself: `item.keys` builds a list per look-up)
    out = []
    for k in names:
        t = d.get(k)
        out.append(None if t is None else (t.data_ptr(), t._version, t.shape))
    return tuple(out)


def item_csr(item):
    """The destination-sorted (CSR) form of ONE dataset item, computed once by the library's host code
    (``yolat_item_csr_host``: the host twin of the device-side ``yolat_graph_prepare``) and cached on the item — a dataset
    item's graph never changes (the reference caches its proposals the same way, Datasets/graph_dict3.py:924-929).
    Raises like ``Graph.check_status`` on ids outside [0, N) / an unsorted ``bbox_idx``."""
    c = item.__dict__.get("_yolat_csr")
    # the CSR is a function of edge, e_attr, bbox_idx and the two row COUNTS only: an in-place normalisation / augmentation of
    # x (or of the boxes) must not throw the sorted graph away
    key = _source_key(item, ("edge", "e_attr", "bbox_idx")) + (int(item.x.shape[0]), int(item.bbox.shape[0]))
    if c is not None and c.get("key") == key:
        return c
    from ._lib import lib, check
    edge, e_attr, bidx = item.edge, item.e_attr.contiguous(), item.bbox_idx.contiguous()
    if edge.dtype != torch.long or bidx.dtype != torch.long or e_attr.dtype != torch.float32:
        raise TypeError("item_csr: edge / bbox_idx must be int64 and e_attr float32 (Datasets/graph_dict3.py:1049-1066)")
    N, E = int(item.x.shape[0]), int(edge.shape[0])
    P = int(item.bbox.shape[0])
    i32 = lambda n: np.empty(max(n, 1), dtype=np.int32)
    c = {"N": N, "E": E, "P": P, "row_ptr": i32(N + 1), "perm": i32(E), "src": i32(E), "dst": i32(E),
         "attr": np.empty((max(E, 1), 4), dtype=np.float32), "seg_ptr": i32(P + 1), "node_seg": i32(N)}
    status = np.zeros(1, dtype=np.int32)
    check(lib.yolat_item_csr_host(edge.data_ptr(), edge.stride(0), edge.stride(1) if edge.dim() == 2 else 1,
                                  e_attr.data_ptr(), bidx.data_ptr(), E, N, P, c["row_ptr"].ctypes.data,
                                  c["perm"].ctypes.data, c["src"].ctypes.data, c["dst"].ctypes.data, c["attr"].ctypes.data,
                                  c["seg_ptr"].ctypes.data, c["node_seg"].ctypes.data, status.ctypes.data),
          "yolat_item_csr_host")
    from . import ops
    st = int(status[0])
    if st & ops.STATUS_EDGE_RANGE:
        raise IndexError("edge_index contains a node id outside [0, N)")
    if st & ops.STATUS_SEG_UNSORTED:
        raise ValueError("bbox_idx is not non-decreasing")
    if st & ops.STATUS_SEG_RANGE:
        raise IndexError("bbox_idx contains a proposal id outside [0, P)")
    from ._lib import ItemCsr
    c["struct"] = ItemCsr(N, E, P, c["row_ptr"].ctypes.data, c["src"].ctypes.data, c["dst"].ctypes.data,
                          c["attr"].ctypes.data, c["seg_ptr"].ctypes.data, c["node_seg"].ctypes.data)
    c["key"] = key
    item.__dict__["_yolat_csr"] = c
    item.__dict__.pop("_yolat_desc", None)        # the descriptor points into the previous CSR arrays
    return c


def item_locality(item):
    """The locality record (``yolat_locality``) of ONE dataset item, computed once by the library's host code
    (``yolat_item_locality_host``) and cached on the item like its CSR: is the edge list grouped by proposal
    (Datasets/graph_dict3.py:725,752-764), does every edge stay inside its proposal (:582-600,733), nodes / edges of the
    largest proposal.  Returns (flags, max_nodes, max_edges)."""
    c = item.__dict__.get("_yolat_loc_item")
    key = _source_key(item, ("edge", "bbox_idx")) + (int(item.x.shape[0]), int(item.bbox.shape[0]))
    if c is not None and c[0] == key:
        return c[1]
    from ._lib import lib, check, Locality
    edge, bidx = item.edge, item.bbox_idx.contiguous()
    if edge.dtype != torch.long or bidx.dtype != torch.long:
        raise TypeError("item_locality: edge / bbox_idx must be int64 (Datasets/graph_dict3.py:1049-1066)")
    N, E, P = int(item.x.shape[0]), int(edge.shape[0]), int(item.bbox.shape[0])
    loc = Locality()
    check(lib.yolat_item_locality_host(edge.data_ptr() if E > 0 else None, edge.stride(0) if E > 0 else 2,
                                       edge.stride(1) if (E > 0 and edge.dim() == 2) else 1, bidx.data_ptr(), E, N, P,
                                       ctypes.byref(loc)), "yolat_item_locality_host")
    rec = (int(loc.flags), int(loc.max_nodes), int(loc.max_edges))
    item.__dict__["_yolat_loc_item"] = (key, rec)
    return rec


def batch_locality(items):
    """The batch's record from its items': collate keeps the order of the items and of their edges and adds per-image
    offsets (train.py:238-258), so the batch's edge list is grouped / local iff every item's is, and its largest proposal
    is the largest of the items'."""
    from ._lib import Locality
    flags = mn = me = 0
    for it in items:
        f, n, e = item_locality(it)
        flags |= f
        mn = n if n > mn else mn
        me = e if e > me else me
    return Locality(1, flags, mn, me)


def _item_desc(item, ship, csr=True):
    """The cached yolat_item_desc of a dataset item: pointers / sizes of its dense arrays in `ship` order + its CSR
    (csr=True), or — COO mode — no CSR and the offset fix-up of the int64 index keys (train.py:238-258: keys containing
    'edge' += the image's node offset, 'bbox_idx' += its proposal offset) described for the native collate."""
    slot = "_yolat_desc" if csr else "_yolat_desc_coo"
    d = item.__dict__.get(slot)
    key = _source_key(item, tuple(ship) + (("edge", "e_attr", "bbox_idx") if csr else ()))
    if d is not None and d[0] == ship and len(d) > 3 and d[3] == key:
        return d[1]
    from ._lib import ItemDesc
    if len(ship) > 8:
        raise ValueError("the native collate ships at most 8 dense keys")
    desc = ItemDesc()
    desc.n_keys = len(ship)
    keep = []
    for f, k in enumerate(ship):
        t = item[k]
        if not t.is_contiguous():
            t = t.contiguous()
        keep.append(t)
        desc.key[f].ptr, desc.key[f].bytes = t.data_ptr(), t.numel() * t.element_size()
        desc.rows[f] = t.shape[0]
    if csr:
        desc.csr = item_csr(item)["struct"]
    else:
        desc.node_key = ship.index("pos") if "pos" in ship else ship.index("x")
        desc.prop_key = ship.index("labels") if "labels" in ship else ship.index("bbox")
        for f, k in enumerate(ship):
            if item[k].dtype == torch.int64 and "edge" in k:
                desc.fix[f] = 1
            elif "bbox_idx" in k:
                if item[k].dtype != torch.int64:
                    raise TypeError("bbox_idx must be int64")
                desc.fix[f] = 2
    item.__dict__[slot] = (ship, desc, keep, key)
    return desc


_ZERO_STATUS = {}
# module flag (no environment switch): measured 5.86 k -> 6.07 k graphs/s at cfg 2 — the Python stream context costs what
# the overlap gains, so the separate copy stream stays off
_COPY_STREAM_ON = False


def _collate_csr(data_list, device, tkeys, ship, batch, slices):
    """csr mode of collate_to_device: ONE native call (yolat_collate_batch) lays out, packs and merges."""
    from . import ops
    from ._lib import lib, check
    first = data_list[0]
    B, nk = len(data_list), len(ship)
    ptrs = (ctypes.c_void_p * B)(*[ctypes.addressof(_item_desc(it, ship)) for it in data_list])
    off = (ctypes_i64 * (nk + 6))()
    tot = (ctypes_i64 * 4)()
    sl = np.empty((nk, B + 1), dtype=np.int64)
    slot = _PINNED["next"] = 1 - _PINNED.get("next", 1)
    pin, ev = _PINNED.get(("buf", slot)), _PINNED.get(("ev", slot))
    if ev is not None:
        ev.synchronize()
    for attempt in range(2):
        cap = pin.numel() if pin is not None else 0
        check(lib.yolat_collate_batch(ptrs, B, pin.data_ptr() if pin is not None else None, cap, off,
                                      ctypes.addressof(tot), sl.ctypes.data, ctypes.addressof(tot) + 8), "yolat_collate_batch")
        if tot[0] <= cap:
            break
        pin = _PINNED[("buf", slot)] = torch.empty(int(tot[0] * 1.5), dtype=torch.uint8).pin_memory()
    total, Nt, Et, Pt = int(tot[0]), int(tot[1]), int(tot[2]), int(tot[3])
    if ev is None:
        ev = _PINNED[("ev", slot)] = torch.cuda.Event()
    if _COPY_STREAM_ON:
        # (_COPY_STREAM_ON) the H2D copy on its own stream, so that batch i + 1 crosses PCIe while batch i's forward
        # runs (on one stream the 1.3 MB copy of cfg 2 — ~65 us with its launch — and the 100 us forward alternate).  The buffer is allocated
        # in the copy stream's pool and handed to the consumer's stream: record_stream() keeps the allocator from
        # re-using it before that stream is done with it.
        cur = ops.current_stream_object()
        cs = _PINNED.get("copy_stream")
        if cs is None:
            cs = _PINNED["copy_stream"] = torch.cuda.Stream(device=device)
        # (torch.cuda.stream(cs) as a context manager costs ~25 us of Python per use; the two calls it ends in, 2)
        _set = torch._C._cuda_setStream
        _set(stream_id=cs.stream_id, device_index=cs.device_index, device_type=cs.device_type)
        try:
            dbuf = torch.empty(total, dtype=torch.uint8, device=device)
            dbuf.copy_(pin[:total], non_blocking=True)          # the one H2D copy
            ev.record(cs)
        finally:
            _set(stream_id=cur.stream_id, device_index=cur.device_index, device_type=cur.device_type)
        cur.wait_event(ev)
        dbuf.record_stream(cur)
    else:
        dbuf = torch.empty(total, dtype=torch.uint8, device=device)
        dbuf.copy_(pin[:total], non_blocking=True)              # the one H2D copy
        ev.record(ops.current_stream_object())
    typed = {}

    def view(o, dtype, shape):
        b = typed.get(dtype)
        if b is None:
            b = typed[dtype] = dbuf.view(dtype)
        es = b.element_size()
        n = 1
        for d_ in shape:
            n *= d_
        return b[o // es:o // es + n].view(shape)

    for f, k in enumerate(ship):
        t = first[k]
        batch[k] = view(off[f], t.dtype, (int(sl[f, B]),) + tuple(t.shape[1:]))
        slices[k] = torch.from_numpy(sl[f])
    # the keys that are not shipped still get their slices (train.py:141-147 builds them for every key)
    for k in tkeys:
        if k not in slices:
            if B == 1:
                slices[k] = torch.tensor([0, first[k].shape[0]], dtype=torch.int64)
            else:
                ends = np.zeros(B + 1, dtype=np.int64)
                np.cumsum([it[k].shape[0] for it in data_list], out=ends[1:])
                slices[k] = torch.from_numpy(ends)
    # the merged CSR stays six offsets into the device buffer until somebody asks for the arrays (ops.PackedGraph)
    g = ops.PackedGraph.from_buffer(dbuf, [off[nk + i] for i in range(6)], Nt, Et, Pt)
    batch.__dict__["_yolat_graph"] = g
    batch.__dict__["_yolat_loc"] = batch_locality(data_list)       # decided on the host, before any forward is enqueued
    batch._device_buffer = dbuf
    return batch, slices


def collate_to_device(data_list, device="cuda", csr=False):
    """List of per-image ``Data`` (CPU tensors) -> (batched ``Data`` of CUDA tensors, ``slices``).

    Same result as ``collate`` + ``fixup_offsets`` + ``.cuda()`` of every tensor, but: the items' arrays are
    packed into ONE pinned staging buffer by ONE native call (``yolat_collate_pack``, GIL released), moved with ONE
    asynchronous H2D copy, the batched tensors are views of that single device buffer, and the per-image index
    offsets are added by one device kernel (``yolat_fixup_offsets``) instead of Python loops over the images.
    Non-tensor keys (``roots`` ...) and ``slices`` are assembled on the host exactly like ``collate`` does.

    ``csr=True``: the batch carries a PREPARED graph instead of ``edge`` / ``e_attr`` / ``bbox_idx``: every item's
    destination-sorted form is computed once (``item_csr``, cached on the item) and the batch's form is their
    concatenation with the offsets added (``yolat_collate_csr_pack``; bit-identical to rebuilding it on the device from
    the collated COO list).  ``SparseCADGCN.forward`` then skips the COO -> CSR conversion.  The raw ``edge`` /
    ``e_attr`` / ``bbox_idx`` tensors are not shipped (``slices`` still describes them)."""
    from . import ops
    from ._lib import lib, check, Span
    keys = data_list[0].keys
    first = data_list[0]
    tkeys = [k for k in keys if isinstance(first[k], torch.Tensor) and first[k].dim() > 0 and k in _DEVICE_KEYS]
    rest = [k for k in keys if k not in tkeys]
    B = len(data_list)
    batch = first.__class__()
    slices = {}
    ship = [k for k in tkeys if not (csr and k in _CSR_SKIP)]
    if csr:
        _collate_csr(data_list, device, tkeys, tuple(ship), batch, slices)
    else:
        for k in tkeys:
            ends = np.zeros(B + 1, dtype=np.int64)
            np.cumsum([it[k].shape[0] for it in data_list], out=ends[1:])
            slices[k] = torch.from_numpy(ends)
    # host-side assembly of everything that is not a device tensor
    if rest:
        _collate_host_keys(data_list, rest, batch, slices)
    if csr:
        return batch, slices
    node_ptr = slices["pos"] if "pos" in slices else slices["x"]
    # layout of the packed buffer (256-byte aligned fields)
    fields, off = [], 0

    def field(name, dtype, shape, esize):
        nonlocal off
        nbytes = int(np.prod(shape)) * esize
        fields.append((name, dtype, shape, off, nbytes))
        off = (off + nbytes + 255) // 256 * 256

    for k in ship:
        field(k, first[k].dtype, (int(slices[k][-1]),) + tuple(first[k].shape[1:]), first[k].element_size())
    n_pack = len(fields)
    tables = {"edge_ptr": slices["edge"], "node_ptr": node_ptr, "node_off": node_ptr[:-1],
              "prop_off": slices["labels"][:-1] if "labels" in slices else slices["bbox"][:-1]}
    for k, t in tables.items():
        field(k, torch.int64, (t.numel(),), 8)
    total = max(off, 256)
    # two pinned staging buffers used alternately; a buffer is only re-packed after the H2D copy that last read
    # it has completed (event recorded right after that copy)
    slot = _PINNED["next"] = 1 - _PINNED.get("next", 1)
    pin, ev = _PINNED.get(("buf", slot)), _PINNED.get(("ev", slot))
    if ev is not None:
        ev.synchronize()
    if pin is None or pin.numel() < total:
        pin = _PINNED[("buf", slot)] = torch.empty(int(total * 1.5), dtype=torch.uint8).pin_memory()
    base = pin.data_ptr()
    dbuf = torch.empty(total, dtype=torch.uint8, device=device)
    # ONE native call moves every shipped key of every item (torch.cat / copy_ wake a thread pool per call)
    spans = (Span * (n_pack * B))()
    foff = (ctypes_i64 * n_pack)()
    keep = []
    for f, (k, dtype, shape, o, nbytes) in enumerate(fields[:n_pack]):
        foff[f] = o
        for i, it in enumerate(data_list):
            t = it[k]
            if not t.is_contiguous():
                t = t.contiguous()
                keep.append(t)
            sp = spans[f * B + i]
            sp.ptr, sp.bytes = t.data_ptr(), t.numel() * t.element_size()
    check(lib.yolat_collate_pack(base, foff, spans, n_pack, B), "yolat_collate_pack")
    pin_np = pin.numpy()
    for k, dtype, shape, o, nbytes in fields[n_pack:]:
        pin_np[o:o + nbytes] = tables[k].numpy().view(np.uint8).reshape(-1)
    dbuf.copy_(pin[:total], non_blocking=True)                  # the one H2D copy
    ev = _PINNED.get(("ev", slot))
    if ev is None:
        ev = _PINNED[("ev", slot)] = torch.cuda.Event()
    ev.record()
    dv = {}
    for k, dtype, shape, o, nbytes in fields:
        dv[k] = dbuf[o:o + nbytes].view(dtype).view(shape)
    for k in ship:
        batch[k] = dv[k]
    E, N = dv["edge"].shape[0], dv["bbox_idx"].shape[0]
    check(lib.yolat_fixup_offsets(dv["edge"].data_ptr(), E, dv["edge_ptr"].data_ptr(), dv["bbox_idx"].data_ptr(), N,
                                  dv["node_ptr"].data_ptr(), dv["node_off"].data_ptr(), dv["prop_off"].data_ptr(), B,
                                  ops._stream()), "yolat_fixup_offsets")
    batch._device_buffer = dbuf
    # the locality record travels with the tensors it describes: (edge, bbox_idx) storage + version, checked by the forward
    e_t, b_t = dv["edge"], dv["bbox_idx"]
    batch.__dict__["_yolat_loc"] = ((e_t.data_ptr(), e_t._version, b_t.data_ptr(), b_t._version), batch_locality(data_list))
    return batch, slices


_UNPACK_I64 = struct.Struct("q").unpack_from


class _BatchState(object):
    """What a DeviceLoader batch is made from on demand: the slot's device buffer, the field offsets, the slices (bytes
    copied out of the slot at draw time) and the host items."""
    __slots__ = ("mem", "offs", "sl_bytes", "items", "ship", "tkeys", "rest", "B", "sl", "host", "names")

    def __init__(self, mem, offs, sl_bytes, items, ship, tkeys, rest, B):
        self.mem, self.offs, self.sl_bytes = mem, offs, sl_bytes
        self.items, self.ship, self.tkeys, self.rest, self.B = items, ship, tkeys, rest, B
        self.sl = None
        self.host = None
        self.names = None

    def slice_rows(self):
        if self.sl is None:
            nk = len(self.ship)
            self.sl = np.frombuffer(self.sl_bytes, dtype=np.int64).reshape(nk, self.B + 1).copy()
        return self.sl

    def all_names(self):
        if self.names is None:
            self.names = tuple(k for k in self.ship) + tuple(self.rest)
        return self.names

    def tensor(self, k):
        """view of the device buffer for shipped key k"""
        f = self.ship.index(k)
        t = self.items[0].__dict__[k]
        _, _, buf, typed = self.mem
        b = typed.get(t.dtype)
        if b is None:
            b = typed[t.dtype] = (buf.view(t.dtype), t.element_size())
        b, es = b
        B = self.B
        rows = _UNPACK_I64(self.sl_bytes, 8 * (f * (B + 1) + B))[0]
        tail = tuple(t.shape[1:])
        if len(tail) == 1:                       # one tensor operation instead of slice + view
            return b.as_strided((rows, tail[0]), (tail[0], 1), self.offs[f] // es)
        n = rows
        for d_ in tail:
            n *= d_
        o = self.offs[f] // es
        return b[o:o + n].view((rows,) + tail)

    def host_keys(self):
        if self.host is None:
            hb, hs = self.items[0].__class__(), {}
            for k in self.tkeys:                     # the fix-up of host-resident edge tensors needs the node slices
                hs[k] = self.make_slices(k)
            _collate_host_keys(self.items, self.rest, hb, hs)
            self.host = (hb, hs)
        return self.host

    def make_slices(self, k):
        if k in self.ship:
            return torch.from_numpy(self.slice_rows()[self.ship.index(k)])
        if k in self.tkeys:
            ends = np.zeros(self.B + 1, dtype=np.int64)
            np.cumsum([it[k].shape[0] for it in self.items], out=ends[1:])
            return torch.from_numpy(ends)
        if k in self.rest:
            return self.host_keys()[1][k]
        raise KeyError(k)


class _LazySlices(dict):
    """`slices` of a DeviceLoader batch: a dict whose values are produced on first access (the forward never reads them)."""
    _st = None

    def __missing__(self, k):
        st = self._st
        if st is None:
            raise KeyError(k)
        v = st.make_slices(k)
        dict.__setitem__(self, k, v)
        return v

    def _names(self):
        st = self._st
        return () if st is None else tuple(st.tkeys) + tuple(st.rest)

    def __contains__(self, k):
        return dict.__contains__(self, k) or k in self._names()

    def get(self, k, default=None):
        return self[k] if k in self else default

    def _all(self):
        for k in self._names():
            self[k]

    def keys(self):
        self._all()
        return dict.keys(self)

    def __iter__(self):
        self._all()
        return dict.__iter__(self)

    def __len__(self):
        self._all()
        return dict.__len__(self)

    def items(self):
        self._all()
        return dict.items(self)

    def values(self):
        self._all()
        return dict.values(self)

    def copy(self):
        self._all()
        return dict(dict.items(self))


class _LazyBatch(Data):
    """The batch of a DeviceLoader: the views of the device buffer and the host-side keys are produced on first access."""

    def __init__(self, *args, **kwargs):
        # the loader builds it empty (attributes appear on first access); code that makes a NEW batch of the same class —
        # predict()'s sub-batches: `data.__class__(x=..., pos=...)`, arch:180 — gets a plain attribute bag
        if args or kwargs:
            Data.__init__(self, *args, **kwargs)

    def __setattr__(self, name, value):
        # the eval fast paths read a loader batch through addresses captured at draw time (`_yolat_x`, `_yolat_raw`) and a
        # locality record taken from the host items: a caller that re-assigns a shipped key (`batch.x = norm(batch.x)`,
        # `batch.edge = ...`) must get the forward on ITS tensors — drop the shortcuts, the regular staged path takes over
        d = self.__dict__
        if name[0] != "_" and ("_yolat_x" in d or "_yolat_raw" in d or "_yolat_loc" in d):
            st = d.get("_lazy")
            if st is not None and name in st.ship:
                d.pop("_yolat_x", None)
                d.pop("_yolat_raw", None)
                if name in ("edge", "bbox_idx"):
                    d.pop("_yolat_loc", None)
        d[name] = value

    def __getattr__(self, name):               # only reached when the normal look-up fails
        st = self.__dict__.get("_lazy")
        if st is not None and name[0] != "_":
            if name in st.ship:
                v = st.tensor(name)
            elif name in st.rest:
                v = st.host_keys()[0][name]
            else:
                raise AttributeError(name)
            self.__dict__[name] = v
            return v
        raise AttributeError(name)

    @property
    def keys(self):
        st = self.__dict__.get("_lazy")
        if st is not None:
            for name in st.all_names():
                getattr(self, name)
        return Data.keys.fget(self)


class _DeviceMemory(object):
    """A device allocation owned by the native loader as a zero-copy torch tensor (CUDA array interface)."""

    def __init__(self, ptr, nbytes):
        self.__cuda_array_interface__ = {"shape": (int(nbytes),), "typestr": "|u1", "data": (int(ptr), False), "version": 2}


class DeviceLoader(object):
    """Iterator over device batches whose collate + host -> device copy run on a NATIVE worker thread
    (csrc/loader.hip) while the consumer still works on the previous batch — what the reference gets from
    DataLoader(num_workers=8) (train.py:178-189) plus the six .cuda() copies of forward() (arch:107-115), without a Python
    thread in the way.

        for batch, slices in DeviceLoader(lists_of_items):      # lists_of_items: iterable of lists of CPU `Data`
            logits, boxes = model(batch, slices)

    Every batch is what ``collate_to_device(items, csr=True)`` returns (bit-identical: same native call), i.e. it carries
    the merged destination-sorted graph (``item_csr`` of each item, cached on the item) instead of edge / e_attr /
    bbox_idx.  ``csr=False`` (COO mode): what ``collate_to_device(items)`` returns — edge / e_attr / bbox_idx travel with
    the offset fix-up of train.py:238-258 applied by the worker while it copies them; the forward rebuilds the CSR on the
    device (and ``predict`` works on such a batch).  A batch's tensors live in the loader's ring of ``slots`` device buffers: they are valid until ``slots - 1``
    further batches have been drawn, and everything that reads them must have been ENQUEUED, on the stream that was current
    when the batch was drawn, by the time the next batch is drawn (the loader then hands the slot back behind an event on
    that stream).  Consecutive batches may be drawn under different streams (``torch.cuda.set_stream`` between draws): their
    forwards then overlap on the GPU and the hand-over disappears behind them (``slots`` >= streams + 2)."""

    def __init__(self, batches, device="cuda", slots=3, csr=True):
        from ._lib import lib
        if slots < 2:
            raise ValueError("DeviceLoader needs at least two slots")
        self._it = iter(batches)
        self._device = torch.device(device)
        self._slots = int(slots)
        with torch.cuda.device(self._device):
            self._h = lib.yolat_loader_create(self._slots)
        if not self._h:
            raise RuntimeError("yolat_loader_create failed")
        from . import ops
        from ._lib import check, LoaderBatch
        self._lib, self._check, self._LoaderBatch = lib, check, LoaderBatch
        self._stream, self._from_buffer = ops._stream, ops.PackedGraph.from_buffer
        self._pending = []          # submitted, not yet drawn: (items, ship, tkeys, rest, pointer array)
        self._csr = bool(csr)
        self._held = None           # slot of the batch the consumer holds
        self._held_stream = None    # ... and the stream it was drawn on (its readers are enqueued there)
        self._mem = {}              # slot -> (ptr, capacity, tensor)
        self._done = False

    def __iter__(self):
        return self

    def _submit_one(self):
        if self._done:
            return False
        try:
            items = next(self._it)
        except StopIteration:
            self._done = True
            return False
        items = list(items)
        first = items[0]
        # the split of the keys (device tensors / host-side rest) is cached on the item beside its descriptor
        split = first.__dict__.get("_yolat_keysplit")
        names = tuple(k for k, v in first.__dict__.items() if v is not None and k[0] != "_")
        if split is None or split[0] != names:
            tkeys = [k for k in names if isinstance(first[k], torch.Tensor) and first[k].dim() > 0 and k in _DEVICE_KEYS]
            rest = [k for k in names if k not in tkeys]
            ship = tuple(k for k in tkeys if k not in _CSR_SKIP)
            split = first.__dict__["_yolat_keysplit"] = (names, tkeys, rest, ship)
        _, tkeys, rest, ship = split
        csr = self._csr
        if not csr:
            ship = tuple(tkeys)          # COO mode: the raw index tensors travel, fixed up by the native collate
        if len(items) == 1:
            ptrs = (ctypes.c_void_p * 1)(ctypes.addressof(_item_desc(first, ship, csr)))
        else:
            ptrs = (ctypes.c_void_p * len(items))(*[ctypes.addressof(_item_desc(it, ship, csr)) for it in items])
        rc = self._lib.yolat_loader_submit(self._h, ptrs, len(items))
        if rc != 0:
            self._check(rc, "yolat_loader_submit")
        try:
            loc = batch_locality(items)
        except (AttributeError, TypeError):      # items without the reference's edge / bbox_idx layout: nothing to vouch for
            loc = None
        self._pending.append((items, ship, tkeys, rest, ptrs, loc))
        return True

    def __next__(self):
        lib = self._lib
        stream = self._stream()
        if self._held is not None:       # the consumer is done ENQUEUEING on the previous batch — on the stream it drew it on
            rc = lib.yolat_loader_release(self._h, self._held, self._held_stream)
            self._held = None
            if rc != 0:
                self._check(rc, "yolat_loader_release")
        pending = self._pending
        while len(pending) < self._slots - 1 and self._submit_one():
            pass
        if not pending:
            raise StopIteration
        # (the entry — host items, descriptors, pointer array — stays referenced until the worker has handed the batch over:
        # it reads them until yolat_loader_next returns)
        ent = pending[0]
        items, ship, tkeys, rest, _ptrs, loc = ent
        out = self._LoaderBatch()
        rc = lib.yolat_loader_next(self._h, stream, ctypes.byref(out))
        pending.pop(0)
        if rc != 0:
            self._check(rc, "yolat_loader_next")
        slot = self._held = out.slot
        self._held_stream = stream
        while len(pending) < self._slots - 1 and self._submit_one():      # keep the worker busy under the forward
            pass
        B, nk = len(items), len(ship)
        total = out.total
        mem = self._mem.get(slot)
        if mem is None or mem[0] != out.device or mem[1] < total:
            t = torch.as_tensor(_DeviceMemory(out.device, total), device=self._device)
            mem = self._mem[slot] = (out.device, int(total), t, {})
        # Nothing but the prepared graph is made eagerly: the forward needs addresses only (x's travels as a number in
        # `_yolat_x`), and every tensor view / slices tensor / host-side key costs this thread 2-10 us of Python — a
        # hand-over is bound by exactly that.  The slices are copied out of the slot now (it is rewritten later).
        st = _BatchState(mem, out.off[:nk + 6], ctypes.string_at(out.slices, 8 * nk * (B + 1)), items, ship, tkeys, rest, B)
        batch = _LazyBatch()
        bd = batch.__dict__
        bd["_lazy"] = st
        slices = _LazySlices()
        slices._st = st                 # (one direction only: no reference cycle per batch)
        offs = st.offs
        N = out.N
        if self._csr:
            bd["_yolat_graph"] = self._from_buffer(mem[2], offs[nk:nk + 6], N, out.E, out.P)
        if self._csr and ship and ship[0] == "x":
            x0 = items[0].x
            if x0.dtype == torch.float32 and x0.dim() == 2:
                bd["_yolat_x"] = (out.device + offs[0], int(x0.shape[1]), N, self._device)
        if not self._csr:
            self._raw_fast_path(bd, items[0], ship, st, out.device)
        bd["_device_buffer"] = mem[2]
        bd["_yolat_loc"] = loc           # prepared graph / slot addresses: nothing the consumer edits in place
        bd["_loader"] = self             # the slot buffers are the loader's: a batch keeps it (and them) alive
        return batch, slices

    def _raw_fast_path(self, bd, first, ship, st, base):
        """COO mode: the eval forward's operands as addresses (architecture._stage / plan.run_raw) when the batch has the
        reference's layout: x [N, C] fp32, edge [E, 2] int64, e_attr [E, 4] fp32, bbox_idx [N] int64, bbox rows = proposals"""
        need = ("x", "edge", "e_attr", "bbox_idx", "bbox")
        if any(k not in ship for k in need):
            return
        d = first.__dict__
        x0, e0, a0, b0 = d["x"], d["edge"], d["e_attr"], d["bbox_idx"]
        if (x0.dtype != torch.float32 or x0.dim() != 2 or e0.dtype != torch.int64 or e0.dim() != 2 or e0.shape[1] != 2 or
                a0.dtype != torch.float32 or a0.dim() != 2 or a0.shape[1] != 4 or b0.dtype != torch.int64 or b0.dim() != 1):
            return
        B = st.B

        def rows(k):
            return _UNPACK_I64(st.sl_bytes, 8 * (ship.index(k) * (B + 1) + B))[0]

        def addr(k):
            return base + st.offs[ship.index(k)]
        N, E, P = rows("x"), rows("edge"), rows("bbox")
        if rows("bbox_idx") != N or rows("e_attr") != E:
            return
        bd["_yolat_raw"] = (addr("x"), int(x0.shape[1]), addr("edge"), 2, 1, addr("e_attr"), addr("bbox_idx"), N, E, P, self._device)

    def close(self):
        from ._lib import lib
        if getattr(self, "_h", None):
            if self._held is not None:
                lib.yolat_loader_release(self._h, self._held, self._held_stream)
                self._held = None
            # draw (and hand back) what the worker has already been given: it may hold pointers into the items
            from ._lib import LoaderBatch
            while self._pending:
                # the worker may still be reading this entry's items / descriptors (the header requires them alive until
                # yolat_loader_next has returned the batch): drop it only afterwards
                ent = self._pending[0]
                out = LoaderBatch()
                if lib.yolat_loader_next(self._h, ops_stream_or_zero(), ctypes.byref(out)) == 0:
                    lib.yolat_loader_release(self._h, int(out.slot), ops_stream_or_zero())
                self._pending.pop(0)
                del ent
            torch.cuda.synchronize(self._device)
            lib.yolat_loader_destroy(self._h)
            self._h = None
            self._mem.clear()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def ops_stream_or_zero():
    from . import ops
    try:
        return ops._stream()
    except Exception:
        return 0


def _collate_host_keys(data_list, rest, batch, slices):
    """host-side assembly of everything that is not a device tensor (roots ...), with the reference's index fix-up of
    host-resident edge tensors (train.py:244)"""
    first = data_list[0]
    B = len(data_list)
    host_items = [first.__class__(**{k: it[k] for k in rest}) for it in data_list]
    hb, hs = collate(host_items)
    for k in rest:
        batch[k] = hb[k]
        slices[k] = hs[k]
    node_slices = slices["pos"] if "pos" in slices else slices["x"]
    for k in rest:
        if "edge" in k and isinstance(batch[k], torch.Tensor) and batch[k].dtype == torch.long:
            for i in range(B):
                batch[k][int(slices[k][i]):int(slices[k][i + 1])] += node_slices[i]


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


def flatten_tree(data, slices):
    """The batch's proposal tree as flat int32 arrays for the one-submission predict (yolat_predict_tree): per root and
    per child the global proposal row and its (idx_pos, idx_edge) ranges with the image offsets added (arch:153-164,
    277-296), `child_ptr` [R + 1], `image_root_ptr` [B + 1].  O(#tree nodes) host work, once per batch (cached by the
    caller: the tree of a dataset item never changes, Datasets/graph_dict3.py:745-767)."""
    roots = data.roots
    slice_root = [int(v) for v in slices["roots"]]
    B = len(slice_root) - 1
    root_row, root_rng, child_ptr, child_row, child_rng, image_ptr = [], [], [0], [], [], [0]
    for i in range(B):
        off_pos, off_edge, off_bbox = int(slices["pos"][i]), int(slices["edge"][i]), int(slices["bbox"][i])
        int(slices["edge_super"][i])               # the reference reads it (KeyError if absent)
        for root in roots[slice_root[i]:slice_root[i + 1]]:
            v = root.value
            root_row.append(int(v["idx_bbox"]) + off_bbox)
            root_rng.append((v["idx_pos"][0] + off_pos, v["idx_pos"][1] + off_pos, v["idx_edge"][0] + off_edge,
                             v["idx_edge"][1] + off_edge))
            for ch in root.children:
                c = ch.value
                child_row.append(int(c["idx_bbox"]) + off_bbox)
                child_rng.append((c["idx_pos"][0] + off_pos, c["idx_pos"][1] + off_pos, c["idx_edge"][0] + off_edge,
                                  c["idx_edge"][1] + off_edge))
            child_ptr.append(len(child_row))
        image_ptr.append(len(root_row))

    def i32(a, shape):
        return np.asarray(a, dtype=np.int64).astype(np.int32).reshape(shape)
    R, C = len(root_row), len(child_row)
    return {"R": R, "Ctot": C, "B": B, "root_row": i32(root_row, (R,)), "root_range": i32(root_rng, (R, 4)),
            "child_ptr": i32(child_ptr, (R + 1,)), "child_row": i32(child_row, (C,)), "child_range": i32(child_rng, (C, 4)),
            "image_root_ptr": i32(image_ptr, (B + 1,))}


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
