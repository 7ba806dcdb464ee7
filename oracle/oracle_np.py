"""CPU ORACLE (naive loops) — TEST INFRASTRUCTURE ONLY.  Never imported by the product path.

Independent, deliberately naive Python/numpy restatement of the *index* semantics of the
hot path, used to cross-check both ``oracle_torch.py`` and the HIP kernels on small cases:
gather / scatter-mean / scatter-max (PyG + torch_scatter semantics),
the integer pre-processing the HIP library derives from ``edge_index`` / ``bbox_idx``
(CSR by destination, CSC by source, segment pointers — build design, SURVEY.md §8 b) and the
caller-side block-diagonal batching (cad_recognition/train.py:123-171,238-258).

Parity status: see ``oracle_torch.py`` (wiring pinned, third-party semantics restated).
Pure-Python loops: only for small cases.
"""
import numpy as np


# ---- third-party index semantics ------------------------------------

def gather_rows(x, idx):
    """index_select(0, idx): PyG __lift__ (called from torch_vertex.py:324)."""
    out = np.empty((len(idx), x.shape[1]), dtype=x.dtype)
    for e, i in enumerate(idx):
        out[e] = x[int(i)]
    return out


def scatter_mean(src, index, dim_size):
    """torch_scatter.scatter(reduce='mean'): sum in input order, / clamp(count, 1)."""
    out = np.zeros((dim_size, src.shape[1]), dtype=src.dtype)
    cnt = np.zeros(dim_size, dtype=src.dtype)
    for e, i in enumerate(index):
        out[int(i)] = out[int(i)] + src[e]          # fp32 adds, edge order
        cnt[int(i)] += 1
    cnt = np.maximum(cnt, 1)
    return out / cnt[:, None]


def scatter_max(src, index, dim_size):
    """torch_scatter.scatter(reduce='max') -> (out, arg); empty rows -> 0, arg = len(src);
    strict '>' update so the first occurrence wins (CPU kernel semantics)."""
    n, d = src.shape
    out = np.full((dim_size, d), np.finfo(src.dtype).min, dtype=src.dtype)
    arg = np.full((dim_size, d), n, dtype=np.int64)
    for e, i in enumerate(index):
        i = int(i)
        for c in range(d):
            if src[e, c] > out[i, c]:
                out[i, c] = src[e, c]
                arg[i, c] = e
    out[arg == n] = 0
    return out, arg


# ---- integer pre-processing of the HIP library (bit-exact contract) ---------------------

def coo_to_csr(src, dst, num_nodes):
    """Stable counting sort of the edges by destination.

    Returns row_ptr[N+1], perm[E] (CSR slot -> original edge id, ascending edge id inside a
    row), src_csr[E], dst_csr[E].  Stable => the fp32 summation order of the aggregation is
    the edge order, like torch_scatter's CPU scatter_add.
    """
    E = len(dst)
    row_ptr = np.zeros(num_nodes + 1, dtype=np.int32)
    for e in range(E):
        row_ptr[int(dst[e]) + 1] += 1
    for i in range(num_nodes):
        row_ptr[i + 1] += row_ptr[i]
    cursor = row_ptr[:-1].copy()
    perm = np.empty(E, dtype=np.int32)
    for e in range(E):
        d = int(dst[e])
        perm[cursor[d]] = e
        cursor[d] += 1
    src_csr = np.asarray(src, dtype=np.int64)[perm].astype(np.int32)
    dst_csr = np.asarray(dst, dtype=np.int64)[perm].astype(np.int32)
    return row_ptr, perm, src_csr, dst_csr


def csc_by_source(src_csr, num_nodes):
    """For the backward scatter to sources: col_ptr[N+1] and, per source node, the CSR slots
    (ascending) of the edges leaving it."""
    E = len(src_csr)
    col_ptr = np.zeros(num_nodes + 1, dtype=np.int32)
    for q in range(E):
        col_ptr[int(src_csr[q]) + 1] += 1
    for i in range(num_nodes):
        col_ptr[i + 1] += col_ptr[i]
    cursor = col_ptr[:-1].copy()
    slots = np.empty(E, dtype=np.int32)
    for q in range(E):
        s = int(src_csr[q])
        slots[cursor[s]] = q
        cursor[s] += 1
    return col_ptr, slots


def segment_ptr(bbox_idx, num_segments):
    """seg_ptr[P+1] from a non-decreasing bbox_idx (Datasets/graph_dict3.py:732,
    train.py:249-258): seg_ptr[p] = first row with bbox_idx >= p."""
    seg = np.zeros(num_segments + 1, dtype=np.int32)
    n = len(bbox_idx)
    r = 0
    for p in range(num_segments + 1):
        while r < n and int(bbox_idx[r]) < p:
            r += 1
        seg[p] = r
    return seg


# ---- caller-side batching (cad_recognition/train.py) -----------------------------------

def collate_fixup(items):
    """Block-diagonal batching of a list of dicts with keys x, pos, edge, e_attr, bbox_idx,
    bbox, labels: concatenate on dim 0 (train.py:123-171), then shift every key containing
    'edge' by the node offset of its image and 'bbox_idx' by the proposal (labels) offset
    (train.py:238-258).  Returns (batch dict, slices dict)."""
    keys = list(items[0].keys())
    out, slices = {}, {}
    for k in keys:
        slices[k] = [0]
        for it in items:
            slices[k].append(slices[k][-1] + len(it[k]))
        out[k] = np.concatenate([np.asarray(it[k]) for it in items], axis=0)
    for k in keys:
        if "edge" in k and "attr" not in k:
            for i in range(len(items)):
                out[k][slices[k][i]:slices[k][i + 1]] += slices["pos"][i]
        elif "bbox_idx" in k:
            for i in range(len(items)):
                out[k][slices[k][i]:slices[k][i + 1]] += slices["labels"][i]
    return out, {k: np.asarray(v, dtype=np.int64) for k, v in slices.items()}


# ---- whole conv layer, naive (eval-mode BN given as running stats) ----------------------

def conv_gp2_eval(x, x_node, src, dst, attr, p, eps=1e-5):
    """AttrRelativeEdgeConvGlobalPool2.forward in eval mode, loops only
    (gcn_lib/sparse/torch_vertex.py:319-337).  ``p`` maps state_dict-style keys
    ('nn.0.weight', 'nn.1.running_mean', 'lin_r.weight', 'mlp_node.0.weight', ...) to arrays."""
    def lin(v, w, b):
        return v @ w.T + b

    def bn(v, pre):
        return (v - p[pre + ".running_mean"]) / np.sqrt(p[pre + ".running_var"] + eps) \
            * p[pre + ".weight"] + p[pre + ".bias"]

    xi, xj = gather_rows(x, dst), gather_rows(x, src)
    f = np.concatenate([xi, xj - xi, attr], axis=1)
    h = np.maximum(bn(lin(f, p["nn.0.weight"], p["nn.0.bias"]), "nn.1"), 0)
    m = np.maximum(bn(lin(h, p["nn.3.weight"], p["nn.3.bias"]), "nn.4"), 0)
    out = scatter_mean(m.astype(np.float32), dst, x.shape[0]) + lin(x, p["lin_r.weight"], p["lin_r.bias"])
    xn = np.maximum(bn(lin(x_node, p["mlp_node.0.weight"], p["mlp_node.0.bias"]), "mlp_node.1"), 0)
    return out.astype(np.float32), xn.astype(np.float32)


def nms(boxes, scores, iou_threshold):
    """torchvision.ops.nms restated (third-party, absent from the image; installed unpinned next to pytorch 1.7.1 by
    the reference's deepgcn_env_install.sh:21; call site cad_recognition/train.py:105).  Published algorithm
    (torchvision/csrc/cpu/nms_kernel.cpp): visit boxes in descending score order; a box is kept unless an already
    kept box overlaps it with IoU > threshold, IoU = inter / (area_i + area_j - inter) in float32 with
    area = (x2-x1)*(y2-y1) and inter = max(0, min(x2)-max(x1)) * max(0, min(y2)-max(y1)).
    Ties in score keep ascending index order (torchvision's sort is unstable; fixtures avoid ties).
    Plain loops: TEST INFRASTRUCTURE for small cases."""
    b = np.asarray(boxes, dtype=np.float32)
    s = np.asarray(scores, dtype=np.float32)
    n = b.shape[0]
    order = np.argsort(-s, kind="stable")
    area = ((b[:, 2] - b[:, 0]) * (b[:, 3] - b[:, 1])).astype(np.float32)
    dead = np.zeros(n, dtype=bool)
    keep = []
    for a in range(n):
        i = order[a]
        if dead[i]:
            continue
        keep.append(i)
        rest = order[a + 1:]
        xx1 = np.maximum(b[i, 0], b[rest, 0]); yy1 = np.maximum(b[i, 1], b[rest, 1])
        xx2 = np.minimum(b[i, 2], b[rest, 2]); yy2 = np.minimum(b[i, 3], b[rest, 3])
        w = np.maximum(np.float32(0), (xx2 - xx1).astype(np.float32))
        h = np.maximum(np.float32(0), (yy2 - yy1).astype(np.float32))
        inter = (w * h).astype(np.float32)
        ovr = inter / ((area[i] + area[rest]).astype(np.float32) - inter).astype(np.float32)
        dead[rest[ovr > np.float32(iou_threshold)]] = True
    return np.asarray(keep, dtype=np.int64)
