"""Shared helpers for the golden fixtures (used by tests/golden/make_golden.py and the tests).

Weights are a pure function of (state_dict key, seed) so the reference modules, the oracle and the
HIP-backed modules get bit-identical parameters without shipping 6.4 MB of weights per fixture;
``state_hash`` pins them (a fixture stores the hash it was generated with).
"""
import hashlib
import zlib

import numpy as np
import torch


def fill_state_(model, seed):
    """Deterministically overwrite every entry of ``model.state_dict()``:
    Linear weight ~ N(0, 2/fan_in) (kaiming-normal-like), Linear bias ~ 0.05 N(0,1),
    BatchNorm weight ~ U(0.5,1.5), bias ~ 0.1 N, running_mean ~ 0.2 N, running_var ~ U(0.5,1.5)."""
    sd = model.state_dict()
    with torch.no_grad():
        for name, t in sd.items():
            g = torch.Generator().manual_seed((seed * 1000003 + zlib.crc32(name.encode())) % (2 ** 31 - 1))
            if name.endswith("num_batches_tracked"):
                t.zero_()
            elif name.endswith("running_mean"):
                t.copy_(0.2 * torch.randn(t.shape, generator=g))
            elif name.endswith("running_var"):
                t.copy_(0.5 + torch.rand(t.shape, generator=g))
            elif t.dim() == 2:
                t.copy_(torch.randn(t.shape, generator=g) * (2.0 / t.shape[1]) ** 0.5)
            elif name.endswith("weight"):          # 1-D weight = BatchNorm gamma
                t.copy_(0.5 + torch.rand(t.shape, generator=g))
            else:                                   # bias: Linear bias or BatchNorm beta
                # BatchNorm bias sits right after a 1-D weight; both cases get a small normal
                t.copy_(0.1 * torch.randn(t.shape, generator=g))
    return model


def state_hash(model):
    h = hashlib.sha256()
    for name, t in model.state_dict().items():
        h.update(name.encode())
        h.update(t.detach().cpu().contiguous().numpy().tobytes())
    return h.hexdigest()


def summarize(t, max_elems=4096):
    """Compact, order-sensitive fingerprint of a tensor: full copy if small, else a strided sample
    plus sum / abs-sum (float64)."""
    a = t.detach().cpu().double().numpy().reshape(-1)
    if a.size <= max_elems:
        return {"full": a.astype(np.float32)}
    stride = a.size // max_elems + 1
    return {"sample": a[::stride].astype(np.float32), "stride": np.int64(stride),
            "sum": np.float64(a.sum()), "abssum": np.float64(np.abs(a).sum())}


def compare_summary(name, got, ref, rtol, atol):
    """Assert tensor ``got`` matches a stored summary ``ref`` (dict of arrays from an npz)."""
    a = got.detach().cpu().double().numpy().reshape(-1)
    if "full" in ref:
        np.testing.assert_allclose(a, ref["full"].astype(np.float64), rtol=rtol, atol=atol, err_msg=name)
        return
    stride = int(ref["stride"])
    np.testing.assert_allclose(a[::stride], ref["sample"].astype(np.float64), rtol=rtol, atol=atol,
                               err_msg=name + " (sample)")
    scale = max(float(ref["abssum"]), 1e-30)
    assert abs(a.sum() - float(ref["sum"])) <= rtol * scale + atol * a.size, name + " (sum)"
    assert abs(np.abs(a).sum() - float(ref["abssum"])) <= rtol * scale + atol * a.size, name + " (abssum)"


def pack(prefix, summary, out):
    for k, v in summary.items():
        out["%s/%s" % (prefix, k)] = v


def unpack(prefix, npz):
    pre = prefix + "/"
    return {k[len(pre):]: npz[k] for k in npz.files if k.startswith(pre)}


def graph_case(kind):
    """Inputs of the float golden cases as numpy arrays (deterministic)."""
    rng = np.random.default_rng({"tiny": 11, "small": 12, "medium": 13, "deep": 14}[kind])
    if kind == "tiny":
        # N=6, E=7, P=2; node 0 has in-degree 0; edge (1->2) appears twice; proposals {0,1,2},{3,4,5}
        edge = np.array([[0, 1], [1, 2], [1, 2], [0, 2], [3, 4], [5, 4], [4, 5]], dtype=np.int64)
        bbox_idx = np.array([0, 0, 0, 1, 1, 1], dtype=np.int64)
        N, P = 6, 2
    else:
        P, lo, hi, epp = {"small": (6, 6, 16, None), "medium": (40, 25, 25, 100), "deep": (5, 5, 12, None)}[kind]
        n_p = rng.integers(lo, hi + 1, size=P)
        N = int(n_p.sum())
        starts = np.concatenate([[0], np.cumsum(n_p)])[:-1]
        bbox_idx = np.repeat(np.arange(P), n_p).astype(np.int64)
        e_p = np.ceil(3.1 * n_p).astype(np.int64) if epp is None else np.full(P, epp)
        owner = np.repeat(np.arange(P), e_p)
        a = rng.integers(0, 1 << 30, size=len(owner)) % n_p[owner]
        b = rng.integers(0, 1 << 30, size=len(owner)) % (n_p[owner] - 1)
        b = np.where(b >= a, b + 1, b)
        edge = np.stack([starts[owner] + a, starts[owner] + b], axis=1).astype(np.int64)
        edge = edge[rng.permutation(len(edge))]
    E = len(edge)
    x = np.zeros((N, 5), dtype=np.float32)
    x[:, 3:5] = rng.random((N, 2)).astype(np.float32) * 1.6 - 0.3
    e_attr = (rng.standard_normal((E, 4)) * 0.05).astype(np.float32)
    e_attr[rng.random(E) < 0.6] = 0.0
    K = 22 if kind == "deep" else 17
    labels = rng.integers(0, K, size=P).astype(np.int64)
    bbox = rng.random((P, 4)).astype(np.float32)
    opt = dict(n_classes=K, n_blocks=4 if kind == "deep" else 2, n_blocks_out=2)
    return dict(x=x, edge=edge, e_attr=e_attr, bbox_idx=bbox_idx, labels=labels, bbox=bbox,
                stat_feats=np.zeros((P, 13), np.float32)), opt


def to_data(arrs, data_cls, dtype=torch.float32):
    d = data_cls(x=torch.from_numpy(arrs["x"].copy()).to(dtype), pos=torch.from_numpy(arrs["x"][:, 3:5].copy()))
    d.edge = torch.from_numpy(arrs["edge"].copy())
    d.e_attr = torch.from_numpy(arrs["e_attr"].copy()).to(dtype)
    d.bbox_idx = torch.from_numpy(arrs["bbox_idx"].copy())
    d.bbox = torch.from_numpy(arrs["bbox"].copy())
    d.stat_feats = torch.from_numpy(arrs["stat_feats"].copy())
    d.labels = torch.from_numpy(arrs["labels"].copy())
    d.is_super = torch.zeros(arrs["x"].shape[0], dtype=torch.bool)
    return d


GRAD_KEYS_FULL = ("cls_net.head.", "cls_net.backbone.")   # conv-layer params: stored in full


def run_case(model, loss_mod, data, lr=2.5e-4, wd=1e-5):
    """eval forward, then one training step (train.py:263-284) with torch.optim.Adam on CPU modules.
    Returns a dict name -> tensor of everything a fixture records."""
    out = {}
    model.eval()
    with torch.no_grad():
        out["eval_logits"] = model(data, None)[0].clone()
    model.train()
    optim = torch.optim.Adam(model.parameters(), lr=lr, weight_decay=wd)
    optim.zero_grad()
    res = model(data, None)
    loss = loss_mod(res, data)["loss"]
    loss.backward()
    out["train_logits"] = res[0].detach().clone()
    out["loss"] = loss.detach().reshape(1).clone()
    for n, p in model.named_parameters():
        out["grad/" + n] = p.grad.detach().clone()
    optim.step()
    for n, p in model.named_parameters():
        out["param_after/" + n] = p.detach().clone()
    for n, b in model.named_buffers():
        if not n.endswith("num_batches_tracked"):
            out["buffer_after/" + n] = b.detach().clone()
    return out


PREDICT_CASE = dict(n_graphs=2, seed=77, num_proposals=14, nodes_lo=4, nodes_hi=9, edge_factor=1.5,
                    n_classes=3, with_roots=True)
PREDICT_OPT = dict(n_classes=3, n_blocks=2, n_blocks_out=2)


def predict_case(synth_batch, dtype=torch.float32):
    """The two-pass inference fixture's inputs: 2 collated synthetic items with proposal trees."""
    data, slices = synth_batch(**PREDICT_CASE)
    data.x = data.x.to(dtype)
    data.e_attr = data.e_attr.to(dtype)
    return data, slices


def input_checksum(data):
    return np.array([float(data.x.double().sum()), float(data.edge.sum()), float(data.e_attr.double().abs().sum()),
                     float(data.bbox_idx.sum()), float(len(data.roots))], dtype=np.float64)
