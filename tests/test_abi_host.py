"""CPU tests (-m "not gpu"): the C-ABI library loads and exports everything include/yolat_hip.h
declares (no compute calls without a GPU), and the host-side mirror of the reference interface."""
import ctypes
import os
import re

import numpy as np
import pytest

pytestmark = pytest.mark.host      # host code: CPU suite, and also the GPU box's -m gpu pass (conftest.py)
import torch

import yolat_vectorgraphicsrecognition_amd as yv
from yolat_vectorgraphicsrecognition_amd import _lib
from oracle import oracle_np as onp

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _header_decls():
    src = open(os.path.join(REPO, "include", "yolat_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    decls = {}
    for m in re.finditer(r"\b(?:int|size_t|int64_t|void|const char\*|yolat_loader\*)\s+(yolat_\w+)\s*\(([^;]*?)\)\s*;", src, flags=re.S):
        args = m.group(2).strip()
        n = 0 if args in ("", "void") else len([a for a in args.split(",") if a.strip()])
        decls[m.group(1)] = n
    return decls


def test_library_exports_every_declared_symbol():
    decls = _header_decls()
    assert len(decls) >= 28
    lib = ctypes.CDLL(_lib.LIB_PATH)
    for name in decls:
        assert hasattr(lib, name), "libyolat_hip.so does not export %s" % name


def test_product_library_carries_no_debug_stamps():
    """The wall-clock stamps of tools/exp/r06_*_stamps.* exist only in debug builds (-DYOLAT_{FX,H8,EDGE}_STAMPS): the
    product library must not export their read-back entry points (a build with the stamps compiled in times differently)."""
    lib = ctypes.CDLL(_lib.LIB_PATH)
    for name in ("yolat_debug_fx_stamps", "yolat_debug_h8_stamps", "yolat_debug_edge_stamps"):
        assert not hasattr(lib, name), "%s is exported: the library was built with debug stamps" % name


def test_ctypes_signatures_match_header():
    decls = _header_decls()
    assert set(decls) == set(_lib.SIGNATURES), set(decls) ^ set(_lib.SIGNATURES)
    for name, n in decls.items():
        assert len(_lib.SIGNATURES[name][1]) == n, name


def test_abi_version_and_strerror():
    assert _lib.lib.yolat_abi_version() == 6
    assert b"invalid" in _lib.lib.yolat_strerror(-1)
    assert _lib.lib.yolat_strerror(0) == b"ok"


def test_invalid_arguments_are_rejected_without_a_gpu():
    # argument validation happens before any launch, so this is safe on a CPU-only box
    rc = _lib.lib.yolat_linear_fwd(None, 4, 8, 0, None, None, 0, None, 4, None, 4, None, None, 0, None, 4, 0,
                                   None, None)
    assert rc == -1
    rc = _lib.lib.yolat_coo_to_csr(None, 2, 1, 5, 0, None, None, None, None, None, None, None)
    assert rc == -1
    with pytest.raises(_lib.YolatLibraryError):
        _lib.check(rc, "yolat_coo_to_csr")


def test_ops_refuse_cpu_tensors():
    a = torch.zeros(4, 8)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        yv.ops.linear_fwd(a, torch.zeros(3, 8), None, torch.zeros(4, 3))


def test_state_dict_keys_match_reference(golden_dir):
    want = [l.split()[0] for l in open(os.path.join(golden_dir, "state_dict_keys.txt"))]
    shapes = {l.split()[0]: l.split()[1] if len(l.split()) > 1 else "" for l in open(os.path.join(golden_dir, "state_dict_keys.txt"))}
    m = yv.SparseCADGCN(yv.Opt())
    sd = m.state_dict()
    assert list(sd.keys()) == want
    for k, v in sd.items():
        assert "x".join(str(s) for s in v.shape) == shapes[k], k
    assert sum(p.numel() for p in m.parameters()) == 1613329
    assert sum(p.numel() for p in yv.SparseCADGCN(yv.Opt(n_classes=22)).parameters()) == 1614614
    assert sum(p.numel() for p in yv.SparseCADGCN(yv.Opt(n_blocks=4)).parameters()) == 1656081


def test_unknown_conv_act_norm_raise_like_reference():
    with pytest.raises(NotImplementedError, match="conv foo is not implemented"):
        yv.GraphConv(5, 64, "foo")
    from yolat_vectorgraphicsrecognition_amd.nn_modules import act_layer, norm_layer
    with pytest.raises(NotImplementedError, match="is not found"):
        act_layer("swish")
    with pytest.raises(NotImplementedError, match="is not found"):
        norm_layer("group", 4)


def test_multiseq_is_indexable_and_splats_tuples():
    class Two(torch.nn.Module):
        def forward(self, a, b=None):
            return (a + 1, a + 2) if b is None else a + b
    ms = yv.MultiSeq(Two(), Two())
    assert isinstance(ms[0], Two)
    assert float(ms(torch.zeros(1))) == 3.0


def test_collate_and_fixup_match_fixture(golden_dir):
    z = np.load(os.path.join(golden_dir, "collate.npz"))
    items = []
    for i in range(3):
        d = yv.Data(x=torch.from_numpy(z["item%d/x" % i]), pos=torch.from_numpy(z["item%d/pos" % i]))
        for k in ("edge", "e_attr", "bbox_idx", "bbox", "labels"):
            d[k] = torch.from_numpy(z["item%d/%s" % (i, k)].copy())
        items.append(d)
    data, slices = yv.collate(items)
    yv.fixup_offsets(data, slices)
    for k in ("x", "pos", "edge", "e_attr", "bbox_idx", "bbox", "labels"):
        np.testing.assert_array_equal(data[k].numpy(), z["batch/" + k], err_msg=k)
        np.testing.assert_array_equal(slices[k].numpy(), z["slices/" + k], err_msg=k)


def test_synthetic_graph_invariants():
    d = yv.synth_graph(num_proposals=50, nodes_lo=4, nodes_hi=40, seed=3)
    N, E, P = d.x.shape[0], d.edge.shape[0], d.bbox.shape[0]
    assert d.x.shape[1] == 5 and torch.all(d.x[:, :3] == 0)
    assert torch.all(d.bbox_idx[1:] >= d.bbox_idx[:-1]) and int(d.bbox_idx[-1]) == P - 1
    assert torch.all(d.edge[:, 0] != d.edge[:, 1])
    assert torch.all(d.bbox_idx[d.edge[:, 0]] == d.bbox_idx[d.edge[:, 1]])      # edges stay inside a proposal
    assert 0.7 < float((d.e_attr.abs().sum(1) == 0).float().mean()) < 0.95
    seg = onp.segment_ptr(d.bbox_idx.numpy(), P)
    for p in range(P):
        pos = d.x[seg[p]:seg[p + 1], 3:5]
        assert float(pos.min()) == 0.0 and float(pos.max()) == 1.0
    assert d.labels.shape[0] == P and d.e_attr.shape == (E, 4)


def test_named_configs_have_the_stated_sizes():
    d, s, kw, n = yv.config("2")
    assert (d.x.shape[0], d.edge.shape[0], d.bbox.shape[0], n) == (10000, 40000, 400, 1)
    assert kw["n_blocks"] == 2


def test_dropin_shims_resolve_the_reference_imports(golden_dir):
    """dropin/ first on sys.path: the reference's own import statements (architecture3cc_rpn_gp_iter2.py:6-9,
    train.py:16-18,30) resolve to the MI355X implementation, and the model built through them has the reference's
    state_dict keys."""
    import subprocess
    import sys
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    dropin = os.path.join(repo, "yolat_vectorgraphicsrecognition_amd", "dropin")
    code = r"""
import torch_geometric as tg
from gcn_lib.sparse import MultiSeq, MLP, GraphConv, PlainDynBlock, ResBlock, DenseDynBlock, DilatedKnnGraph
from torch_scatter import scatter
from torch_geometric.data import Data
import torch_geometric.transforms as T
from torch_geometric.nn.data_parallel import DataParallel
from torch_geometric.data import InMemoryDataset
from architecture3cc_rpn_gp_iter2 import SparseCADGCN, DetectionLoss
import yolat_vectorgraphicsrecognition_amd as yv
assert SparseCADGCN is yv.SparseCADGCN and GraphConv is yv.GraphConv and Data is yv.Data and scatter is yv.scatter
m = SparseCADGCN(yv.Opt())
print("\n".join(m.state_dict().keys()))
for cls in (PlainDynBlock, DenseDynBlock, DilatedKnnGraph, DataParallel, InMemoryDataset):
    try:
        cls()
    except NotImplementedError:
        pass
    else:
        raise SystemExit("%s must not be instantiable" % cls.__name__)
d = Data(x=1)
assert d.x == 1 and "x" in d.keys
"""
    env = dict(os.environ)
    env["PYTHONPATH"] = os.pathsep.join([dropin, repo, env.get("PYTHONPATH", "")])
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    want = [l.split()[0] for l in open(os.path.join(golden_dir, "state_dict_keys.txt")) if l.strip()]
    assert out.stdout.split() == want


@pytest.mark.parametrize("prefix", ["", "module."])
def test_reference_format_checkpoint_loads(tmp_path, prefix):
    """A checkpoint shaped like the reference's (train.py:313-321), with and without the `module.` prefix that
    utils/ckpt_util.py:51-64 strips, loads into SparseCADGCN and reproduces every tensor."""
    import sys
    import yolat_vectorgraphicsrecognition_amd as yv
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import golden_util as gu
    from oracle import oracle_torch as orc
    ref = gu.fill_state_(orc.SparseCADGCN(orc.Opt()), 5)
    sd = {prefix + k: v.clone() for k, v in ref.state_dict().items()}
    path = str(tmp_path / "ckpt_3.pth")
    torch.save({"epoch": 3, "state_dict": sd, "best_value": 0.5}, path)
    model = yv.SparseCADGCN(yv.Opt())
    epoch, best = yv.load_reference_checkpoint(model, path)
    assert epoch == 3 and best == 0.5
    own = model.state_dict()
    assert list(own.keys()) == list(ref.state_dict().keys())
    for k, v in ref.state_dict().items():
        assert torch.equal(own[k], v), k
    with pytest.raises(RuntimeError):
        bad = dict(sd)
        bad.pop(next(iter(bad)))
        yv.load_reference_checkpoint(yv.SparseCADGCN(yv.Opt()), {"state_dict": bad})


def test_reference_checkpoint_with_numpy_best_value_and_optimizer_state_loads_from_path(tmp_path):
    """What train.py:313-321 really writes: `best_value` is a numpy.float64 (max(np.mean(AP), ...), train.py:311,508)
    next to optimizer and scheduler state dicts — a full pickle, which torch >= 2.6 refuses under its default
    weights_only=True.  load_reference_checkpoint(path) must read it."""
    import sys
    import yolat_vectorgraphicsrecognition_amd as yv
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import golden_util as gu
    from oracle import oracle_torch as orc
    ref = gu.fill_state_(orc.SparseCADGCN(orc.Opt()), 7)
    adam = torch.optim.Adam(ref.parameters(), lr=2.5e-4, weight_decay=1e-5)
    sched = torch.optim.lr_scheduler.StepLR(adam, 20, 0.5)
    best = max(np.mean([0.25, 0.75]), np.float64(0.1))
    assert isinstance(best, np.float64)
    path = str(tmp_path / "ckpt_7.pth")
    torch.save({"epoch": 7, "state_dict": ref.state_dict(), "optimizer_state_dict": adam.state_dict(),
                "scheduler_state_dict": sched.state_dict(), "best_value": best}, path)
    model = yv.SparseCADGCN(yv.Opt())
    epoch, got_best = yv.load_reference_checkpoint(model, path)
    assert epoch == 7 and float(got_best) == 0.5
    for k, v in ref.state_dict().items():
        assert torch.equal(model.state_dict()[k], v), k


def test_reference_checkpoint_saved_under_numpy_1_loads_with_the_safe_unpickler(tmp_path):
    """The reference's own checkpoints were written under numpy 1.x: their pickle names `numpy.core.multiarray.scalar`
    (numpy 2.x writes `numpy._core.multiarray.scalar`).  The safe loader must accept both spellings without
    trust_pickle."""
    import sys
    import yolat_vectorgraphicsrecognition_amd as yv
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import golden_util as gu
    from oracle import oracle_torch as orc
    ref = gu.fill_state_(orc.SparseCADGCN(orc.Opt()), 8)
    path = str(tmp_path / "ckpt_np1.pth")
    torch.save({"epoch": 3, "state_dict": ref.state_dict(), "best_value": np.float64(0.625)}, path,
               _use_new_zipfile_serialization=False)
    raw = open(path, "rb").read()
    new_name, old_name = b"numpy._core.multiarray\nscalar", b"numpy.core.multiarray\nscalar"
    if new_name in raw:                      # running numpy 2.x: rewrite the GLOBAL to the numpy 1.x module path
        raw = raw.replace(new_name, old_name)
    assert old_name in raw
    open(path, "wb").write(raw)
    model = yv.SparseCADGCN(yv.Opt())
    epoch, best = yv.load_reference_checkpoint(model, path)
    assert epoch == 3 and float(best) == 0.625
    for k, v in ref.state_dict().items():
        assert torch.equal(model.state_dict()[k], v), k


class _Evil(object):
    def __reduce__(self):
        return (os.system, ("echo pwned > /dev/null",))


def test_reference_checkpoint_loader_does_not_unpickle_arbitrary_objects(tmp_path):
    """Reference checkpoints are third-party downloads: load_reference_checkpoint(path) uses torch's safe unpickler (numpy's
    scalar reconstructors allow-listed, previous test) and REFUSES a file whose pickle would call into arbitrary code;
    only trust_pickle=True takes the full unpickler."""
    import yolat_vectorgraphicsrecognition_amd as yv
    model = yv.SparseCADGCN(yv.Opt())
    path = str(tmp_path / "evil.pth")
    torch.save({"epoch": 1, "state_dict": model.state_dict(), "best_value": _Evil()}, path)
    with pytest.raises(RuntimeError, match="safe unpickler"):
        yv.load_reference_checkpoint(yv.SparseCADGCN(yv.Opt()), path)
    epoch, _ = yv.load_reference_checkpoint(yv.SparseCADGCN(yv.Opt()), path, trust_pickle=True)
    assert epoch == 1


def test_item_csr_cache_is_rebuilt_when_the_item_changes():
    """data.item_csr caches the destination-sorted form ON the dataset item; an in-place edit or a re-assignment of the
    arrays it was computed from (edge / e_attr / bbox_idx) must invalidate the entry, not ship a stale graph."""
    import yolat_vectorgraphicsrecognition_amd as yv
    from yolat_vectorgraphicsrecognition_amd import data as ydata
    item, _ = yv.synth_batch(1, 5, num_proposals=6, nodes_lo=4, nodes_hi=9)
    c0 = ydata.item_csr(item)
    assert ydata.item_csr(item) is c0                          # unchanged item: the cached entry
    item.x.mul_(0.5)                                           # node features / boxes do not enter the CSR: a normalisation
    item.bbox.add_(0.25)                                       # or augmentation of them keeps the sorted graph
    assert ydata.item_csr(item) is c0
    attr0 = c0["attr"].copy()
    item.e_attr.mul_(2.0)                                      # in place (an augmentation)
    c1 = ydata.item_csr(item)
    assert c1 is not c0 and np.array_equal(c1["attr"], 2.0 * attr0)
    item.edge = item.edge.flip(1).contiguous()                 # re-assigned: every edge reversed
    c2 = ydata.item_csr(item)
    assert c2 is not c1 and np.array_equal(np.sort(c2["dst"][:c2["E"]]), np.sort(c1["src"][:c1["E"]]))


# ---------------------------------------------------------------------------------------------
# native host side of the batch hand-over (csrc/collate.hip)
# ---------------------------------------------------------------------------------------------
def _host_item_csr(src, dst, N, attr=None, bbox_idx=None, P=1):
    E = len(src)
    edge = torch.from_numpy(np.stack([src, dst], 1).astype(np.int64)) if E else torch.zeros(0, 2, dtype=torch.long)
    attr = torch.from_numpy(attr) if attr is not None else torch.zeros(E, 4)
    i32 = lambda n: np.zeros(max(n, 1), dtype=np.int32)
    out = dict(row_ptr=i32(N + 1), perm=i32(E), src=i32(E), dst=i32(E), attr=np.zeros((max(E, 1), 4), np.float32),
               seg_ptr=i32(P + 1), node_seg=i32(N))
    status = np.zeros(1, np.int32)
    bb = torch.from_numpy(bbox_idx.astype(np.int64)) if bbox_idx is not None else None
    _lib.check(_lib.lib.yolat_item_csr_host(edge.data_ptr(), 2, 1, attr.data_ptr(), bb.data_ptr() if bb is not None else None, E, N,
                                            P, out["row_ptr"].ctypes.data, out["perm"].ctypes.data, out["src"].ctypes.data,
                                            out["dst"].ctypes.data, out["attr"].ctypes.data, out["seg_ptr"].ctypes.data,
                                            out["node_seg"].ctypes.data, status.ctypes.data), "yolat_item_csr_host")
    return out, int(status[0])


@pytest.mark.parametrize("case", ["csr_a", "csr_b", "csr_c"])
def test_item_csr_host_matches_the_integer_fixture_and_the_loop_oracle(golden_dir, case):
    """yolat_item_csr_host (the host twin of yolat_graph_prepare): stable sort by destination — bit-exact against the
    committed integer fixture and the naive-loop oracle; e_attr travels with its edge."""
    z = np.load(os.path.join(golden_dir, "integer_ops.npz"))
    src, dst, N = z[case + "/src"], z[case + "/dst"], int(z[case + "/N"])
    attr = np.random.default_rng(3).standard_normal((len(src), 4)).astype(np.float32)
    out, status = _host_item_csr(src, dst, N, attr)
    assert status == 0
    E = len(src)
    np.testing.assert_array_equal(out["row_ptr"][:N + 1], z[case + "/row_ptr"])
    np.testing.assert_array_equal(out["perm"][:E], z[case + "/perm"])
    np.testing.assert_array_equal(out["src"][:E], z[case + "/src_csr"])
    np.testing.assert_array_equal(out["dst"][:E], z[case + "/dst_csr"])
    np.testing.assert_array_equal(out["attr"][:E], attr[z[case + "/perm"]])
    rp, perm, s_csr, d_csr = onp.coo_to_csr(src, dst, N)
    np.testing.assert_array_equal(out["row_ptr"][:N + 1], rp)
    np.testing.assert_array_equal(out["perm"][:E], perm)


def test_item_csr_host_segments_and_status_flags(golden_dir):
    z = np.load(os.path.join(golden_dir, "integer_ops.npz"))
    bb, P = z["seg/bbox_idx"], int(z["seg/P"])
    out, status = _host_item_csr(np.array([0, 1]), np.array([1, 0]), len(bb), bbox_idx=bb, P=P)
    assert status == 0
    np.testing.assert_array_equal(out["seg_ptr"][:P + 1], z["seg/seg_ptr"])
    np.testing.assert_array_equal(out["node_seg"][:len(bb)], bb.astype(np.int32))
    # the device kernels' flags, on the host: ids outside [0, N) are clamped and reported; unsorted bbox_idx reported
    out, status = _host_item_csr(np.array([0, 9]), np.array([1, 0]), 3, bbox_idx=np.array([0, 0, 0]), P=1)
    assert status & yv.ops.STATUS_EDGE_RANGE and out["src"][:2].max() <= 2
    _, status = _host_item_csr(np.array([0]), np.array([1]), 3, bbox_idx=np.array([0, 1, 0]), P=2)
    assert status & yv.ops.STATUS_SEG_UNSORTED
    _, status = _host_item_csr(np.array([0]), np.array([1]), 3, bbox_idx=np.array([0, 1, 5]), P=2)
    assert status & yv.ops.STATUS_SEG_RANGE
    item = yv.Data(x=torch.zeros(3, 5), pos=torch.zeros(3, 2), edge=torch.tensor([[0, 7]]), e_attr=torch.zeros(1, 4),
                   bbox_idx=torch.tensor([0, 0, 0]), bbox=torch.zeros(1, 4), labels=torch.zeros(1, dtype=torch.long))
    with pytest.raises(IndexError):
        yv.item_csr(item)


def test_collate_csr_pack_equals_csr_of_the_collated_batch(golden_dir):
    """Block-diagonal merge: the items' CSRs concatenated with offsets == the CSR of the reference-collated, fixed-up
    batch (tests/golden/collate.npz is the output of the reference's OWN collate + fix-up loop), bit for bit; and
    yolat_collate_pack == torch.cat of every key."""
    z = np.load(os.path.join(golden_dir, "collate.npz"))
    keys = ("x", "pos", "edge", "e_attr", "bbox_idx", "bbox", "labels")
    items = [yv.Data(**{k: torch.from_numpy(z["item%d/%s" % (i, k)].copy()) for k in keys}) for i in range(3)]
    cs = [yv.item_csr(it) for it in items]
    assert yv.item_csr(items[0]) is cs[0]                        # cached on the item
    Nt, Et, Pt = (sum(c[k] for c in cs) for k in ("N", "E", "P"))
    i32 = lambda n: np.zeros(n, dtype=np.int32)
    out = dict(row_ptr=i32(Nt + 1), src=i32(Et), dst=i32(Et), attr=np.zeros((Et, 4), np.float32), seg_ptr=i32(Pt + 1),
               node_seg=i32(Nt))
    arr = (_lib.ItemCsr * 3)(*[c["struct"] for c in cs])
    _lib.check(_lib.lib.yolat_collate_csr_pack(arr, 3, out["row_ptr"].ctypes.data, out["src"].ctypes.data,
                                               out["dst"].ctypes.data, out["attr"].ctypes.data, out["seg_ptr"].ctypes.data,
                                               out["node_seg"].ctypes.data), "yolat_collate_csr_pack")
    # reference: the CSR of the golden batch (edge already offset-fixed by the reference's loop)
    be, ba, bb = z["batch/edge"], z["batch/e_attr"], z["batch/bbox_idx"]
    want, status = _host_item_csr(be[:, 0], be[:, 1], Nt, ba, bbox_idx=bb, P=Pt)
    assert status == 0
    for k in ("row_ptr", "src", "dst", "attr", "seg_ptr", "node_seg"):
        n = out[k].shape[0]
        np.testing.assert_array_equal(out[k], want[k][:n], err_msg=k)
    rp, perm, s_csr, d_csr = onp.coo_to_csr(be[:, 0], be[:, 1], Nt)
    np.testing.assert_array_equal(out["row_ptr"], rp)
    np.testing.assert_array_equal(out["src"], s_csr)
    np.testing.assert_array_equal(out["seg_ptr"], onp.segment_ptr(bb, Pt))
    # yolat_collate_pack: every key of every item back to back
    buf = np.zeros(4096, np.uint8)
    spans = (_lib.Span * (len(keys) * 3))()
    foff = (ctypes.c_int64 * len(keys))()
    off = 0
    for f, k in enumerate(keys):
        foff[f] = off
        for i, it in enumerate(items):
            spans[f * 3 + i].ptr, spans[f * 3 + i].bytes = it[k].data_ptr(), it[k].numel() * it[k].element_size()
        off += z["batch/" + k].nbytes if k != "edge" else sum(it[k].numel() * 8 for it in items)
    _lib.check(_lib.lib.yolat_collate_pack(buf.ctypes.data, foff, spans, len(keys), 3), "yolat_collate_pack")
    for f, k in enumerate(keys):
        want_k = np.concatenate([it[k].numpy() for it in items], 0)
        got = buf[foff[f]:foff[f] + want_k.nbytes].view(want_k.dtype).reshape(want_k.shape)
        np.testing.assert_array_equal(got, want_k, err_msg=k)


def test_item_locality_host_against_a_numpy_restatement():
    """yolat_item_locality_host (data.item_locality / batch_locality): flags and sizes against a direct numpy restatement of
    the property — bbox_idx[dst] non-decreasing along the edge list (grouped by proposal, Datasets/graph_dict3.py:725,752-764),
    both end points in the same proposal (:582-600,733), nodes / edges of the largest proposal — on clean items, an item with
    a crossing edge, an ungrouped edge list and malformed ids; merged over a batch by OR / max."""
    from yolat_vectorgraphicsrecognition_amd import data as D

    def numpy_record(it):
        bb, e = it.bbox_idx.numpy(), it.edge.numpy()
        N, P = it.x.shape[0], it.bbox.shape[0]
        flags = 0
        if ((bb < 0) | (bb >= P)).any() or (np.diff(bb) < 0).any() or ((e < 0) | (e >= N)).any():
            flags |= 4
        ec = np.clip(e, 0, N - 1)
        owner = np.clip(bb, 0, P - 1)[ec]
        if (owner[:, 0] != owner[:, 1]).any():
            flags |= 2
        if (np.diff(owner[:, 1]) < 0).any():
            flags |= 1
        return flags, int(np.bincount(np.clip(bb, 0, P - 1), minlength=P).max()), \
            int(np.bincount(owner[:, 1], minlength=P).max()) if len(e) else 0

    items = [yv.synth_graph(num_proposals=30 + 7 * j, nodes_lo=3, nodes_hi=15 + j, edge_factor=1.6, seed=300 + j) for j in range(4)]
    crossing = yv.synth_graph(num_proposals=25, nodes_lo=3, nodes_hi=12, seed=310)
    crossing.edge = crossing.edge.clone()
    crossing.edge[3, 0] = crossing.x.shape[0] - 1
    ungrouped = yv.synth_graph(num_proposals=25, nodes_lo=3, nodes_hi=12, seed=311)
    perm = torch.randperm(ungrouped.edge.shape[0], generator=torch.Generator().manual_seed(1))
    ungrouped.edge = ungrouped.edge[perm].contiguous()
    malformed = yv.synth_graph(num_proposals=25, nodes_lo=3, nodes_hi=12, seed=312)
    malformed.edge = malformed.edge.clone()
    malformed.edge[0, 1] = malformed.x.shape[0] + 3
    for it in items + [crossing, ungrouped, malformed]:
        got = D.item_locality(it)
        want = numpy_record(it)
        assert got[0] == want[0] and got[1] == want[1], (got, want)
        if not got[0] & 1:
            assert got[2] == want[2]
    loc = D.batch_locality(items)
    recs = [numpy_record(it) for it in items]
    assert (loc.known, loc.flags, loc.max_nodes, loc.max_edges) == (1, 0, max(r[1] for r in recs), max(r[2] for r in recs))
    assert D.batch_locality(items + [crossing]).flags == 2
    assert D.batch_locality([ungrouped] + items).flags & 1
    # yolat_conv_local_fits: what the forward decides from the record (tile shapes 64 / 512 below 2048 proposals, 128 / 1024 from there)
    from yolat_vectorgraphicsrecognition_amd._lib import Locality
    for P, n, e, fits in ((100, 64, 512, 1), (100, 65, 512, 0), (100, 64, 513, 0), (4000, 128, 1024, 1), (4000, 129, 10, 0)):
        assert _lib.lib.yolat_conv_local_fits(ctypes.byref(Locality(1, 0, n, e)), P) == fits
    assert _lib.lib.yolat_conv_local_fits(ctypes.byref(Locality(1, 1, 8, 8)), 4000) == 0
    assert _lib.lib.yolat_conv_local_fits(None, 4000) == 0


def test_flatten_tree_lists_what_select_tree_ranges_walks():
    """data.flatten_tree (the one-submission predict's tree, uploaded once) against data.select_tree_ranges (the two-pass
    walk of arch:153-164 / :277-296): the same global ranges and proposal rows, roots in order, every root's children behind
    child_ptr, per-image root offsets."""
    from yolat_vectorgraphicsrecognition_amd import data as D
    data, slices = yv.synth_batch(3, 41, num_proposals=40, nodes_lo=3, nodes_hi=12, edge_factor=1.4, with_roots=True)
    ft = D.flatten_tree(data, slices)
    ps, pe, es, ee, rows, image_off = D.select_tree_ranges(data, slices)
    assert ft["R"] == len(rows) == len(data.roots) and ft["B"] == 3
    np.testing.assert_array_equal(ft["root_row"], np.asarray(rows))
    np.testing.assert_array_equal(ft["root_range"], np.stack([ps, pe, es, ee], 1))
    np.testing.assert_array_equal(ft["image_root_ptr"], np.asarray(image_off))
    has = np.ones(len(data.roots), dtype=bool)
    ps, pe, es, ee, rows, image_off = D.select_tree_ranges(data, slices, has)
    assert ft["Ctot"] == len(rows) == int(ft["child_ptr"][-1])
    np.testing.assert_array_equal(ft["child_row"], np.asarray(rows))
    np.testing.assert_array_equal(ft["child_range"], np.stack([ps, pe, es, ee], 1))
    assert [int(ft["child_ptr"][i + 1] - ft["child_ptr"][i]) for i in range(ft["R"])] == [len(r.children) for r in data.roots]
