"""GPU tests (-m gpu) of the native batch loader (csrc/loader.hip, data.DeviceLoader): collate + offset fix-up + merged CSR +
the one host -> device copy of a batch on a native worker thread, a ring of pinned / device slots, events instead of host
synchronisation.  Reference behaviour: cad_recognition/train.py:123-171,238-258 (collate + fix-up; the bytes are those of
yolat_collate_batch, pinned by tests/test_abi_host.py against the reference's own collate) behind
DataLoader(num_workers=8), train.py:178-189."""
import numpy as np
import pytest
import torch

import golden_util as gu

pytestmark = pytest.mark.gpu


def _yv():
    import yolat_vectorgraphicsrecognition_amd as yv
    return yv


def _lists(yv, n_lists, seed0=0, with_roots=False):
    out = []
    for i in range(n_lists):
        items = [yv.synth_graph(num_proposals=5 + 7 * ((i + j) % 4), nodes_lo=3, nodes_hi=9 + 3 * j, edge_factor=1.3,
                                seed=seed0 + 10 * i + j, with_roots=with_roots) for j in range(1 + i % 3)]
        out.append(items)
    return out


def _same_batch(yv, got, want):
    (gb, gs), (wb, ws) = got, want
    assert sorted(gs.keys()) == sorted(ws.keys())
    for k in ws:
        assert torch.equal(torch.as_tensor(gs[k]), torch.as_tensor(ws[k])), k
    for k in wb.keys:
        a, b = gb[k], wb[k]
        if isinstance(b, torch.Tensor):
            assert a.dtype == b.dtype and a.shape == b.shape and a.device.type == b.device.type, k
            assert torch.equal(a, b), k
    gg, wg = gb._yolat_graph, wb._yolat_graph
    assert (gg.N, gg.E, gg.P) == (wg.N, wg.E, wg.P)
    for name in ("row_ptr", "src", "dst", "attr", "seg_ptr", "node_seg"):
        assert torch.equal(getattr(gg, name), getattr(wg, name)), name


def test_device_loader_batches_are_bit_identical_to_collate_to_device():
    """ragged batches of 1..3 items of different sizes (so that the ring's buffers grow), in order; tensors, slices and the
    merged destination-sorted graph equal what collate_to_device(items, csr=True) ships"""
    yv = _yv()
    lists = _lists(yv, 9)
    # a late, much larger batch: the slot it lands in must be re-allocated
    lists.append([yv.synth_graph(num_proposals=300, nodes_lo=4, nodes_hi=30, seed=991)])
    lists += _lists(yv, 3, seed0=500)
    loader = yv.DeviceLoader(lists, slots=3)
    n = 0
    for (batch, slices), items in zip(loader, lists):
        want = yv.collate_to_device(items, csr=True)
        _same_batch(yv, (batch, slices), want)
        n += 1
    assert n == len(lists)
    with pytest.raises(StopIteration):
        next(loader)
    loader.close()


def test_device_loader_forward_soak_1000_batches_no_stale_or_overwritten_buffers():
    """1000 batches cycling over 7 different item lists through a 3-slot ring with a forward on each: every batch's logits
    equal the logits of the same items handed over synchronously — a slot rewritten while its forward is still in flight,
    or read before its copy has landed, shows up as a mismatch"""
    yv = _yv()
    lists = _lists(yv, 7, seed0=40)
    model = gu.fill_state_(yv.SparseCADGCN(yv.Opt()), 5).cuda().eval()
    want = []
    with torch.no_grad():
        for items in lists:
            b, sl = yv.collate_to_device(items, csr=True)
            want.append(model(b, sl)[0].clone())
    torch.cuda.synchronize()
    seq = [i % len(lists) for i in range(1000)]
    loader = yv.DeviceLoader((lists[i] for i in seq), slots=3)
    outs = []
    with torch.no_grad():
        for k, (batch, slices) in enumerate(loader):
            outs.append((seq[k], model(batch, slices)[0]))      # no synchronisation between batches
    torch.cuda.synchronize()
    assert len(outs) == 1000
    bad = [k for k, (i, o) in enumerate(outs) if not torch.equal(o, want[i])]
    assert not bad, "batches %s differ from the synchronous hand-over" % bad[:10]
    loader.close()


def test_device_loader_close_with_batches_in_flight_and_host_keys():
    """the loader can be dropped with submitted batches not drawn (worker joined, slots handed back); non-tensor keys
    (`roots`) are assembled on the host like collate does"""
    yv = _yv()
    lists = _lists(yv, 6, seed0=77, with_roots=True)
    loader = yv.DeviceLoader(lists, slots=4)
    batch, slices = next(loader)
    wb, ws = yv.collate_to_device(lists[0], csr=True)
    assert len(batch.roots) == len(wb.roots) and torch.equal(torch.as_tensor(slices["roots"]), torch.as_tensor(ws["roots"]))
    loader.close()
    loader.close()                 # idempotent
    # a second loader right after: fresh worker, fresh ring
    l2 = yv.DeviceLoader(lists[:2], slots=2)
    assert sum(1 for _ in l2) == 2
    l2.close()


def test_device_loader_round_robin_over_three_streams_soak():
    """consecutive batches drawn (and consumed) under three different streams: the slot of a batch is handed back behind an
    event on ITS stream — 600 batches, no host synchronisation, logits equal the synchronous hand-over"""
    yv = _yv()
    lists = _lists(yv, 5, seed0=140)
    model = gu.fill_state_(yv.SparseCADGCN(yv.Opt()), 7).cuda().eval()
    want = []
    with torch.no_grad():
        for items in lists:
            b, sl = yv.collate_to_device(items, csr=True)
            want.append(model(b, sl)[0].clone())
    torch.cuda.synchronize()
    streams = [torch.cuda.Stream() for _ in range(3)]
    seq = [(3 * i + i // 7) % len(lists) for i in range(600)]
    loader = yv.DeviceLoader((lists[i] for i in seq), slots=5)
    outs = []
    try:
        with torch.no_grad():
            k = 0
            torch.cuda.set_stream(streams[0])
            for batch, slices in loader:
                outs.append((seq[k], model(batch, slices)[0]))
                k += 1
                torch.cuda.set_stream(streams[k % 3])
    finally:
        torch.cuda.set_stream(torch.cuda.default_stream())
    torch.cuda.synchronize()
    assert len(outs) == 600
    bad = [k for k, (i, o) in enumerate(outs) if not torch.equal(o, want[i])]
    assert not bad, "batches %s differ from the synchronous hand-over" % bad[:10]
    loader.close()


def test_device_loader_coo_mode_matches_collate_to_device_and_the_reference_fixup():
    """csr=False: the raw edge / e_attr / bbox_idx tensors travel, with the offset fix-up of train.py:238-258 applied by the
    native worker while it copies them — tensors equal collate_to_device(items) (device-side fix-up) AND host collate +
    fixup_offsets (the reference's loops); the forward (CSR rebuilt on the device) gives the same logits"""
    yv = _yv()
    lists = _lists(yv, 8, seed0=300)
    model = gu.fill_state_(yv.SparseCADGCN(yv.Opt()), 11).cuda().eval()
    loader = yv.DeviceLoader(lists, slots=3, csr=False)
    n = 0
    with torch.no_grad():
        for (batch, slices), items in zip(loader, lists):
            wb, ws = yv.collate_to_device(items)
            hb, hs = yv.collate(items)
            hb = yv.fixup_offsets(hb, hs)
            assert getattr(batch, "_yolat_graph", None) is None
            for k in ("x", "pos", "edge", "e_attr", "bbox_idx", "bbox", "stat_feats", "labels"):
                if k in wb.keys:
                    a = batch[k]
                    assert a.is_cuda and a.dtype == wb[k].dtype and a.shape == wb[k].shape, k
                    assert torch.equal(a, wb[k]), k
                    assert torch.equal(a.cpu(), hb[k]), k
                    assert torch.equal(torch.as_tensor(slices[k]), torch.as_tensor(ws[k])), k
            got = model(batch, slices)[0]
            want = model(wb, ws)[0]
            assert torch.equal(got, want)
            n += 1
    assert n == len(lists)
    loader.close()


def test_predict_on_a_coo_mode_loader_batch_equals_predict_on_the_synchronous_batch():
    """two-pass inference (arch:139-356) cuts sub-graphs out of the raw edge list: it runs on a csr=False loader batch (lazy
    tensor views, lazily collated `roots`) and returns what it returns for collate_to_device(items)"""
    yv = _yv()
    lists = _lists(yv, 3, seed0=410, with_roots=True)
    model = gu.fill_state_(yv.SparseCADGCN(yv.Opt()), 13).cuda().eval()
    loader = yv.DeviceLoader(lists, slots=3, csr=False)
    with torch.no_grad():
        for (batch, slices), items in zip(loader, lists):
            wb, ws = yv.collate_to_device(items)
            got = model.predict(batch, slices)
            want = model.predict(wb, ws)
            assert torch.equal(got[0], want[0]) and torch.equal(got[1], want[1])
            assert list(got[3]) == list(want[3]) and list(got[4]) == list(want[4])
    loader.close()


@pytest.mark.parametrize("csr", [True, False])
def test_training_step_on_a_loader_batch_equals_the_step_on_the_synchronous_batch(csr):
    """Trainer.step reads x / labels / the graph (prepared or raw) off the lazy batch: same loss and same updated parameters as
    on collate_to_device's batch"""
    yv = _yv()
    items = _lists(yv, 3, seed0=520)[2]
    opt = yv.Opt()

    def run(batch, slices):
        model = gu.fill_state_(yv.SparseCADGCN(opt), 17).cuda()
        tr = yv.Trainer(model, opt, lr=1e-3, weight_decay=1e-5)
        loss = float(tr.step(batch, slices))
        torch.cuda.synchronize()
        return loss, [p.detach().clone() for p in model.parameters()]

    loader = yv.DeviceLoader([items], slots=2, csr=csr)
    batch, slices = next(loader)
    l1, p1 = run(batch, slices)
    loader.close()
    wb, ws = yv.collate_to_device(items, csr=csr)
    l2, p2 = run(wb, ws)
    assert l1 == l2
    assert all(torch.equal(a, b) for a, b in zip(p1, p2))


# ----------------------------------------------------------------------------------------------------------------------
# Round 6: the locality record travels with the batch (decided on the host from the items, data.batch_locality), and the
# lifetime / shortcut rules of loader batches.
# ----------------------------------------------------------------------------------------------------------------------

def _stage_names(fn):
    import ctypes
    from yolat_vectorgraphicsrecognition_amd._lib import lib
    lib.yolat_profile_reset()
    lib.yolat_profile_enable(1)
    try:
        out = fn()
        torch.cuda.synchronize()
    finally:
        lib.yolat_profile_enable(0)
    names = []
    buf = ctypes.create_string_buffer(128)
    ms, calls, fl, by = ctypes.c_float(), ctypes.c_int(), ctypes.c_double(), ctypes.c_double()
    for i in range(lib.yolat_profile_count()):
        lib.yolat_profile_get(i, buf, 128, ctypes.byref(ms), ctypes.byref(calls), ctypes.byref(fl), ctypes.byref(by))
        names.append(buf.value.decode())
    lib.yolat_profile_reset()
    return out, names


def test_host_locality_of_items_equals_the_device_examination_of_the_batch():
    """data.batch_locality (yolat_item_locality_host per item, merged) against yolat_batch_locality on the collated batch:
    the same flags and sizes, for clean items, an item with a crossing edge and an item whose edge list is not grouped"""
    import ctypes
    yv = _yv()
    from yolat_vectorgraphicsrecognition_amd._lib import lib, check
    from yolat_vectorgraphicsrecognition_amd import data as D
    items = [yv.synth_graph(num_proposals=40 + 5 * j, nodes_lo=3, nodes_hi=20 + j, edge_factor=1.7, seed=60 + j) for j in range(4)]
    bad1 = yv.synth_graph(num_proposals=30, nodes_lo=3, nodes_hi=12, seed=70)
    bad1.edge = bad1.edge.clone()
    bad1.edge[4, 0] = bad1.x.shape[0] - 1
    bad2 = yv.synth_graph(num_proposals=30, nodes_lo=3, nodes_hi=12, seed=71)
    bad2.edge = torch.cat([bad2.edge[1:], bad2.edge[:1]], 0)
    bad2.e_attr = torch.cat([bad2.e_attr[1:], bad2.e_attr[:1]], 0)
    for lst in (items, items[:1], [items[0], bad1, items[1]], [bad2, items[2]]):
        host = D.batch_locality(lst)
        batch, _ = yv.collate_to_device(lst)
        N, E, P = batch.x.shape[0], batch.edge.shape[0], batch.bbox.shape[0]
        ws = torch.empty(int(lib.yolat_batch_locality_workspace_bytes(N, E, P)) + 16, dtype=torch.uint8, device="cuda")
        info = torch.empty(4, dtype=torch.int32, device="cuda")
        check(lib.yolat_batch_locality(batch.edge.data_ptr(), 2, 1, batch.bbox_idx.data_ptr(), N, E, P, info.data_ptr(),
                                       ws.data_ptr(), ws.numel(), torch.cuda.current_stream().cuda_stream), "locality")
        flags, mn, me, st = info.tolist()
        assert (host.known, host.flags, host.max_nodes) == (1, flags, mn) and st == 0
        if not flags & 1:            # (per-proposal edge RANGES only exist for a grouped list)
            assert host.max_edges == me
    # cached on the item, invalidated by an in-place edit of the edge list
    rec = D.item_locality(items[0])
    assert D.item_locality(items[0]) is rec
    items[0].edge[0, 0] = items[0].x.shape[0] - 1
    assert D.item_locality(items[0])[0] & 2


@pytest.mark.parametrize("csr", [True, False])
def test_bf16_forward_on_loader_batches_takes_the_one_launch_stack_without_gated_launches(csr):
    """a bf16 model on DeviceLoader / collate_to_device batches of >= 1024 proposals: the locality record comes with the
    batch, so the forward is local prep (COO mode) or nothing (CSR mode) + ONE conv launch — no gated fall-back launches, no
    device-side examination — and its logits equal the resident forward's bit for bit"""
    yv = _yv()
    import os
    lists = [[yv.synth_graph(num_proposals=420 + 30 * i + 10 * j, nodes_lo=3, nodes_hi=25, edge_factor=1.6,
                             seed=900 + 10 * i + j) for j in range(3)] for i in range(4)]
    model = gu.fill_state_(yv.SparseCADGCN(yv.Opt(n_classes=17, n_blocks=3, n_blocks_out=2)), 2).cuda().eval()
    model.set_eval_precision("bf16")
    n = 0
    for (batch, slices), items in zip(yv.DeviceLoader(lists, slots=3, csr=csr), lists):
        with torch.no_grad():
            examined = len(model.__dict__["_yolat_plan"]._loc) if "_yolat_plan" in model.__dict__ else 0
            got, names = _stage_names(lambda: model(batch, slices)[0].clone())
            sync, names_sync = _stage_names(lambda: model(*yv.collate_to_device(items, csr=csr))[0].clone())
            assert len(model._yolat_plan._loc) == examined          # the record came with the batch: nothing examined
            host = yv.collate(items)
            yv.fixup_offsets(*host)
            for k in ("x", "edge", "e_attr", "bbox_idx", "bbox"):
                setattr(host[0], k, getattr(host[0], k).cuda())
            want = model(*host)[0].clone()
        for nm in (names, names_sync):
            assert any(s.startswith("conv_local") for s in nm), nm
            assert not any("gated" in s for s in nm), nm
        assert torch.equal(got, want) and torch.equal(sync, want)
        n += 1
    assert n == len(lists)
    model._yolat_plan.check_status()


def test_a_loader_batch_outlives_the_loop_that_drew_it_and_sees_reassigned_tensors():
    """(1) the last batch of `for batch, sl in DeviceLoader(...)` is used after the loop — its tensors live in
    slot buffers the loader owns, so the batch keeps the loader alive; (2) a caller that re-assigns a shipped key before the
    forward gets the forward on ITS tensor, not on the slot data captured at draw time."""
    import gc
    yv = _yv()
    lists = _lists(yv, 4)
    model = gu.fill_state_(yv.SparseCADGCN(yv.Opt()), 3).cuda().eval()
    for csr in (True, False):
        last = None
        for batch, slices in yv.DeviceLoader(lists, slots=2, csr=csr):
            last = (batch, slices)
        gc.collect()
        torch.cuda.synchronize()
        junk = [torch.full((1 << 20,), 7.0, device="cuda") for _ in range(8)]        # recycle whatever was freed
        with torch.no_grad():
            got = model(*last)[0].clone()
            want = model(*yv.collate_to_device(lists[-1], csr=csr))[0]
        assert torch.equal(got, want)
        del junk
        # (2) re-assignment of x: the forward reads the new tensor
        batch, slices = last
        batch.x = batch.x * 0.5
        assert "_yolat_x" not in batch.__dict__ and "_yolat_raw" not in batch.__dict__
        ref_b, ref_s = yv.collate_to_device(lists[-1], csr=csr)
        ref_b.x = ref_b.x * 0.5
        with torch.no_grad():
            assert torch.equal(model(batch, slices)[0], model(ref_b, ref_s)[0])
