"""MI355X-native ``cad_recognition/architecture3cc_rpn_gp_iter2.py``.

Same classes and signatures as the reference model file:

    Backbone(opt).forward(x, edges, edge_weights, edge_attrs, bbox_idx)       arch:15-71
    SparseCADGCN(opt).forward(data, slices) -> (pred_cls, pred_bbox)          arch:73-137
    SparseCADGCN.predict(data, slices) -> 6-tuple                             arch:139-356
    DetectionLoss(opt)(out, data) -> {'loss', 'loss_cls'}                     arch:358-379

``SparseCADGCN.forward`` moves the CPU batch to the GPU itself (like the reference's ``.cuda()`` calls,
arch:107-115), derives the CSR / segment structure on the device (ops.build_graph) and runs the
hand-scheduled HIP sequence of engine.model_fwd; in training mode the result carries a custom
autograd node whose backward is engine.model_bwd.  Module attribute names reproduce the reference
state_dict keys so reference checkpoints load.
"""
import torch
from torch import nn

from . import engine, ops
from .engine import GradSink
from .nn_modules import MultiSeq, MLP, GraphConv, ResBlock, scatter, graph_for
from .data import Data


class Backbone(nn.Module):
    def __init__(self, opt, n_edges=3, edge_max_pool=None):
        super(Backbone, self).__init__()
        channels = opt.n_filters
        act, norm, bias = opt.act, opt.norm, opt.bias
        conv = "attr_edge_gp2"          # hard-coded in the reference too (arch:22); opt.conv is ignored
        c_growth = channels
        self.n_edges = 1
        self.n_blocks = opt.n_blocks
        self.n_blocks_out = opt.n_blocks_out
        self.heads = nn.ModuleList()
        self.n_classes = opt.n_classes
        self.class_specific = opt.class_specific
        self.head = GraphConv(opt.in_channels, channels, conv, act, norm, bias)
        self.backbone = MultiSeq(*[ResBlock(channels, conv, act, norm, bias)
                                   for _ in range(self.n_blocks - 1)])
        fusion_dims = int(channels + c_growth * (self.n_blocks_out - 1))
        self.fusion_block = MLP([fusion_dims, 1024], act, norm, bias)
        self.fusion_block_super = MLP([fusion_dims, 1024], act, norm, bias)
        self.fusion_dims = fusion_dims

    def forward(self, x, edges, edge_weights, edge_attrs, bbox_idx):
        """Module-by-module path (each sub-module launches its own HIP kernels and has its own
        autograd node).  SparseCADGCN.forward does not go through here: it uses the fused
        schedule in engine.model_fwd."""
        f, f_super = self.head(x, edges[0], edge_weights[0], edge_attrs[0], x_node=x)
        feats, feats_super = [f], [f_super]
        for i in range(self.n_blocks - 1):
            f, f_super = self.backbone[i](feats[-1], edges[0], edge_weights[0], edge_attrs[0],
                                          x_node=feats_super[-1])
            feats.append(f)
            feats_super.append(f_super)
        lo = self.n_blocks - self.n_blocks_out
        feats = torch.cat(feats[lo:self.n_blocks], dim=1)
        out_feat = torch.cat((self.fusion_block(feats), feats), dim=1)
        feats_super = torch.cat(feats_super[lo:self.n_blocks], dim=1)
        feats_super = scatter(feats_super, bbox_idx, dim=0, reduce="mean")
        out_feat_super = torch.cat((self.fusion_block_super(feats_super), feats_super), dim=1)
        return out_feat, out_feat_super


class _ModelFn(torch.autograd.Function):
    """One autograd node for the whole SparseCADGCN forward."""

    @staticmethod
    def forward(ctx, model, g, x, *params):
        training = model.training
        logits, sv = engine.model_fwd(model, g, x, training)
        ctx.model, ctx.g, ctx.sv, ctx.params = model, g, sv, params
        # the saved state holds the logits buffer itself (the last Linear's output): returning that very tensor object
        # would close a cycle output -> grad_fn -> ctx -> sv -> output, and a step's saved activations (0.6 GB at cfg 4,
        # several GB at cfg 5) would wait for Python's cyclic collector instead of being freed with the loss — up to
        # seven steps of garbage in the caching allocator, and hipMalloc calls in the middle of timed steps
        return logits.detach() if sv is not None else logits

    @staticmethod
    def backward(ctx, dlogits):
        if ctx.sv is None:
            if getattr(ctx, "released", False):
                raise RuntimeError("second backward through the same SparseCADGCN forward: the saved activations are "
                                   "released after the first one (retain_graph is not supported by the HIP path)")
            raise RuntimeError("backward through an eval-mode SparseCADGCN is not supported by the HIP path")
        model = ctx.model
        flat = getattr(model, "_yolat_flat", None)
        dl = dlogits.contiguous()          # read only: the logits layer has no BatchNorm, nothing below writes into it
        sv, g, params = ctx.sv, ctx.g, ctx.params
        ctx.released = True
        ctx.sv = ctx.g = None           # one backward per forward (as retain_graph=False means): release the activations now
        if flat is not None and flat.direct_grads:
            # trainer path: gradients land in the flat buffer, autograd sees no per-tensor grads
            engine.model_bwd(model, g, sv, dl, GradSink(flat.grad_views, getattr(flat, "on_head_done", None)))
            flat.grads_ready = True
            return (None, None, None) + tuple(None for _ in params)
        sink = engine.model_bwd(model, g, sv, dl, GradSink())
        return (None, None, None) + tuple(sink.out.get(id(p)) for p in params)


# predict(): one forward over the whole batch + integer selection on the device (csrc/subgraph.hip, yolat_predict_select);
# False: always the two-pass sub-graph extraction (module flag; the tests run both)
PREDICT_ONE_SUBMISSION = True


class SparseCADGCN(nn.Module):
    def __init__(self, opt, n_edges=3, edge_max_pool=None, expand_ratio=0.25):
        super(SparseCADGCN, self).__init__()
        self.expand_ratio = expand_ratio
        act, norm, bias = opt.act, opt.norm, opt.bias
        self.n_classes = opt.n_classes
        self.classifier = opt.classifier
        self.class_specific = opt.class_specific
        self.dim_stat = 0
        self.cls_net = Backbone(opt)
        d = (self.cls_net.fusion_dims + 1024) * 2 + self.dim_stat
        self.prediction_cls = MultiSeq(*[MLP([d, 512], act, norm, bias),
                                         MLP([512, 256], act, norm, bias, drop=opt.dropout),
                                         MLP([256, opt.n_classes], None, None, bias)])
        self.model_init()

    def model_init(self):
        """arch:97-104: kaiming_normal_ on every Linear weight, zero bias."""
        for m in self.modules():
            if isinstance(m, nn.Linear):
                nn.init.kaiming_normal_(m.weight)
                m.weight.requires_grad = True
                if m.bias is not None:
                    m.bias.data.zero_()
                    m.bias.requires_grad = True

    # -- device staging ---------------------------------------------------------------------
    @staticmethod
    def _stage(data, need_graph=True):
        """H2D of the tensors forward reads (arch:107-115), cached on the batch object so a second
        forward on the same batch (predict, epochs over a cached batch) re-uses them; the device-side
        CSR / segment structure is added lazily for the training path (the eval plan builds its own)."""
        pre = data.__dict__.get("_yolat_graph") if hasattr(data, "__dict__") else None
        if pre is not None:
            # the batch carries a prepared graph (data.collate_to_device(csr=True)); x / bbox are device tensors already
            xref = data.__dict__.get("_yolat_x") if not need_graph else None
            if xref is not None:
                # a DeviceLoader batch in the eval forward: x's address / row stride / rows / device, no tensor view
                return {"x": None, "xref": xref, "bbox": data.bbox, "g": pre, "prepared": True,
                        "loc": data.__dict__.get("_yolat_loc")}
            x = data.x if data.x.dtype == torch.float32 else data.x.float()
            return {"x": x, "bbox": data.bbox, "g": pre, "prepared": True, "loc": data.__dict__.get("_yolat_loc")}
        raw = data.__dict__.get("_yolat_raw") if (not need_graph and hasattr(data, "__dict__")) else None
        if raw is not None:
            # a DeviceLoader batch in COO mode in the eval forward: addresses only (plan.run_raw), no tensor views
            return {"raw": raw, "bbox": data.bbox, "g": None, "loc": data.__dict__.get("_yolat_loc")}
        st = SparseCADGCN._stage_tensors(data)
        if need_graph and st["g"] is None:
            st["g"] = ops.build_graph(st["edge"], st["e_attr"], st["bbox_idx"], st["x"].shape[0],
                                      st["bbox"].shape[0])
        return st

    @staticmethod
    def _stage_tensors(data):
        """the H2D half of `_stage`: device copies of x / edge / e_attr / bbox_idx / bbox, cached on the batch object and
        invalidated by in-place edits (`_version`) or re-assignment (`data_ptr`) of any of them"""
        cache = getattr(data, "_yolat_stage", None)
        key = (data.x.data_ptr(), data.x._version, data.edge.data_ptr(), data.edge._version, data.bbox_idx.data_ptr(),
               data.bbox_idx._version, data.e_attr.data_ptr(), data.e_attr._version, data.bbox.data_ptr(),
               data.bbox._version, tuple(data.x.shape), tuple(data.edge.shape))
        if cache is None or cache[0] != key:
            x = data.x.cuda(non_blocking=True)
            if x.dtype != torch.float32:
                x = x.float()
            st = {"x": x, "edge": data.edge.cuda(non_blocking=True), "e_attr": data.e_attr.cuda(non_blocking=True),
                  "bbox_idx": data.bbox_idx.cuda(non_blocking=True), "bbox": data.bbox.cuda(non_blocking=True),
                  "g": None}
            cache = (key, st)
            try:
                data._yolat_stage = cache
            except AttributeError:
                pass
        return cache[1]

    def forward(self, data, slices=None):
        """arch:106-137.  Preconditions the dataset guarantees (Datasets/graph_dict3.py:594-600,732):
        `data.bbox_idx` non-decreasing, edge ids inside [0, N).  Violations are flagged in a device status word
        (ops.STATUS_*) and the kernels stay memory-safe; `check_last_status()` (synchronising) raises on them —
        `predict` and `Trainer.step`'s first call check it where the host synchronises anyway."""
        if not self.training and not torch.is_grad_enabled():
            # eval fast path: one call into libyolat_hip.so (plan.EvalPlan / yolat_forward_eval)
            st = self._stage(data, need_graph=False)
            # one plan (folded weights, workspace, status word) per launch stream: independent forwards
            # issued on different streams overlap on the GPU (no kernel of a 10k-node graph fills 256 CUs)
            plans = self.__dict__.setdefault("_yolat_plans", {})
            sid = ops._stream()
            plan = plans.get(sid)
            if plan is None:
                from .plan import EvalPlan
                plan = plans[sid] = EvalPlan(self, self.__dict__.get("_yolat_precision", "fp32"))
            ug = self.__dict__.get("_yolat_use_graph")
            if ug is not None:
                plan.use_graph = ug
            self.__dict__["_yolat_plan"] = plan      # the plan of the most recent forward (status checks); not through
            #                                          nn.Module.__setattr__: 3 us of a 100 us hand-over
            if st.get("raw") is not None:
                pred_cls = plan.run_raw(st["raw"], st.get("loc"))
            elif st.get("prepared"):
                pred_cls = plan.run_prepared(st["x"], st["g"], st.get("xref"), st.get("loc"))
            else:
                # a batch from collate_to_device carries its locality record, decided on the host from the items: valid
                # for exactly the (edge, bbox_idx) tensors it was taken for
                loc = data.__dict__.get("_yolat_loc") if hasattr(data, "__dict__") else None
                if loc is not None:
                    e_t, b_t = st["edge"], st["bbox_idx"]
                    loc = loc[1] if (isinstance(loc, tuple) and
                                     loc[0] == (e_t.data_ptr(), e_t._version, b_t.data_ptr(), b_t._version)) else None
                pred_cls = plan.run(st["x"], st["edge"], st["e_attr"], st["bbox_idx"], st["bbox"].shape[0], loc)
            st["plan_status"] = plan
        else:
            st = self._stage(data)
            self._yolat_plan = st["g"]       # ops.Graph carries the status word of the training path
            pred_cls = _ModelFn.apply(self, st["g"], st["x"], *list(self.parameters()))
        if self.classifier != "softmax":
            pred_cls = torch.sigmoid(pred_cls)
        return pred_cls, st["bbox"]

    def check_last_status(self):
        """Synchronising check of the most recent forward's input-validity flags (raises IndexError / ValueError)."""
        last = self.__dict__.get("_yolat_plan")
        return True if last is None else last.check_status()

    def predict(self, data, slices):
        """arch:139-356.  Eval mode, softmax classifier, a batch with its raw edge list: ONE forward over the whole batch
        + integer selection on the device (`_predict_one_submission`: no host round trip between the two passes of the
        reference, one read at the end); when the tree of the batch is not made of the ranges of its own proposals — or in
        any other mode — the two-pass extraction (`_predict_two_pass`).  Same 6-tuple either way."""
        if PREDICT_ONE_SUBMISSION and not self.training and self.classifier == "softmax":
            out = self._predict_one_submission(data, slices)
            if out is not None:
                return out
        return self._predict_two_pass(data, slices)

    def _predict_one_submission(self, data, slices):
        """One forward over the whole batch, then yolat_predict_select / yolat_predict_gather (csrc/subgraph.hip): in eval
        mode the logits of a proposal depend only on its own nodes and edges, so the root pass and the child pass of the
        reference are rows of the same forward; `has_object` (:259-281), the per-image interleaving (:317-328) and the 5 %
        box enlargement (:341-346) are device kernels.  Returns None when the shortcut does not apply (the device found
        tree ranges that are not the ranges of the proposals, or an edge between proposals: the duplicate / KeyError
        cases of the reference) — the caller then runs the two-pass extraction."""
        import ctypes
        import numpy as np
        from ._lib import lib, check, PredictTree
        from .data import flatten_tree
        d = data.__dict__ if hasattr(data, "__dict__") else {}
        if d.get("_yolat_graph") is not None:
            return None                                  # prepared-CSR batch: no raw edge list to validate the tree against
        with torch.no_grad():
            logits, bbox = self.forward(data, slices)
        raw = d.get("_yolat_raw")
        if raw is not None:
            _, _, ep, se, sc, _, bp, N, E, P, dev = raw
        else:
            st = self._stage_tensors(data)
            edge, bidx = st["edge"], st["bbox_idx"]
            if edge.dim() != 2 or edge.shape[1] != 2:
                return None
            ep, bp = ops._i(edge, torch.int64, "edge"), ops._i(bidx, torch.int64, "bbox_idx")
            se, sc = edge.stride(0), edge.stride(1)
            N, E, P, dev = st["x"].shape[0], edge.shape[0], bbox.shape[0], logits.device
        # the flattened tree, on the device once per batch object (keyed by the identity of the tree and the image offsets)
        key = (id(data.roots), len(data.roots), tuple(int(v) for v in slices["roots"]), tuple(int(v) for v in slices["pos"]),
               tuple(int(v) for v in slices["edge"]), tuple(int(v) for v in slices["bbox"]))
        ent = d.get("_yolat_tree")
        if ent is None or ent[0] != key:
            ft = flatten_tree(data, slices)
            order = ("root_row", "root_range", "child_ptr", "child_row", "child_range", "image_root_ptr")
            flat = np.concatenate([ft[k].reshape(-1) for k in order])
            buf = torch.from_numpy(flat).to(dev, non_blocking=True)
            offs, o = {}, 0
            for k in order:
                offs[k] = o
                o += ft[k].size
            ent = (key, ft, buf, offs)
            try:
                data.__dict__["_yolat_tree"] = ent
            except AttributeError:
                pass
        _, ft, buf, offs = ent
        R, Ctot, B = ft["R"], ft["Ctot"], ft["B"]
        t = PredictTree()
        t.R, t.Ctot, t.B = R, Ctot, B
        base = buf.data_ptr()
        for k in ("root_row", "root_range", "child_ptr", "child_row", "child_range", "image_root_ptr"):
            setattr(t, k, base + 4 * offs[k])
        K = logits.shape[1]
        out = torch.empty(4 + B + 1 + R + Ctot, dtype=torch.int32, device=dev)
        ws = torch.empty(int(lib.yolat_predict_select_workspace_bytes(N, P, R)) + 16, dtype=torch.uint8, device=dev)
        stream = ops._stream()
        check(lib.yolat_predict_select(logits.data_ptr(), logits.stride(0), P, K, ep if E > 0 else None, se, sc, bp, N, E,
                                       ctypes.byref(t), out.data_ptr(), ws.data_ptr(), ws.numel(), stream),
              "yolat_predict_select")
        plan = self.__dict__.get("_yolat_plan")
        if getattr(plan, "_status", None) is not None:    # the forward's input-validity word rides along in out[2]
            out[2:3].copy_(plan._status)
        host = out.cpu().numpy()                          # the ONE host read of the call (synchronises)
        if int(host[2]) != 0:
            self.check_last_status()                      # raises IndexError / ValueError like the forward's own check
        if int(host[1]) != 0:
            return None
        total = int(host[0])
        slice_image_bbox = [int(v) for v in host[4:4 + B + 1]]
        slice_bbox = torch.from_numpy(host[4 + B + 1:4 + B + 1 + total].astype(np.int64))
        pred_cls = torch.empty(total, K, dtype=torch.float32, device=dev)
        pred_bbox = torch.empty(total, 4, dtype=torch.float32, device=dev)
        bb = bbox if (bbox.dtype == torch.float32 and bbox.is_contiguous()) else bbox.float().contiguous()
        check(lib.yolat_predict_gather(logits.data_ptr(), logits.stride(0), P, K, bb.data_ptr(),
                                       out.data_ptr() + 4 * (4 + B + 1), total, pred_cls.data_ptr(), pred_bbox.data_ptr(),
                                       stream), "yolat_predict_gather")
        pred_cls._yolat_keep = (out, logits, bb)
        return pred_cls, pred_bbox, None, slice_bbox, slice_image_bbox, None

    def _predict_two_pass(self, data, slices):
        """Two-pass root/children inference, arch:139-356: forward on the sub-batch of all root
        proposals; proposals classified as class ``n_classes-1`` get their children evaluated in a
        second forward; per image the root rows are followed by the child rows; boxes are enlarged by
        5 %.  Returns the reference's 6-tuple ``(pred_cls, pred_bbox, None, slice_bbox,
        slice_image_bbox, None)``.  The host walks the proposal tree; node / edge re-indexing and the
        gathers are device kernels (`ops.extract_subgraph`); both forwards run through the HIP path."""
        from .data import select_tree_ranges, interleave_root_child
        if getattr(data, "_yolat_graph", None) is not None and not all(k in data.keys for k in ("edge", "e_attr", "bbox_idx")):
            raise ValueError("predict() cuts sub-graphs out of the raw edge list: batches from collate_to_device(csr=True) "
                             "carry the prepared CSR only and support forward() — collate with csr=False for predict()")
        # the whole batch goes to the device once; both sub-batches are cut out of it by integer kernels
        # (csrc/subgraph.hip) — the host only walks the proposal tree (O(#tree nodes))
        dev = {k: getattr(data, k).cuda(non_blocking=True) for k in ("x", "pos", "edge", "e_attr", "bbox_idx", "bbox",
                                                                    "stat_feats")}
        if dev["x"].dtype != torch.float32:
            dev["x"] = dev["x"].float()
        ps, pe, es, ee, slice_bbox_root, image_root = select_tree_ranges(data, slices)
        sub, status1 = ops.extract_subgraph(data.__class__, dev, ps, pe, es, ee, slice_bbox_root)
        pred_cls, pred_bbox = self.forward(sub, slices)
        is_object = pred_cls.max(1)[1]
        has_object = (is_object == self.n_classes - 1).cpu().numpy()     # D2H sync, as in the reference
        if int(status1.item()) & ops.STATUS_EDGE_RANGE:
            raise KeyError("an edge of a root proposal references a node outside the selected sub-batch")
        self.check_last_status()          # the host has just synchronised: the flags of pass 1 are free to read
        ps, pe, es, ee, slice_bbox_child, image_child = select_tree_ranges(data, slices, has_object)
        if int((pe - ps).sum()) == 0:
            slice_image_bbox, slice_bbox = image_root, slice_bbox_root
        else:
            sub2, status2 = ops.extract_subgraph(data.__class__, dev, ps, pe, es, ee, slice_bbox_child)
            pred_cls2, pred_bbox2 = self.forward(sub2, slices)
            parts, slice_image_bbox = interleave_root_child(image_root, image_child, pred_cls, pred_cls2)
            pred_cls = torch.cat(parts, dim=0)
            parts, _ = interleave_root_child(image_root, image_child, pred_bbox, pred_bbox2)
            pred_bbox = torch.cat(parts, dim=0)
            parts, _ = interleave_root_child(image_root, image_child, torch.tensor(slice_bbox_root),
                                             torch.tensor(slice_bbox_child))
            slice_bbox = torch.cat(parts, dim=0)
            if int(status2.item()) & ops.STATUS_EDGE_RANGE:
                raise KeyError("an edge of a child proposal references a node outside the selected sub-batch")
        w = (pred_bbox[:, 2] - pred_bbox[:, 0]) * 1.05
        h = (pred_bbox[:, 3] - pred_bbox[:, 1]) * 1.05
        cx = (pred_bbox[:, 2] + pred_bbox[:, 0]) / 2
        cy = (pred_bbox[:, 3] + pred_bbox[:, 1]) / 2
        pred_bbox = torch.stack([cx - w / 2, cy - h / 2, cx + w / 2, cy + h / 2], dim=1)
        return pred_cls, pred_bbox, None, slice_bbox, slice_image_bbox, None

    def set_eval_precision(self, precision="fp32"):
        """Storage precision of the eval fast path: "fp32" (default, 1e-4 parity with the reference) or "bf16"
        (bf16 node activations / weights, fp32 accumulation — csrc/bf16_eval.hip; ~1e-2 of the logits' scale).
        Training always runs in fp32."""
        if precision not in ("fp32", "bf16"):
            raise ValueError("precision must be 'fp32' or 'bf16'")
        if self.__dict__.get("_yolat_precision", "fp32") != precision:
            self.__dict__["_yolat_precision"] = precision
            self.__dict__.pop("_yolat_plans", None)
        return self

    def set_train_precision(self, precision="fp32"):
        """Storage precision of the training step's per-edge tensors: "fp32" (default, the parity mode) or "bf16" —
        the [E,64] activations of the edge MLP and their gradients are stored as bfloat16 (fp32 accumulation,
        statistics, parameters, optimizer); used where the factorised edge layer applies (E >= 2 N).  Gradients agree
        with the fp32 step to ~1e-2 (tests/test_gpu_bf16.py)."""
        if precision not in ("fp32", "bf16"):
            raise ValueError("precision must be 'fp32' or 'bf16'")
        self.__dict__["_yolat_train_precision"] = precision
        return self

    def use_hip_graphs(self, on=True):
        """Eval forwards replay a captured hipGraph when they are called again with the same input buffers
        (plan.EvalPlan._run_graph): one graph launch instead of a memset + 12 kernel launches."""
        self.__dict__["_yolat_use_graph"] = bool(on)
        return self

    def forward_scheduled(self, data, slices=None):
        """The Python-scheduled kernel sequence (engine.model_fwd) regardless of mode — the training
        path; in eval mode it must agree with the eval plan."""
        st = self._stage(data)
        pred_cls = _ModelFn.apply(self, st["g"], st["x"], *list(self.parameters()))
        if self.classifier != "softmax":
            pred_cls = torch.sigmoid(pred_cls)
        return pred_cls, st["bbox"]

    def forward_modular(self, data, slices=None):
        """Same result through the module-by-module path (Backbone.forward + scatter + MLPs)."""
        st = self._stage(data)
        x, g, pred_bbox, bbox_idx, e_attr = st["x"], st["g"], st["bbox"], st["bbox_idx"], st["e_attr"]
        out_feat, out_super = self.cls_net(x, [g], [None], [e_attr], bbox_idx)
        out_feat = scatter(out_feat, bbox_idx, dim=0, reduce="max")
        pred_cls = self.prediction_cls(torch.cat([out_feat, out_super], dim=1))
        if self.classifier != "softmax":
            pred_cls = torch.sigmoid(pred_cls)
        return pred_cls, pred_bbox


class _CEFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits, labels):
        loss = torch.empty(1, dtype=torch.float32, device=logits.device)
        dl = torch.empty_like(logits)
        ops.softmax_ce(logits, labels, loss, dl)
        ctx.save_for_backward(dl)
        return loss[0]

    @staticmethod
    def backward(ctx, gout):
        (dl,) = ctx.saved_tensors
        return dl * gout, None


class DetectionLoss(nn.Module):
    """arch:358-379: CrossEntropyLoss (mean over proposals) for classifier='softmax'."""

    def __init__(self, opt):
        super(DetectionLoss, self).__init__()
        self.classifier = opt.classifier
        if opt.classifier != "softmax":
            self.cls_loss = nn.BCELoss()

    def forward(self, out, data):
        pred_cls = out[0]
        labels = data.labels
        if not labels.is_cuda and labels.numel():
            # host-side range check while the labels are still on the host (torch's CrossEntropyLoss raises
            # "Target ... is out of bounds"); device-resident labels are guarded in the kernel (NaN loss)
            lo, hi = int(labels.min()), int(labels.max())
            if lo < 0 or hi >= pred_cls.shape[1]:
                raise IndexError("Target %d is out of bounds." % (lo if lo < 0 else hi))
        gt_cls = labels.cuda(non_blocking=True)
        if self.classifier == "softmax":
            l0 = _CEFn.apply(pred_cls.contiguous(), gt_cls)
        else:
            # non-default branch of the reference (BCE on sigmoid outputs): plain torch ops
            tgt = torch.zeros(pred_cls.size(), device=pred_cls.device).scatter_(1, gt_cls.unsqueeze(1), 1)
            l0 = self.cls_loss(pred_cls, tgt)
        return {"loss": l0, "loss_cls": l0}


class Opt(object):
    """The ``opt`` attributes the model reads (cad_recognition/config.py:26-85, train.py:195-197),
    defaulting to the README recipe (README.md:33-42)."""

    def __init__(self, **kw):
        self.n_filters = 64
        self.act = "relu"
        self.norm = "batch"
        self.bias = True
        self.conv = "attr_edge"
        self.n_classes = 17
        self.classifier = "softmax"
        self.class_specific = False
        self.in_channels = 5
        self.n_blocks = 2
        self.n_blocks_out = 2
        self.dropout = 0.0
        for k, v in kw.items():
            setattr(self, k, v)
