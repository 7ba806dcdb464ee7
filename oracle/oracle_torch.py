"""CPU ORACLE — TEST INFRASTRUCTURE ONLY.  Never imported by the product path.

Op-for-op, *unfused* pure-torch (fp32, CPU) restatement of the YOLaT hot path
(``SparseCADGCN.forward`` / train step).  Only ``tests/``, ``__graft_entry__.smoke()``
and ``bench.py``'s ``cpu_baseline`` leg may import this file.

Parity status: **wiring pinned, third-party semantics restated**.
 * The reference publishes no tests / golden vectors for this path, so
   parity is not pinned by reference-owned vectors.
 * The wiring (which tensor goes where) is pinned against the reference's *own*
   module code executed in the build container by ``tests/golden/make_golden.py``
   (it imports /root/reference with the absent third-party packages replaced by the
   restatements in this file) — see the ``ref_*`` entries of the golden fixtures.
 * ``torch_geometric.nn.conv.MessagePassing.propagate`` (PyG, unpinned, ~1.6/1.7,
   deepgcn_env_install.sh:31) and ``torch_scatter.scatter`` (wheel index
   torch-1.7.0+cu102, unpinned, deepgcn_env_install.sh:27) are absent from
   /root/reference and from the image; their published algorithms are restated in
   ``scatter`` / ``propagate_mean`` below and cross-checked by the
   independent naive-loop oracle in ``oracle_np.py``.

Every symbol cites the reference file:line it follows (paths relative to
/root/reference).  Module attribute names reproduce the reference's state_dict keys
 so a reference checkpoint loads into these classes.
"""
import torch
from torch import nn

# --------------------------------------------------------------------------------------
# third-party semantics (not in /root/reference) — SURVEY.md App. B
# --------------------------------------------------------------------------------------


class _ScatterMax(torch.autograd.Function):
    """torch_scatter.scatter(..., reduce='max') along dim 0.

    CPU semantics of torch_scatter: running max initialised to lowest(), strict ``>``
    update (first occurrence wins), rows that received nothing are filled with 0;
    backward routes the gradient to the arg-max row only.
    Call sites: cad_recognition/architecture3cc_rpn_gp_iter2.py:122.
    """

    @staticmethod
    def forward(ctx, src, index, dim_size):
        n, d = src.shape
        idx = index.view(-1, 1).expand(n, d)
        out = torch.full((dim_size, d), float("-inf"), dtype=src.dtype)
        out = out.scatter_reduce(0, idx, src, "amax", include_self=True)
        # first occurrence of the max inside each segment
        is_max = src == out.index_select(0, index)
        rows = torch.arange(n).view(-1, 1).expand(n, d)
        cand = torch.where(is_max, rows, torch.full_like(rows, n))
        arg = torch.full((dim_size, d), n, dtype=torch.long)
        arg = arg.scatter_reduce(0, idx, cand, "amin", include_self=True)
        empty = arg == n
        out = out.masked_fill(empty, 0.0)
        ctx.save_for_backward(arg)
        ctx.n = n
        ctx.mark_non_differentiable(arg)
        return out, arg

    @staticmethod
    def backward(ctx, gout, _garg):
        (arg,) = ctx.saved_tensors
        g = torch.zeros(ctx.n + 1, gout.shape[1], dtype=gout.dtype)
        g.scatter_(0, arg, gout)
        return g[: ctx.n], None, None


def scatter(src, index, dim=0, dim_size=None, reduce="sum"):
    """torch_scatter.scatter restated (dim 0 only; that is all the path uses).

    Output rows = ``dim_size`` or ``index.max()+1`` (the reference passes no dim_size at
    architecture3cc_rpn_gp_iter2.py:67,122).  'sum': zeros + scatter_add in input order.
    'mean': sum / count.clamp(min=1).  'max': see _ScatterMax.
    """
    assert dim == 0 and src.dim() == 2
    if dim_size is None:
        dim_size = int(index.max()) + 1 if index.numel() > 0 else 0
    if reduce in ("sum", "add"):
        out = torch.zeros(dim_size, src.shape[1], dtype=src.dtype)
        return out.index_add(0, index, src)
    if reduce == "mean":
        out = torch.zeros(dim_size, src.shape[1], dtype=src.dtype)
        out = out.index_add(0, index, src)
        cnt = torch.zeros(dim_size, dtype=src.dtype).index_add(
            0, index, torch.ones(index.shape[0], dtype=src.dtype))
        return out / cnt.clamp(min=1).view(-1, 1)
    if reduce == "max":
        return _ScatterMax.apply(src, index, dim_size)[0]
    raise ValueError(reduce)


def propagate_mean(message_fn, edge_index, x, num_nodes, **kwargs):
    """PyG ``MessagePassing(aggr='mean', flow='source_to_target').propagate`` restated.

    x_j = x[edge_index[0]] (source), x_i = x[edge_index[1]] (target); extra kwargs go to
    ``message`` untouched; aggregate = scatter-mean over edge_index[1] with
    dim_size = N; update = identity.  Call site: gcn_lib/sparse/torch_vertex.py:324.
    """
    x_j = x.index_select(0, edge_index[0])
    x_i = x.index_select(0, edge_index[1])
    m = message_fn(x_i=x_i, x_j=x_j, **kwargs)
    return scatter(m, edge_index[1], dim=0, dim_size=num_nodes, reduce="mean")


# --------------------------------------------------------------------------------------
# gcn_lib/sparse/torch_nn.py
# --------------------------------------------------------------------------------------


def act_layer(act_type):
    """gcn_lib/sparse/torch_nn.py:9-20 (only 'relu' is reachable from the arch)."""
    act = act_type.lower()
    if act == "relu":
        return nn.ReLU(False)
    if act == "leakyrelu":
        return nn.LeakyReLU(0.2, False)
    if act == "prelu":
        return nn.PReLU(num_parameters=1, init=0.2)
    raise NotImplementedError("activation layer [%s] is not found" % act)


def norm_layer(norm_type, nc):
    """gcn_lib/sparse/torch_nn.py:23-34 (only 'batch' is reachable from the arch)."""
    norm = norm_type.lower()
    if norm == "batch":
        return nn.BatchNorm1d(nc, affine=True)
    if norm == "layer":
        return nn.LayerNorm(nc, elementwise_affine=True)
    if norm == "instance":
        return nn.InstanceNorm1d(nc, affine=False)
    raise NotImplementedError("normalization layer [%s] is not found" % norm)


class MultiSeq(nn.Sequential):
    """gcn_lib/sparse/torch_nn.py:37-47 — Sequential that splats tuple outputs."""

    def forward(self, *inputs):
        for module in self._modules.values():
            if type(inputs) == tuple:
                inputs = module(*inputs)
            else:
                inputs = module(inputs)
        return inputs


class MLP(nn.Sequential):
    """gcn_lib/sparse/torch_nn.py:50-71 — [Linear, (norm), (act), (Dropout2d)] per layer."""

    def __init__(self, channels, act="relu", norm=None, bias=True, drop=0.0, last_lin=False):
        layers = []
        for i in range(1, len(channels)):
            layers.append(nn.Linear(channels[i - 1], channels[i], bias))
            if i == len(channels) - 1 and last_lin:
                continue
            if norm is not None and norm.lower() != "none":
                layers.append(norm_layer(norm, channels[i]))
            if act is not None and act.lower() != "none":
                layers.append(act_layer(act))
            if drop > 0:
                layers.append(nn.Dropout2d(drop))
        super().__init__(*layers)


# --------------------------------------------------------------------------------------
# gcn_lib/sparse/torch_vertex.py
# --------------------------------------------------------------------------------------


class AttrRelativeEdgeConvGlobalPool2(nn.Module):
    """gcn_lib/sparse/torch_vertex.py:288-341.

    out_i = mean_{e:(j->i)} nn([x_i, x_j - x_i, a_e]) + lin_r(x_i);  x_node' = mlp_node(x_node).
    """

    def __init__(self, in_channels, out_channels):
        super().__init__()
        self.nn = MLP([in_channels * 2 + 4, out_channels, out_channels], "relu", "batch")  # :309
        self.lin_r = nn.Linear(in_channels, out_channels, bias=True)                       # :310
        self.mlp_node = MLP([in_channels, out_channels], "relu", "batch")                  # :311
        self.in_channels = in_channels

    def message(self, x_i, x_j, norm, attr):                                              # :330-337
        f = torch.cat([x_i, x_j - x_i, attr], dim=1)
        if norm is None:
            return self.nn(f)
        return norm.view(-1, 1) * self.nn(f)

    def forward(self, x, x_node, edge_index, edge_weight=None, edge_attr=None):           # :319-328
        out = propagate_mean(self.message, edge_index, x, x.shape[0],
                             norm=edge_weight, attr=edge_attr)
        out = out + self.lin_r(x)
        x_node = self.mlp_node(x_node)
        return out, x_node


class GraphConv(nn.Module):
    """gcn_lib/sparse/torch_vertex.py:730-775 — only the 'attr_edge_gp2' branch is on the path."""

    def __init__(self, in_channels, out_channels, conv="gcn", act="relu", norm=None, bias=True, heads=8):
        super().__init__()
        self.conv = conv.lower()
        if self.conv == "attr_edge_gp2":
            self.gconv = AttrRelativeEdgeConvGlobalPool2(in_channels, out_channels)
        else:
            raise NotImplementedError("conv {} is not implemented".format(conv))

    def forward(self, x, edge_index, edge_weight=None, edge_attr=None, pos=None, x_node=None):
        return self.gconv(x, x_node, edge_index, edge_weight, edge_attr)                  # :772-773


class ResBlock(nn.Module):
    """gcn_lib/sparse/torch_vertex.py:808-829 — gp2 branch: NO residual (:825-826 commented out)."""

    def __init__(self, channels, conv="edge", act="relu", norm=None, bias=True, res_scale=1):
        super().__init__()
        self.body = GraphConv(channels, channels, conv, act, norm, bias)
        self.res_scale = res_scale
        self.channels = channels

    def forward(self, x, edge, edge_weight=None, edge_attr=None, pos=None, x_node=None):
        return self.body(x, edge, edge_weight, edge_attr, x_node=x_node)


# --------------------------------------------------------------------------------------
# cad_recognition/architecture3cc_rpn_gp_iter2.py
# --------------------------------------------------------------------------------------


class Backbone(nn.Module):
    """cad_recognition/architecture3cc_rpn_gp_iter2.py:15-71."""

    def __init__(self, opt):
        super().__init__()
        channels = opt.n_filters
        act, norm, bias = opt.act, opt.norm, opt.bias
        conv = "attr_edge_gp2"                                                             # :22 (hard-coded)
        self.n_edges = 1
        self.n_blocks = opt.n_blocks
        self.n_blocks_out = opt.n_blocks_out
        self.heads = nn.ModuleList()
        self.n_classes = opt.n_classes
        self.class_specific = opt.class_specific
        self.head = GraphConv(opt.in_channels, channels, conv, act, norm, bias)            # :34
        self.backbone = MultiSeq(*[ResBlock(channels, conv, act, norm, bias)
                                   for _ in range(self.n_blocks - 1)])                    # :36
        fusion_dims = int(channels + channels * (self.n_blocks_out - 1))
        self.fusion_block = MLP([fusion_dims, 1024], act, norm, bias)                      # :40
        self.fusion_block_super = MLP([fusion_dims, 1024], act, norm, bias)                # :41
        self.fusion_dims = fusion_dims

    def forward(self, x, edges, edge_weights, edge_attrs, bbox_idx):                       # :44-71
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
        feats_super = scatter(feats_super, bbox_idx, dim=0, reduce="mean")                 # :67
        out_feat_super = torch.cat((self.fusion_block_super(feats_super), feats_super), dim=1)
        return out_feat, out_feat_super


def predict_gather(data, slices, has_object):                                                            # :153-164, :277-296
    roots, slice_root = data.roots, slices["roots"]
    slice_pos, slice_edge, slice_bbox, image_off = [], [], [], [0]
    count = 0
    for i in range(0, len(slice_root) - 1):
        for root in roots[int(slice_root[i]):int(slice_root[i + 1])]:
            if has_object is None:
                nodes = [root]
            else:
                nodes = root.children if bool(has_object[count]) else []
                count += 1
            for nd in nodes:
                slice_pos += list(range(nd.value["idx_pos"][0] + int(slices["pos"][i]),
                                        nd.value["idx_pos"][1] + int(slices["pos"][i])))
                slice_edge += list(range(nd.value["idx_edge"][0] + int(slices["edge"][i]),
                                         nd.value["idx_edge"][1] + int(slices["edge"][i])))
                _ = slices["edge_super"][i]
                slice_bbox.append(int(nd.value["idx_bbox"] + int(slices["bbox"][i])))
        image_off.append(len(slice_bbox))
    return slice_pos, slice_edge, slice_bbox, image_off

def predict_build_data(data, slice_pos, slice_edge, slice_bbox):                                 # :166-242
    o2n = {}
    for new_i, old_i in enumerate(slice_pos):
        o2n[old_i] = new_i
    nd = data.__class__(x=data.x[slice_pos], pos=data.pos[slice_pos])
    old_idx = data.bbox_idx[slice_pos]
    edge = [[o2n[int(e[0])], o2n[int(e[1])]] for e in data.edge[slice_edge].numpy()]
    nd.edge = torch.tensor(edge, dtype=torch.long).reshape(-1, 2)
    nd.e_attr = data.e_attr[slice_edge]
    nd.bbox = data.bbox[slice_bbox]
    nd.stat_feats = data.stat_feats[slice_bbox]
    new_idx, count = [0], 0
    for i in range(1, old_idx.size(0)):
        if old_idx[i] != old_idx[i - 1]:
            count += 1
        new_idx.append(count)
    nd.bbox_idx = torch.tensor(new_idx, dtype=torch.long)
    return nd


class SparseCADGCN(nn.Module):
    """cad_recognition/architecture3cc_rpn_gp_iter2.py:73-137 (forward only; CPU, no .cuda())."""

    def __init__(self, opt):
        super().__init__()
        act, norm, bias = opt.act, opt.norm, opt.bias
        self.n_classes = opt.n_classes
        self.classifier = opt.classifier
        self.class_specific = opt.class_specific
        self.dim_stat = 0                                                                  # :87
        self.cls_net = Backbone(opt)
        d = (self.cls_net.fusion_dims + 1024) * 2 + self.dim_stat
        self.prediction_cls = MultiSeq(MLP([d, 512], act, norm, bias),                     # :91-93
                                       MLP([512, 256], act, norm, bias, drop=opt.dropout),
                                       MLP([256, opt.n_classes], None, None, bias))
        self.model_init()

    def model_init(self):                                                                  # :97-104
        for m in self.modules():
            if isinstance(m, nn.Linear):
                nn.init.kaiming_normal_(m.weight)
                if m.bias is not None:
                    m.bias.data.zero_()

    def forward(self, data, slices=None):                                                  # :106-137
        x = data.x
        bbox_idx = data.bbox_idx
        edges = [data.edge.T]
        pred_bbox = data.bbox
        out_feat, out_super = self.cls_net(x, edges, [None], [data.e_attr], bbox_idx)
        out_feat = scatter(out_feat, bbox_idx, dim=0, reduce="max")                        # :122
        pred_cls = self.prediction_cls(torch.cat([out_feat, out_super], dim=1))
        if self.classifier != "softmax":
            pred_cls = torch.sigmoid(pred_cls)
        return pred_cls, pred_bbox


    def predict(self, data, slices):                                                       # :139-356
        """Two-pass root/children inference, restated loop for loop (the element-wise Python loops of
        the reference are kept on purpose: this is the checker for the product's vectorised slicing)."""
        sp, se, sb_root, img_root = predict_gather(data, slices, None)
        pred_cls, pred_bbox = self.forward(predict_build_data(data, sp, se, sb_root), slices)
        _, is_object = pred_cls.max(1)
        has_object = is_object == self.n_classes - 1
        sp, se, sb_child, img_child = predict_gather(data, slices, has_object)
        if len(sp) == 0:
            slice_image_bbox, slice_bbox = img_root, sb_root
        else:
            pred_cls2, pred_bbox2 = self.forward(predict_build_data(data, sp, se, sb_child), slices)

            def interleaf(out_p, out_c):                                                   # :317-328
                out, s = [], [0]
                for i in range(len(img_child) - 1):
                    out.append(out_p[img_root[i]:img_root[i + 1]])
                    out.append(out_c[img_child[i]:img_child[i + 1]])
                    s.append(s[-1] + img_root[i + 1] - img_root[i] + img_child[i + 1] - img_child[i])
                return torch.cat(out, dim=0), s
            pred_cls, slice_image_bbox = interleaf(pred_cls, pred_cls2)
            pred_bbox, _ = interleaf(pred_bbox, pred_bbox2)
            slice_bbox, _ = interleaf(torch.tensor(sb_root), torch.tensor(sb_child))
        w = (pred_bbox[:, 2] - pred_bbox[:, 0]) * 1.05                                     # :338-351
        h = (pred_bbox[:, 3] - pred_bbox[:, 1]) * 1.05
        cx = (pred_bbox[:, 2] + pred_bbox[:, 0]) / 2
        cy = (pred_bbox[:, 3] + pred_bbox[:, 1]) / 2
        pred_bbox = torch.cat([(cx - w / 2).unsqueeze(1), (cy - h / 2).unsqueeze(1),
                               (cx + w / 2).unsqueeze(1), (cy + h / 2).unsqueeze(1)], dim=1)
        return pred_cls, pred_bbox, None, slice_bbox, slice_image_bbox, None


class DetectionLoss(nn.Module):
    """cad_recognition/architecture3cc_rpn_gp_iter2.py:358-379."""

    def __init__(self, opt):
        super().__init__()
        self.cls_loss = nn.CrossEntropyLoss() if opt.classifier == "softmax" else nn.BCELoss()
        self.classifier = opt.classifier

    def forward(self, out, data):
        pred_cls = out[0]
        gt = data.labels
        if self.classifier != "softmax":
            gt = torch.zeros(pred_cls.size()).scatter_(1, gt.unsqueeze(1), 1)
        l0 = self.cls_loss(pred_cls, gt)
        return {"loss": l0, "loss_cls": l0}


class Opt:
    """The attributes of ``OptInit().get_args()`` the model reads (cad_recognition/config.py:26-85,
    train.py:195-197), with the README training recipe as defaults (README.md:33-42)."""

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


def train_step(model, criterion, optimizer, data):
    """cad_recognition/train.py:263-284: zero_grad -> forward -> loss -> backward -> step."""
    optimizer.zero_grad()
    out = model(data, None)
    loss = criterion(out, data)["loss"]
    loss.backward()
    optimizer.step()
    return loss.detach(), out[0].detach()
