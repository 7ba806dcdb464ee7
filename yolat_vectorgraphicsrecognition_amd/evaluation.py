"""The reference's evaluation loop (cad_recognition/train.py:324-508, ``test``) on the device path: per batch the
index fix-up, the two-pass ``model.predict``, the loss, top-1 accuracy, and per image
softmax -> [1 - P(last class), P(other classes)] -> class-aware NMS -> true-positive statistics at the IoU
thresholds 0.5 .. 0.95; then AP per class and threshold.  Same inputs (``(data, slices)`` batches of collated items
with ``labels, has_obj, gt_bbox, gt_labels, width, height`` next to the graph tensors) and the same value written to
``opt.test_value`` / returned (the mean AP at the LAST threshold, train.py:507-508); the intermediate numbers the
reference only logs are returned in ``opt.test_report``."""
import numpy as np
import torch

from .data import fixup_offsets
from .postprocess import non_max_suppression, get_batch_statistics, ap_per_class


def evaluate_batch(model, criterion, data, slices, classifier="softmax", iou_thresholds=None, conf_thres=0.0,
                   iou_thres=0.5, fixup=True):
    """One iteration of the loop (train.py:343-460).  Returns a dict: per-threshold ``sample_metrics`` (lists of
    [true_positives, scores, labels] per image), the ground-truth ``labels`` list, ``loss`` dict, ``n_true`` /
    ``n_total`` of the top-1 accuracy, ``y_pred`` / ``y_true``."""
    if iou_thresholds is None:
        iou_thresholds = np.linspace(0.5, 0.95, 10)
    if fixup:
        fixup_offsets(data, slices)                                   # train.py:346-366
    if not hasattr(data, "edge_control"):
        data.edge_control = None
    out = model.predict(data, slices)
    keep = torch.as_tensor([int(v) for v in out[3]], dtype=torch.long)
    data.labels = data.labels[keep]
    if hasattr(data, "has_obj"):
        data.has_obj = data.has_obj[keep]
    slices["bbox"] = out[4]
    loss = criterion(out, data)
    pred_cls, pred_coord = out[0], out[1].clone()
    pred_label = pred_cls.max(1)[1]
    truth = data.labels.to(pred_label.device)
    rep = {"loss": {k: float(v) for k, v in loss.items()}, "n_true": int((pred_label == truth).sum()),
           "n_total": int(pred_label.shape[0]), "y_pred": pred_label.cpu().numpy(), "y_true": data.labels.cpu().numpy(),
           "sample_metrics": [[] for _ in iou_thresholds], "labels": []}
    image_ptr, label_ptr = slices["bbox"], slices["gt_labels"]
    for i in range(len(image_ptr) - 1):
        pc = pred_cls[image_ptr[i]:image_ptr[i + 1]]
        pb = pred_coord[image_ptr[i]:image_ptr[i + 1]]
        w, h = float(data.width[i]), float(data.height[i])
        scale = torch.tensor([w, h, w, h], dtype=pb.dtype, device=pb.device)
        gt = data.gt_bbox[int(label_ptr[i]):int(label_ptr[i + 1])].float() * scale.cpu()
        gl = data.gt_labels[int(label_ptr[i]):int(label_ptr[i + 1])]
        targets = torch.cat((torch.zeros(gl.shape[0], 1), gl.float().unsqueeze(1), gt), 1)
        rep["labels"] += gl.tolist()
        if classifier == "softmax":
            pc = torch.softmax(pc, dim=1)
        conf = torch.cat((1 - pc[:, -1:], pc[:, :-1]), 1)
        pred = torch.cat((pb * scale, conf), 1).unsqueeze(0)
        outputs = [o.cpu() for o in non_max_suppression(pred, conf_thres=conf_thres, iou_thres=iou_thres)]
        for t, th in enumerate(iou_thresholds):
            rep["sample_metrics"][t] += get_batch_statistics(outputs, targets, iou_threshold=th)
    return rep


def test(model, test_loader, criterion, opt):
    """train.py:324-508.  Returns the mean AP at the last IoU threshold (None when nothing was detected), like the
    reference; the per-threshold mAPs, MAP@ALL, top-1 accuracy and mean losses are left in ``opt.test_report``."""
    model.eval()
    steps = int(getattr(opt, "map_step", 10))
    ths = np.linspace(0.5, 0.95, steps)
    metrics, labels, losses = [[] for _ in range(steps)], [], {}
    n_true = n_total = 0
    with torch.no_grad():
        for data, slices in test_loader:
            rep = evaluate_batch(model, criterion, data, slices, classifier=getattr(opt, "classifier", "softmax"),
                                 iou_thresholds=ths)
            for t in range(steps):
                metrics[t] += rep["sample_metrics"][t]
            labels += rep["labels"]
            n_true, n_total = n_true + rep["n_true"], n_total + rep["n_total"]
            for k, v in rep["loss"].items():
                losses.setdefault(k, []).append(v)
    report = {"iou_thresholds": ths, "map": [], "top1": n_true / max(n_total, 1),
              "loss": {k: float(np.mean(v)) for k, v in losses.items()}}
    ap = None
    for t in range(steps):
        if len(metrics[t]) == 0:
            return None
        tp, scores, plabels = [np.concatenate([np.asarray(a) for a in col], 0) for col in zip(*metrics[t])]
        _, _, ap, _, _ = ap_per_class(tp, scores, plabels, np.asarray(labels))
        report["map"].append(float(np.mean(ap)) if len(ap) else 0.0)
    report["map_all"] = float(np.sum(report["map"]) / 10)             # the reference divides by 10 (train.py:484)
    opt.test_report = report
    opt.test_value = float(np.mean(ap)) if ap is not None and len(ap) else 0.0
    return opt.test_value
