"""Detection post-processing of the reference's evaluation loop: the class-aware
``non_max_suppression`` of cad_recognition/train.py:34-121 (same function in detect.py:47-134) with
``torchvision.ops.nms`` replaced by the HIP kernel (ops.nms / csrc/nms.hip), and the detection metrics of
utils/det_util.py:71-202 (``get_batch_statistics``, ``ap_per_class``, ``compute_ap``, ``bbox_iou``).
Same names, arguments, defaults and return layouts as the reference, so train.test / detect.py call them unchanged.
"""
import numpy as np
import torch

from . import ops

MAX_WH = 4096          # class offset of the batched NMS (train.py:44)
MAX_DET = 300          # detections kept per image (train.py:45)
MAX_NMS = 30000        # boxes handed to nms (train.py:47)


def _candidates(rows, nc, conf_thres, multi_label):
    """rows [m, 5 + nc] of one image (already past the objectness threshold) -> [k, 6] = (box, conf, cls) candidates:
    conf = obj_conf * cls_conf; several classes: one candidate per (row, class) pair above the threshold, row-major
    order (train.py:76-88); one class: the best class of every row."""
    scores = rows[:, 5:] * rows[:, 4:5]
    if multi_label:
        r, c = (scores > conf_thres).nonzero(as_tuple=True)
        return torch.cat((rows[r, :4], scores[r, c].unsqueeze(1), c.unsqueeze(1).to(rows.dtype)), 1)
    conf, c = scores.max(1, keepdim=True)
    return torch.cat((rows[:, :4], conf, c.to(rows.dtype)), 1)[conf.view(-1) > conf_thres]


def non_max_suppression(prediction, conf_thres=0.25, iou_thres=0.45, classes=None, agnostic=False, labels=()):
    """prediction [B, n, 5 + nc] = (x1, y1, x2, y2, obj_conf, cls_conf...) -> list of B tensors [k, 6] =
    (x1, y1, x2, y2, conf, cls) sorted by descending conf — the contract of the reference's function
    (cad_recognition/train.py:34-121, detect.py:47-134), fixtures tests/golden/postprocess.npz.  The greedy suppression
    itself is ONE device kernel per image (ops.nms, csrc/nms.hip): boxes of class c are shifted by c * 4096 so that the
    single call is class-aware, the strongest 30 000 candidates enter it and at most 300 detections leave it.
    (The reference's 10-second wall-clock guard is not reproduced: it protects a host-side NMS; the kernel takes
    6 ms at the 30 000-box cap.)"""
    nc = prediction.shape[2] - 5
    dev = prediction.device
    keep_rows = prediction[..., 4] > conf_thres
    wanted = None if classes is None else torch.as_tensor(classes, device=dev, dtype=prediction.dtype)
    out = []
    for xi in range(prediction.shape[0]):
        rows = prediction[xi][keep_rows[xi]]
        if labels and len(labels[xi]):
            # a-priori labels (train.py:62-69): every label row becomes a certain detection of its class
            lb = labels[xi]
            extra = torch.zeros((len(lb), nc + 5), device=dev, dtype=rows.dtype)
            extra[:, :4], extra[:, 4] = lb[:, 1:5], 1.0
            extra[torch.arange(len(lb)), lb[:, 0].long() + 5] = 1.0
            rows = torch.cat((rows, extra), 0)
        det = _candidates(rows, nc, conf_thres, nc > 1) if rows.shape[0] else rows.new_zeros((0, 6))
        if wanted is not None and det.shape[0]:
            det = det[(det[:, 5:6] == wanted).any(1)]
        if det.shape[0] > MAX_NMS:
            det = det[det[:, 4].argsort(descending=True)[:MAX_NMS]]
        if det.shape[0]:
            shift = det[:, 5:6] * (0 if agnostic else MAX_WH)
            det = det[ops.nms(det[:, :4] + shift, det[:, 4], iou_thres)[:MAX_DET]]
        out.append(det if det.shape[0] else torch.zeros((0, 6), device=dev))
    return out


# ---------------------------------------------------------------------------------------------
# utils/det_util.py:71-202 — host-side metric code (numpy), same conventions (+1 pixel box sizes in bbox_iou)
# ---------------------------------------------------------------------------------------------
def bbox_iou(box1, box2, x1y1x2y2=True):
    """IoU of box1 [1 or m, 4] against box2 [m, 4] with the reference's +1-pixel convention (det_util.py:213-240)."""
    box1, box2 = torch.as_tensor(box1), torch.as_tensor(box2)
    if not x1y1x2y2:
        b1x1, b1x2 = box1[:, 0] - box1[:, 2] / 2, box1[:, 0] + box1[:, 2] / 2
        b1y1, b1y2 = box1[:, 1] - box1[:, 3] / 2, box1[:, 1] + box1[:, 3] / 2
        b2x1, b2x2 = box2[:, 0] - box2[:, 2] / 2, box2[:, 0] + box2[:, 2] / 2
        b2y1, b2y2 = box2[:, 1] - box2[:, 3] / 2, box2[:, 1] + box2[:, 3] / 2
    else:
        b1x1, b1y1, b1x2, b1y2 = box1[:, 0], box1[:, 1], box1[:, 2], box1[:, 3]
        b2x1, b2y1, b2x2, b2y2 = box2[:, 0], box2[:, 1], box2[:, 2], box2[:, 3]
    iw = torch.clamp(torch.min(b1x2, b2x2) - torch.max(b1x1, b2x1) + 1, min=0)
    ih = torch.clamp(torch.min(b1y2, b2y2) - torch.max(b1y1, b2y1) + 1, min=0)
    inter = iw * ih
    a1 = (b1x2 - b1x1 + 1) * (b1y2 - b1y1 + 1)
    a2 = (b2x2 - b2x1 + 1) * (b2y2 - b2y1 + 1)
    return inter / (a1 + a2 - inter + 1e-16)


def get_batch_statistics(outputs, targets, iou_threshold):
    """Per image [true_positives (np.float64 [n]), pred_scores, pred_labels] (det_util.py:148-199): predictions are
    visited in order; one is a true positive when its best same-label target (IoU >= threshold, first on ties) has
    not been claimed yet; the walk stops once every target of the image is claimed."""
    batch_metrics = []
    for sample_i in range(len(outputs)):
        output = outputs[sample_i]
        if output is None:
            continue
        pred_boxes, pred_scores, pred_labels = output[:, :4], output[:, 4], output[:, -1]
        true_positives = np.zeros(pred_boxes.shape[0])
        annotations = targets[targets[:, 0] == sample_i][:, 1:]
        if len(annotations):
            target_labels, target_boxes = annotations[:, 0], annotations[:, 1:]
            claimed = set()
            for pred_i in range(pred_boxes.shape[0]):
                if len(claimed) == len(annotations):
                    break
                same = target_labels == pred_labels[pred_i]
                if not bool(same.any()):
                    continue
                iou = bbox_iou(pred_boxes[pred_i].unsqueeze(0), target_boxes)
                matched = torch.where(same & (iou >= iou_threshold), iou, torch.zeros_like(iou))
                iou_max, box_index = matched.max(0)
                if iou_max >= iou_threshold and int(box_index) not in claimed:
                    true_positives[pred_i] = 1
                    claimed.add(int(box_index))
        batch_metrics.append([true_positives, pred_scores, pred_labels])
    return batch_metrics


def compute_ap(recall, precision):
    """Area under the precision envelope at the recall change points (det_util.py:124-145)."""
    mrec = np.concatenate(([0.0], recall, [1.0]))
    mpre = np.concatenate(([0.0], precision, [0.0]))
    mpre = np.maximum.accumulate(mpre[::-1])[::-1]
    i = np.where(mrec[1:] != mrec[:-1])[0]
    return np.sum((mrec[i + 1] - mrec[i]) * mpre[i + 1])


def ap_per_class(tp, conf, pred_cls, target_cls):
    """(precision, recall, AP, f1, classes) per class present in the targets (det_util.py:71-121)."""
    order = np.argsort(-conf)
    tp, conf, pred_cls = tp[order], conf[order], pred_cls[order]
    unique_classes = np.unique(target_cls)
    ap, p, r = [], [], []
    for c in unique_classes:
        sel = pred_cls == c
        n_gt, n_p = (target_cls == c).sum(), sel.sum()
        if n_p == 0 and n_gt == 0:
            continue
        if n_p == 0 or n_gt == 0:
            ap.append(0); r.append(0); p.append(0)
            continue
        fpc, tpc = (1 - tp[sel]).cumsum(), tp[sel].cumsum()
        recall_curve = tpc / (n_gt + 1e-16)
        precision_curve = tpc / (tpc + fpc)
        r.append(recall_curve[-1]); p.append(precision_curve[-1])
        ap.append(compute_ap(recall_curve, precision_curve))
    p, r, ap = np.array(p), np.array(r), np.array(ap)
    f1 = 2 * p * r / (p + r + 1e-16)
    return p, r, ap, f1, unique_classes.astype("int32")
