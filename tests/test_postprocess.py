"""CPU tests of the detection post-processing host logic against fixtures produced by the reference's
own ``non_max_suppression`` (train.py:34-121) and utils/det_util.py (tests/golden/make_golden_post.py): the oracle
restatement of torchvision.ops.nms, the candidate selection / class offsets / limits, and the metric code."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.host      # host code: CPU suite, and also the GPU box's -m gpu pass (conftest.py)
import torch

from oracle import oracle_np as onp


@pytest.fixture(scope="module")
def z(golden_dir):
    return np.load(os.path.join(golden_dir, "postprocess.npz"))


def test_oracle_nms_tiny_hand_checked(z):
    keep = onp.nms(z["tiny/boxes"], z["tiny/scores"], 0.5)
    # A kept; B (IoU .75 with A) dropped; C kept; D (.75 with C) dropped; E disjoint kept; F IoU exactly .5: kept (strict >)
    assert keep.tolist() == [0, 2, 4, 5] == z["tiny/keep_0.5"].tolist()
    assert onp.nms(z["tiny/boxes"], z["tiny/scores"], 0.49).tolist() == [0, 2, 4]
    assert onp.nms(z["tiny/boxes"][:0], z["tiny/scores"][:0], 0.5).tolist() == []


@pytest.mark.parametrize("name", ["a", "b", "c", "d"])
def test_non_max_suppression_host_logic_matches_reference(name, z, monkeypatch):
    """postprocess.non_max_suppression with the NMS kernel replaced by the oracle (no GPU here) reproduces the
    reference function's detections exactly: candidate pairs, conf products, class offsets, 300-detection cap,
    class filter and a-priori labels."""
    import yolat_vectorgraphicsrecognition_amd as yv
    from yolat_vectorgraphicsrecognition_amd import postprocess
    monkeypatch.setattr(postprocess.ops, "nms",
                        lambda b, s, t: torch.from_numpy(onp.nms(b.numpy(), s.numpy(), float(t))))
    pred = torch.from_numpy(z["nms_%s/pred" % name].copy())
    if name == "d":
        out = yv.non_max_suppression(pred, conf_thres=0.1, iou_thres=0.5, classes=[2, 5],
                                     labels=[torch.from_numpy(z["nms_d/labels"])])
    else:
        conf, iou, agn = z["nms_%s/args" % name]
        out = yv.non_max_suppression(pred, conf_thres=float(conf), iou_thres=float(iou), agnostic=bool(agn))
    assert len(out) == 1
    np.testing.assert_array_equal(out[0].numpy(), z["nms_%s/out" % name])
    # an image without candidates yields the empty [0, 6] result
    empty = yv.non_max_suppression(torch.zeros(2, 5, 9), conf_thres=0.5)
    assert [tuple(o.shape) for o in empty] == [(0, 6), (0, 6)]


def test_detection_metrics_match_reference(z):
    import yolat_vectorgraphicsrecognition_amd as yv
    det = torch.from_numpy(z["nms_a/out"])
    targets = torch.from_numpy(z["metrics/targets"])
    tps = {}
    for th in (0.5, 0.75):
        m = yv.get_batch_statistics([det], targets, iou_threshold=th)
        np.testing.assert_array_equal(m[0][0], z["metrics/tp_%g" % th])
        tps[th] = m[0][0]
        assert torch.equal(m[0][1], det[:, 4]) and torch.equal(m[0][2], det[:, 5])
    assert 0 < tps[0.75].sum() <= tps[0.5].sum()
    p, r, ap, f1, cls = yv.ap_per_class(tps[0.5], det[:, 4].numpy(), det[:, 5].numpy(), z["metrics/targets"][:, 1])
    for got, key in ((p, "p"), (r, "r"), (ap, "ap"), (f1, "f1")):
        np.testing.assert_allclose(got, z["metrics/" + key], rtol=1e-12, atol=0)
    np.testing.assert_array_equal(cls, z["metrics/cls"])
    iou = yv.bbox_iou(det[:1, :4], targets[:, 2:])
    np.testing.assert_array_equal(iou.numpy(), z["metrics/bbox_iou_row0"])
    # None outputs are skipped, images without targets give all-zero true positives
    m = yv.get_batch_statistics([None, det], targets, iou_threshold=0.5)
    assert len(m) == 1 and m[0][0].sum() == 0
