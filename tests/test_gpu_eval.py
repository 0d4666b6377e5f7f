"""GPU suite: evaluation post-processing on the device (/root/reference/test.py:88-206) -- score filter +
compaction, greedy rotated-box suppression (separating axes / 3-D IoU), bird's-eye-IoU matching -- through the Test harness,
against tests/golden/eval.npz (survivor indices and counters produced by the imported reference) and against the host
statement (evalgeom.py) on larger random box sets."""
import numpy as np
import pytest
import torch

from _util import golden_cfg, load_golden, pkg

pytestmark = pytest.mark.gpu


def _harness():
    T = pkg("test").Test.__new__(pkg("test").Test)
    T.initialize_ap()
    return T


def _idx(kept, boxes):
    b = boxes.cpu().numpy()
    return np.array([int(np.where((b == k.cpu().numpy()).all(1))[0][0]) for k in kept], dtype=np.int64)


def test_nms_and_counters_match_reference_golden_on_device():
    z = load_golden("eval.npz")
    T = _harness()
    pred = [torch.from_numpy(z["pred"][b]).cuda() for b in range(2)]
    assert T._on_device(pred)
    ki, ks = T.NMS_IOU(pred, 0.01), T.NMS_SAT(pred)
    for b in range(2):
        assert np.array_equal(_idx(ki[b], pred[b]), z["keep_iou_%d" % b]), "IoU survivors of sample %d" % b
        assert np.array_equal(_idx(ks[b], pred[b]), z["keep_sat_%d" % b]), "SAT survivors of sample %d" % b
        assert all(k.is_cuda for k in ks[b])
    T.precision_recall_singleshot(ks, torch.from_numpy(z["ref"]))
    assert T.get_num_T() == int(z["num_T"]) and T.get_num_P() == int(z["num_P"])
    assert [T.get_num_TP_set()[t] for t in T.IOU_threshold] == z["num_TP"].tolist()
    # edge cases: no predictions, no labels
    assert T.NMS_SAT([torch.zeros(0, 7, device="cuda")]) == [[]] and T.NMS_IOU([torch.zeros(0, 7, device="cuda")]) == [[]]


@pytest.mark.parametrize("n", [65, 700, 2000])
def test_nms_device_equals_host_statement(n):
    """Larger sets (several 64-box words of the survivor bit set, long suppression chains): device == numpy statement."""
    det = pkg("detfill")
    u = det.uniform((n, 7), 4000 + n, 0.0, 1.0)
    span = 6.0 * np.sqrt(n)                          # density chosen so that roughly a third of the boxes survive
    boxes = np.stack([u[:, 0] * span, u[:, 1] * span, -1.0 + 0.2 * u[:, 2], 3.5 + 1.3 * u[:, 3], 1.6 + 0.5 * u[:, 4], 1.4 + 0.4 * u[:, 5],
                      3.14159 * u[:, 6]], 1).astype(np.float32)
    T = _harness()
    host = torch.from_numpy(boxes)
    dev = host.cuda()
    ks_h, ks_d = T.NMS_SAT([host])[0], T.NMS_SAT([dev])[0]
    assert 0.1 * n < len(ks_h) < 0.95 * n
    assert np.array_equal(_idx(ks_d, dev), _idx(ks_h, host))
    if n <= 700:                                     # the host IoU statement is O(n^2) polygon clips in Python
        # IoU flavour on boxes laid out in the vertical (x, z) footprint plane the reference's 3-D IoU uses
        b2 = boxes.copy()
        b2[:, 2] = u[:, 1] * span * 0.25
        b2[:, 1] = -1.0 + 0.2 * u[:, 2]
        host2 = torch.from_numpy(b2)
        ki_h, ki_d = T.NMS_IOU([host2], 0.01)[0], T.NMS_IOU([host2.cuda()], 0.01)[0]
        assert len(ki_h) < n
        assert np.array_equal(_idx(ki_d, host2), _idx(ki_h, host2))


def test_score_filter_and_eval_step_on_device():
    """get_bboxes_device == get_bboxes (order included), and a whole Test.get_eval_value_onestep on the device equals the same
    step with the post-processing on the host."""
    from test_gpu_model import build, tiny_input
    z = load_golden("model_tiny.npz")
    lz = load_golden("loss.npz")
    net, cfg = build(golden_cfg(z), "f32")
    cfg["score_threshold"] = 0.5
    Tm = pkg("test")
    x = tiny_input().cuda()
    img = torch.zeros(x.shape[0], 3, 8, 8, dtype=torch.uint8, device="cuda")
    boxes, nb = torch.from_numpy(lz["bboxes"]), torch.from_numpy(lz["nbox"])
    T = Tm.Test(net, cfg)
    with torch.no_grad():
        pred = net(x, img)
    a = T.get_bboxes_device(pred)
    cls, _, bb = torch.split(pred, [4, 14, 14], dim=1)
    b = T.get_bboxes(cls, bb)
    assert len(a) == len(b) and all(torch.equal(p, q) for p, q in zip(a, b)) and sum(p.shape[0] for p in a) > 0
    np.random.seed(3)
    loss_d, sel_d = T.get_eval_value_onestep(x, img, boxes, nb)
    dev = (T.get_num_T(), T.get_num_P(), dict(T.get_num_TP_set()), [len(k) for k in T.refined_bbox])
    T2 = Tm.Test(net, cfg)
    T2._on_device = lambda pb: False                 # force the numpy statement
    np.random.seed(3)
    loss_h, sel_h = T2.get_eval_value_onestep(x, img, boxes, nb)
    host = (T2.get_num_T(), T2.get_num_P(), dict(T2.get_num_TP_set()), [len(k) for k in T2.refined_bbox])
    assert dev == host and abs(loss_d - loss_h) < 1e-6
