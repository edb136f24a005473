"""Hand-checkable cases anchoring the oracle's NMS restatement (torchvision.ops.nms is
absent and un-pinned in the reference: PARITY UNPINNED)."""
import numpy as np

from oracle import ref_cpu


def test_iou_exactly_at_threshold_is_kept():
    # boxes (0,0,10,10) and (0,0,10,5): inter 50, union 100 -> IoU 0.5, NOT > 0.5 -> both kept
    b = np.array([[0, 0, 10, 10], [0, 0, 10, 5]], dtype=np.float32)
    s = np.array([0.9, 0.8], dtype=np.float32)
    assert ref_cpu.nms(b, s, 0.5).tolist() == [0, 1]
    b[1, 3] = 5.5        # IoU 0.55 -> suppressed
    assert ref_cpu.nms(b, s, 0.5).tolist() == [0]


def test_descending_score_order_and_chain():
    # A suppresses B, B would suppress C, but B is dead so C survives (greedy semantics)
    b = np.array([[0, 0, 10, 10], [4, 0, 14, 10], [8, 0, 18, 10], [100, 100, 110, 110]],
                 dtype=np.float32)
    s = np.array([0.5, 0.9, 0.7, 0.6], dtype=np.float32)
    # order by score: 1, 2, 3, 0.  IoU(1,2)=6/14<0.5 keep; IoU(1,0)=6/14 keep -> all kept
    assert ref_cpu.nms(b, s, 0.5).tolist() == [1, 2, 3, 0]
    assert ref_cpu.nms(b, s, 0.4).tolist() == [1, 3]       # 6/14 = 0.43 > 0.4: 2 and 0 die


def test_equal_scores_stable_and_empty():
    b = np.array([[0, 0, 10, 10], [0, 0, 10, 10], [20, 20, 30, 30]], dtype=np.float32)
    s = np.array([0.5, 0.5, 0.5], dtype=np.float32)
    assert ref_cpu.nms(b, s, 0.5).tolist() == [0, 2]
    assert ref_cpu.nms(np.zeros((0, 4), np.float32), np.zeros((0,), np.float32), 0.5).shape == (0,)


def test_zero_area_boxes_do_not_suppress():
    b = np.array([[5, 5, 5, 5], [5, 5, 5, 5]], dtype=np.float32)    # 0/0 -> NaN > thr is False
    s = np.array([0.9, 0.8], dtype=np.float32)
    assert ref_cpu.nms(b, s, 0.5).tolist() == [0, 1]
