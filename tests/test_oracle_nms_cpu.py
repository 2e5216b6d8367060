"""Rotated IoU / NMS oracle (oracle/nms_oracle.c) against the two other formulations
that live in the reference tree:
  - the numba-CUDA devRotateIoU (nms_gpu.py:180-416) run as Python -> rotate_iou.npz
  - detectron2's single_box_iou_rotated (box_iou_rotated_utils.h:311-338), compiled
    into oracle/_ref (degrees, counter-clockwise => pass -r)."""
import os

import numpy as np
import pytest

from conftest import GOLDEN


def test_rotate_iou_matches_reference_numba_golden(oracle_lib):
    z = np.load(os.path.join(GOLDEN, "rotate_iou.npz"))
    a, b, gold = z["a"], z["b"], z["iou"]
    mine = np.array([oracle_lib.rotate_iou(a[i:i + 1], b[i:i + 1])[0, 0] for i in range(len(a))])
    # reference arithmetic is float32 with a different (vertex sort + fan) algorithm
    assert np.abs(mine - gold).max() < 2e-4
    assert gold[0] == pytest.approx(1.0, abs=1e-5) and gold[3] == 0.0


def test_rotate_iou_matches_compiled_detectron2(oracle_lib):
    mod = oracle_lib.ref_module()
    if mod is None:
        pytest.skip("oracle/_ref not built")
    rng = np.random.default_rng(3)
    for _ in range(300):
        a = np.array([rng.uniform(0, 4), rng.uniform(0, 4), rng.uniform(.5, 3), rng.uniform(.5, 5), rng.uniform(-3, 3)])
        b = a + rng.normal(0, 0.7, 5)
        b[2:4] = np.abs(b[2:4]) + 0.2
        mine = oracle_lib.rotate_iou(a[None].astype(np.float32), b[None].astype(np.float32))[0, 0]
        af, bf = a.astype(np.float32).astype(np.float64), b.astype(np.float32).astype(np.float64)
        ref = mod.box_iou_rotated_f64(np.array([af[0], af[1], af[2], af[3], -np.degrees(af[4])]),
                                      np.array([bf[0], bf[1], bf[2], bf[3], -np.degrees(bf[4])]))
        assert mine == pytest.approx(ref, abs=5e-6)


def _brute_nms(dets, thresh, iou_fn):
    order = dets[:, 5].argsort()[::-1]
    keep, sup = [], np.zeros(len(dets), bool)
    for a, i in enumerate(order):
        if sup[i]:
            continue
        keep.append(int(i))
        for j in order[a + 1:]:
            if not sup[j] and iou_fn(dets[i, :5], dets[j, :5]) >= thresh:
                sup[j] = True
    return keep


def test_rotate_nms_greedy_semantics(oracle_lib):
    rng = np.random.default_rng(0)
    n = 120
    dets = np.stack([rng.uniform(0, 20, n), rng.uniform(0, 20, n), rng.uniform(1.4, 2, n),
                     rng.uniform(3.5, 4.5, n), rng.uniform(-3, 3, n), rng.permutation(n) / n], 1).astype(np.float32)
    for thr in (0.01, 0.2, 0.5):
        keep, margin = oracle_lib.rotate_nms_cc(dets, thr, return_margin=True)
        ref = _brute_nms(dets, thr, lambda p, q: oracle_lib.rotate_iou(p[None], q[None])[0, 0])
        assert keep == ref, (thr, margin)
    assert oracle_lib.rotate_nms_cc(np.zeros((0, 6), np.float32), 0.5) == []


def test_corners_follow_reference_convention(oracle_lib):
    # center_to_corner_box2d: clockwise start at the minimum point, +r rotates clockwise
    c = oracle_lib.box_corners(np.array([[1, 2, 2, 4, 0.0]], np.float32))[0]
    assert np.allclose(c, [[0, 0], [0, 4], [2, 4], [2, 0]])
    c = oracle_lib.box_corners(np.array([[0, 0, 2, 4, np.pi / 2]], np.float32))[0]
    assert np.allclose(c, [[-2, 1], [2, 1], [2, -1], [-2, -1]], atol=1e-6)


def test_axis_aligned_nms_plus_one_convention(oracle_lib):
    boxes = np.array([[0, 0, 9, 9], [0, 0, 9, 4], [20, 20, 29, 29]], np.float32)
    # IoU(+1) of box0/box1 = 50/100 = 0.5 -> strict `>` keeps both at thr 0.5
    assert oracle_lib.nms_sorted(boxes, 0.5) == [0, 1, 2]
    assert oracle_lib.nms_sorted(boxes, 0.49) == [0, 2]


@pytest.mark.parametrize("case", ["dense", "sparse", "tiny", "dupes"])
def test_axis_aligned_nms_vs_reference_nms_gpu_golden(oracle_lib, case):
    """a12: oracle nms_sorted == the reference's own nms_gpu.py (iou_device :22-33 with the +1-pixel
    convention, nms_kernel :67-101 `>` threshold, nms_postprocess :109-126) executed as plain Python
    (tests/golden/make_golden.py::nms_gpu_cases) -- kept indices in order, incl. exact duplicates."""
    g = np.load(os.path.join(GOLDEN, "nms_gpu.npz"))
    dets, thr, keep = g[case + "_dets"], float(g[case + "_thr"]), g[case + "_keep"]
    order = dets[:, 4].argsort()[::-1]  # nms_gpu(): `order = dets[:, 4].argsort()[::-1]`
    got = [int(order[i]) for i in oracle_lib.nms_sorted(dets[order, :4], thr)]
    assert got == keep.tolist()
