"""Pins the iou3d / evaluation-IoU oracle (oracle/iou3d_oracle.c) to the reference itself:
  - iou3d_kernel.cu device functions compiled for the host -> oracle/_ref/d3d_ref_iou3d
  - rotate_iou.py devRotateIoUEval run as Python           -> tests/golden/rotate_iou_eval.npz
"""
import os

import numpy as np
import pytest

from conftest import GOLDEN

from oracle import clib


def iou3d_boxes(n, seed, spread=6.0):
    rng = np.random.default_rng(seed)
    c = rng.uniform(0, spread, (n, 2))
    d = rng.uniform(0.8, 4.0, (n, 2))
    r = rng.uniform(-3.2, 3.2, (n, 1))
    return np.concatenate([c - d / 2, c + d / 2, r], 1).astype(np.float32)


@pytest.fixture(scope="module")
def ref3d():
    mod = clib.ref_iou3d_module()
    if mod is None:
        pytest.skip("oracle/_ref/d3d_ref_iou3d not built (needs /root/reference once)")
    return mod


@pytest.mark.parametrize("seed", [0, 1, 2])
def test_iou3d_oracle_equals_compiled_reference(ref3d, seed):
    a, b = iou3d_boxes(70, seed), iou3d_boxes(90, seed + 100)
    b[:10] = a[:10]  # identical boxes
    b[10:20, 4] = a[10:20, 4] + np.float32(np.pi / 2)
    for mode, fn in (("overlap", ref3d.boxes_overlap_bev), ("iou_bev", ref3d.boxes_iou_bev),
                     ("iou_normal", ref3d.boxes_iou_normal)):
        mine, ref = clib.iou3d_pairwise(a, b, mode), fn(a, b)
        assert np.array_equal(np.isnan(mine), np.isnan(ref))
        assert np.nanmax(np.abs(mine - ref)) == 0.0, mode  # same float32 operations, same libm


@pytest.mark.parametrize("normal", [False, True])
@pytest.mark.parametrize("thresh", [0.1, 0.5])
def test_iou3d_nms_oracle_equals_compiled_reference(ref3d, normal, thresh):
    boxes = iou3d_boxes(400, 5, spread=12.0)
    mine = clib.iou3d_nms(boxes, thresh, normal)
    ref = (ref3d.nms_normal if normal else ref3d.nms)(boxes, thresh)
    assert mine == list(ref) and 0 < len(mine) < 400


def test_iou3d_axis_aligned_closed_form():
    a = np.array([[0, 0, 2, 2, 0]], np.float32)
    b = np.array([[1, 1, 3, 3, 0], [0, 0, 2, 2, 0], [5, 5, 6, 6, 0.3]], np.float32)
    ov = clib.iou3d_pairwise(a, b, "overlap")[0]
    assert abs(ov[0] - 1.0) < 1e-6 and abs(ov[1] - 4.0) < 1e-5 and ov[2] == 0.0
    iou = clib.iou3d_pairwise(a, b, "iou_bev")[0]
    assert abs(iou[0] - 1 / 7) < 1e-6 and abs(iou[1] - 1.0) < 1e-5


def test_rotate_iou_eval_matches_reference_numba_golden():
    z = np.load(os.path.join(GOLDEN, "rotate_iou_eval.npz"))
    r1, r2 = z["rbox1"], z["rbox2"]
    for ci, crit in enumerate(z["criteria"]):
        # value[i] = devRotateIoUEval(rbox1[i], rbox2[i]); the batched call puts rbox1 on the query axis
        mine = np.array([clib.rotate_iou_eval(r2[i:i + 1], r1[i:i + 1], int(crit))[0, 0] for i in range(len(r1))])
        assert np.abs(mine - z["value"][:, ci]).max() < 2e-5, crit
    assert (z["value"][:, 0] > 0.05).sum() > 40
