"""d3d_assign_targets (device AssignTarget) vs the numpy restatement, which is pinned to the
reference's create_target_np by tests/golden/anchors_targets.npz."""
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN

pytestmark = pytest.mark.gpu


def test_matches_reference_golden():
    from det3d_amd import synthetic

    z = np.load(os.path.join(GOLDEN, "anchors_targets.npz"))
    anchors = z["anchors6d"].reshape(-1, 7)
    gt = z["gt"]
    lab, tgt, w = synthetic.assign_targets_device(torch.from_numpy(anchors).cuda(), torch.from_numpy(gt[None]).cuda(),
                                                  torch.tensor([len(gt)], dtype=torch.int32).cuda(), 0.6, 0.45)
    assert np.array_equal(lab[0].cpu().numpy(), z["labels"])
    assert np.array_equal(w[0].cpu().numpy(), z["weights"])
    assert np.allclose(tgt[0].cpu().numpy(), z["bbox_targets"], atol=1e-6)


def test_full_kitti_anchor_set_batched_and_ragged():
    from det3d_amd import synthetic

    ds = synthetic.KittiCarSynthetic(n_points=2000, n_seeds=3)
    rng = np.random.default_rng(0)
    gts = [synthetic.kitti_like_cloud(s, 100, n_cars=k)[1] for s, k in ((0, 8), (1, 1), (2, 0), (3, 23))]
    gts[0][1, :2] = ds.anchors[5000, :2]  # a gt exactly on an anchor centre (tie rows)
    gmax = max(len(g) for g in gts)
    pad = np.zeros((len(gts), gmax, 7), np.float32)
    for i, g in enumerate(gts):
        pad[i, : len(g)] = g
    cnt = np.array([len(g) for g in gts], np.int32)
    lab, tgt, w = synthetic.assign_targets_device(torch.from_numpy(ds.anchors).cuda(), torch.from_numpy(pad).cuda(),
                                                  torch.from_numpy(cnt).cuda(), 0.6, 0.45)
    for i, g in enumerate(gts):
        rl, rt, rw = synthetic.assign_targets(ds.anchors, g, 0.6, 0.45)
        assert np.array_equal(lab[i].cpu().numpy(), rl), i
        assert np.array_equal(w[i].cpu().numpy(), rw)
        assert np.allclose(tgt[i].cpu().numpy(), rt, atol=1e-6)
    assert int((lab[2] != 0).sum()) == 0  # frame without gt: all background


def test_make_example_device_targets_equal_cached():
    from det3d_amd import synthetic

    ds = synthetic.KittiCarSynthetic(n_points=3000, n_seeds=2)
    b = ds.device_batch(0, 2, torch.device("cuda"))
    a = synthetic.make_example(b, synthetic.KITTI_CAR, device_targets=True)
    c = synthetic.make_example(b, synthetic.KITTI_CAR, device_targets=False)
    assert torch.equal(a["labels"][0], c["labels"][0])
    assert torch.allclose(a["reg_targets"][0], c["reg_targets"][0], atol=1e-6)
