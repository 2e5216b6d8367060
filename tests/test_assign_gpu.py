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


@pytest.mark.parametrize("tag", ["nusc", "lyft"])
def test_task_group_matches_reference_assign_v2_golden(tag):
    """d3d_assign_targets_task vs the reference's TargetAssigner.assign_v2 output (tests/golden/task_targets.npz)."""
    from det3d_amd import synthetic
    from test_models_cpu import TASK_GENS

    z = np.load(os.path.join(GOLDEN, "task_targets.npz"))
    nd, vec, gens = TASK_GENS[tag]
    anchors, k, r, hi, lo = synthetic.task_anchors([1, 32, 32], gens)
    gts = [z["%s_f%d_gt" % (tag, f)] for f in range(3)]
    gmax = max(max(len(g) for g in gts), 1)
    pad = np.zeros((3, gmax, nd), np.float32)
    cls = np.zeros((3, gmax), np.int32)
    for f, g in enumerate(gts):
        pad[f, : len(g)] = g
        cls[f, : len(g)] = z["%s_f%d_cls" % (tag, f)]
    cnt = np.array([len(g) for g in gts], np.int32)
    lab, tgt, w = synthetic.assign_targets_task_device(
        torch.from_numpy(anchors).cuda(), k, r, hi, lo, torch.from_numpy(pad).cuda(), torch.from_numpy(cls).cuda(),
        torch.from_numpy(cnt).cuda(), encode_angle_vector=vec)
    for f in range(3):
        assert np.array_equal(lab[f].cpu().numpy(), z["%s_f%d_labels" % (tag, f)]), f
        assert np.array_equal(w[f].cpu().numpy(), z["%s_f%d_weights" % (tag, f)])
        assert np.allclose(tgt[f].cpu().numpy(), z["%s_f%d_targets" % (tag, f)], atol=2e-6)


def test_task_group_full_nuscenes_feature_map_vs_numpy():
    """the [bus, trailer] group on the real 128x128 CBGS feature map (65 536 anchors), 4 ragged frames."""
    from det3d_amd import synthetic
    from test_models_cpu import TASK_GENS

    nd, vec, gens = TASK_GENS["nusc"]
    anchors, k, r, hi, lo = synthetic.task_anchors([1, 128, 128], gens)
    rng = np.random.default_rng(3)
    frames = []
    for n in (12, 1, 0, 30):
        gt = np.zeros((n, nd), np.float32)
        c = rng.integers(1, 3, n).astype(np.int32)
        gt[:, :2] = rng.uniform(-45, 45, (n, 2))
        for i in range(n):
            gt[i, 2] = gens[c[i] - 1]["anchor_ranges"][2]
            gt[i, 3:6] = np.array(gens[c[i] - 1]["sizes"]) * rng.uniform(0.8, 1.2, 3)
        gt[:, 6:8] = rng.normal(0, 3, (n, 2))
        gt[:, 8] = rng.uniform(-3.1, 3.1, n)
        frames.append((gt, c))
    gmax = 30
    pad = np.zeros((4, gmax, nd), np.float32)
    cls = np.zeros((4, gmax), np.int32)
    for f, (g, c) in enumerate(frames):
        pad[f, : len(g)], cls[f, : len(g)] = g, c
    cnt = np.array([len(g) for g, _ in frames], np.int32)
    lab, tgt, w = synthetic.assign_targets_task_device(
        torch.from_numpy(anchors).cuda(), k, r, hi, lo, torch.from_numpy(pad).cuda(), torch.from_numpy(cls).cuda(),
        torch.from_numpy(cnt).cuda(), encode_angle_vector=vec)
    npos = 0
    for f, (g, c) in enumerate(frames):
        rl, rt, rw = synthetic.assign_targets_task(anchors, k, r, hi, lo, g, c, encode_angle_vector=vec)
        assert np.array_equal(lab[f].cpu().numpy(), rl), f
        assert np.array_equal(w[f].cpu().numpy(), rw)
        assert np.allclose(tgt[f].cpu().numpy(), rt, atol=2e-6)
        npos += int((rl > 0).sum())
    assert npos > 40 and tgt.shape == (4, anchors.shape[0], 10)


def test_task_group_argument_errors():
    from det3d_amd import _lib, synthetic

    a = torch.zeros((8, 7), device="cuda")
    g = torch.zeros((1, 1, 7), device="cuda")
    c = torch.zeros((1,), dtype=torch.int32, device="cuda")
    with pytest.raises(_lib.D3dError):  # 8 anchors are not a multiple of 3 classes x 2
        synthetic.assign_targets_task_device(a, 3, 2, [0.5] * 3, [0.3] * 3, g, torch.ones((1, 1), device="cuda"), c)
    with pytest.raises(_lib.D3dError):  # two classes need gt_classes
        synthetic.assign_targets_task_device(a, 2, 2, [0.5] * 2, [0.3] * 2, g, None, c)
