"""det3d_amd.augment (device-side global augmentation of a collated batch) vs the reference's NumPy functions
(random_flip / random_flip_both, global_rotation, global_scaling_v2) run with the same seeded NumPy stream
(tests/golden/augment.npz).  Pure tensor ops, so the same code is checked here on CPU tensors."""
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN

from det3d_amd import augment


@pytest.mark.parametrize("tag,flip,rot", [("kitti", "y", [-0.78539816, 0.78539816]), ("nusc", "both", [-0.3925, 0.3925])])
def test_batched_augmentation_reproduces_reference_per_frame(tag, flip, rot):
    z = np.load(os.path.join(GOLDEN, "augment.npz"))
    frames = range(4)
    params = dict(flip_y=[], flip_x=[], rotation=[], scale=[])
    for f in frames:  # the reference seeds per frame; draw each frame's parameters from the same stream
        np.random.seed(7 + f)
        p = augment.sample_global_params(1, flip=flip, rotation=rot, scaling=(0.95, 1.05), rng=np.random)
        for k in params:
            params[k].append(p[k][0])
    params = {k: np.array(v) for k, v in params.items()}
    assert params["flip_y"].any() and not params["flip_y"].all()  # both branches are exercised
    pts_in = [z["%s_f%d_points_in" % (tag, f)] for f in frames]
    off = torch.tensor(np.cumsum([0] + [len(p) for p in pts_in]), dtype=torch.int32)
    gt_in = np.stack([z["%s_f%d_gt_in" % (tag, f)] for f in frames])
    pts, gt = augment.apply_global_augmentation(torch.from_numpy(np.concatenate(pts_in)), off,
                                                torch.from_numpy(gt_in), params)
    for f in frames:
        a, b = int(off[f]), int(off[f + 1])
        want_p, want_g = z["%s_f%d_points_out" % (tag, f)], z["%s_f%d_gt_out" % (tag, f)]
        assert np.allclose(pts[a:b].numpy(), want_p, atol=2e-5), f
        assert np.allclose(gt[f].numpy(), want_g, atol=2e-5), f
    # inputs are not modified in place
    assert np.array_equal(gt_in, np.stack([z["%s_f%d_gt_in" % (tag, f)] for f in frames]))


def test_shuffle_is_a_per_frame_permutation():
    g = torch.Generator().manual_seed(0)
    pts = torch.arange(50, dtype=torch.float32)[:, None].repeat(1, 4)
    off = torch.tensor([0, 20, 20, 50], dtype=torch.int32)  # an empty frame in the middle
    out = augment.shuffle_points(pts, off, generator=g)
    assert sorted(out[:20, 0].tolist()) == list(range(20)) and sorted(out[20:, 0].tolist()) == list(range(20, 50))
    assert out[:20, 0].tolist() != list(range(20))
    assert augment.frame_index(off, 50).tolist() == [0] * 20 + [2] * 30


def test_identity_parameters_leave_the_batch_unchanged():
    pts = torch.randn(30, 5)
    gt = torch.randn(2, 3, 9)
    off = torch.tensor([0, 10, 30], dtype=torch.int32)
    p = dict(flip_y=np.zeros(2, bool), flip_x=np.zeros(2, bool), rotation=np.zeros(2), scale=np.ones(2))
    a, b = augment.apply_global_augmentation(pts, off, gt, p)
    assert torch.allclose(a, pts) and torch.allclose(b, gt)
