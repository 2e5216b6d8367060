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


def _object_noise_batch(z, frames, num_try=40):
    """the golden's frames as one collated batch + the reference's draws re-made from the same seeds"""
    pts = [z["f%d_points_in" % f] for f in frames]
    off = torch.tensor(np.cumsum([0] + [len(p) for p in pts]), dtype=torch.int32)
    gts = [z["f%d_gt_in" % f] for f in frames]
    g = max(len(x) for x in gts) + 2                      # two padding slots: they must stay inert
    gt = np.zeros((len(frames), g, 7), np.float32)
    valid = np.zeros((len(frames), g), bool)
    loc = np.zeros((len(frames), g, num_try, 3))
    rot = np.zeros((len(frames), g, num_try))
    for i, f in enumerate(frames):
        n = len(gts[i])
        gt[i, :n] = gts[i]
        valid[i, :n] = z["f%d_valid" % f]
        np.random.seed(11 + f)
        p = augment.sample_object_noise_params(gts[i], num_try, [-0.78539816, 0.78539816], [1.0, 1.0, 0.5], [0.0, 0.0])
        loc[i, :n], rot[i, :n] = p["loc_noises"], p["rot_noises"]
    count = torch.tensor([len(x) for x in gts])
    return torch.from_numpy(np.concatenate(pts)), off, torch.from_numpy(gt), count, \
        dict(loc_noises=loc, rot_noises=rot), torch.from_numpy(valid)


def check_per_object_noise(device):
    """f2: noise_per_object_v3_ (core/sampler/preprocess.py:587-709) on a collated batch: the same candidate is
    accepted for every box (collision tests incl. the boxes already moved), boxes and the points inside them land
    where the reference's numba functions put them (tests/golden/object_noise.npz, made by running those functions).
    `device`: where the batch lives (tests/test_pipeline_gpu.py runs the same check on CUDA tensors)."""
    z = np.load(os.path.join(GOLDEN, "object_noise.npz"))
    frames = [0, 1, 2]
    pts, off, gt, count, params, valid = _object_noise_batch(z, frames)
    pts, off, gt, valid = pts.to(device), off.to(device), gt.to(device), valid.to(device)
    new_pts, new_gt, selected = augment.noise_per_object(pts, off, gt, count, params, valid_mask=valid)
    assert new_pts.device.type == new_gt.device.type == torch.device(device).type
    new_pts, new_gt, selected, gt = new_pts.cpu(), new_gt.cpu(), selected.cpu(), gt.cpu()
    for i, f in enumerate(frames):
        n = int(count[i])
        ref_gt, ref_pts = z["f%d_gt_out" % f], z["f%d_points_out" % f]
        assert np.allclose(new_gt[i, :n].numpy(), ref_gt, atol=2e-5), f
        assert torch.equal(new_gt[i, n:], gt[i, n:])          # padding untouched
        a, b = int(off[i]), int(off[i + 1])
        got = new_pts[a:b].numpy()
        assert np.allclose(got, ref_pts, atol=5e-5), (f, np.abs(got - ref_pts).max())
        moved = (z["f%d_gt_in" % f][:, :2] != ref_gt[:, :2]).any(1)
        assert (selected[i, :n].numpy() >= 0).tolist() == moved.tolist()
        assert not moved[5] and int(selected[i, 5]) == -1      # valid_mask False: an obstacle that stays put
    assert (selected[:, -2:] == -1).all()


def test_per_object_noise_reproduces_the_reference_function():
    check_per_object_noise("cpu")


def test_per_object_noise_never_leaves_boxes_colliding():
    """property: after the walk no two real boxes overlap in the bird's-eye view unless they already did and found no
    free candidate; points that were in no box do not move."""
    rng = np.random.default_rng(5)
    b, g, t = 3, 10, 30
    gt = np.zeros((b, g, 7), np.float32)
    gt[..., 0] = rng.uniform(0, 40, (b, g))
    gt[..., 1] = rng.uniform(-20, 20, (b, g))
    gt[..., 3:6] = rng.uniform(1.5, 4.5, (b, g, 3))
    gt[..., 6] = rng.uniform(-3, 3, (b, g))
    count = torch.tensor([10, 7, 0])
    params = dict(loc_noises=rng.normal(scale=1.0, size=(b, g, t, 3)), rot_noises=rng.uniform(-0.78, 0.78, (b, g, t)))
    pts = torch.from_numpy(rng.uniform(-60, -50, (300, 4)).astype(np.float32))   # far from every box
    off = torch.tensor([0, 100, 200, 300], dtype=torch.int32)
    new_pts, new_gt, sel = augment.noise_per_object(pts, off, torch.from_numpy(gt), count, params)
    assert torch.equal(new_pts, pts)
    assert (sel[2] == -1).all() and (sel[1, 7:] == -1).all()
    for i in range(2):
        n = int(count[i])
        c = augment._bev_corners(new_gt[i, :n, 0:2], new_gt[i, :n, 3:5], new_gt[i, :n, 6])
        coll = augment._box_collision(c[:, None], c[None, :])
        coll.fill_diagonal_(False)
        moved = sel[i, :n] >= 0
        assert not coll[moved].any(), "a moved box collides with another box"


def test_per_object_noise_handles_empty_frames_and_empty_clouds():
    pts = torch.rand(10, 4)
    off = torch.tensor([0, 10], dtype=torch.int32)
    p, g, s = augment.noise_per_object(pts, off, torch.zeros((1, 0, 7)), torch.tensor([0]),
                                       dict(loc_noises=np.zeros((1, 0, 5, 3)), rot_noises=np.zeros((1, 0, 5))))
    assert torch.equal(p, pts) and g.shape == (1, 0, 7) and s.shape == (1, 0)
    boxes = torch.tensor([[[0.0, 0, 0, 1, 2, 1, 0], [10, 0, 0, 1, 2, 1, 0]]])
    p, g, s = augment.noise_per_object(torch.zeros((0, 4)), torch.tensor([0, 0], dtype=torch.int32), boxes,
                                       torch.tensor([2]), dict(loc_noises=np.full((1, 2, 3, 3), 0.25), rot_noises=np.zeros((1, 2, 3))))
    assert p.shape == (0, 4) and s.tolist() == [[0, 0]] and torch.allclose(g[0, :, :3], boxes[0, :, :3] + 0.25)
