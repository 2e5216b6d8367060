"""f2: ground-truth database sampling with the database resident in device memory (det3d_amd/gt_sampler.py) against
the reference's DataBaseSamplerV2 run here on a synthetic database (tests/golden/gt_sampler.npz): same objects, same
order, same survivors of the collision filter over four consecutive frames (the samplers are stateful), same pasted
cloud.  CPU tensors here; the class takes a device argument and only gathers / masks on it."""
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN

from det3d_amd.gt_sampler import BatchSampler, DeviceGTSampler, points_in_rbbox

CLASSES = ("Car", "Pedestrian", "Cyclist")
GROUPS = [dict(Car=15), dict(Pedestrian=8), dict(Cyclist=8)]
PREP = [dict(filter_by_min_num_points=dict(Car=5, Pedestrian=8, Cyclist=8)), dict(filter_by_difficulty=[-1])]


def _db(z):
    db = {}
    for name in CLASSES:
        boxes, npts, diff = z["db_%s_boxes" % name], z["db_%s_npts" % name], z["db_%s_difficulty" % name]
        db[name] = [dict(name=name, box3d_lidar=boxes[k], num_points_in_gt=int(npts[k]), difficulty=int(diff[k]),
                         points=z["db_%s_%d_points" % (name, k)]) for k in range(len(boxes))]
    return db


def test_device_gt_sampler_reproduces_the_reference_sampler():
    check_gt_sampler("cpu")


def check_gt_sampler(device):
    """`device`: where the database and the frame's cloud live (tests/test_pipeline_gpu.py: "cuda")"""
    z = np.load(os.path.join(GOLDEN, "gt_sampler.npz"))
    np.random.seed(123)
    sampler = DeviceGTSampler(_db(z), GROUPS, PREP, rate=1.0, device=device)
    assert sampler.points.device.type == torch.device(device).type
    for frame in range(4):
        gt, names, pts = z["f%d_gt" % frame], z["f%d_names" % frame], z["f%d_points" % frame]
        s = sampler.sample_all(gt, names)
        assert s is not None
        assert s["gt_names"].tolist() == z["f%d_s_gt_names" % frame].tolist(), frame
        assert np.array_equal(s["gt_boxes"], z["f%d_s_gt_boxes" % frame])
        assert np.array_equal(s["difficulty"], z["f%d_s_difficulty" % frame])
        assert np.array_equal(s["group_ids"], z["f%d_s_group_ids" % frame]) and s["gt_masks"].all()
        assert np.allclose(s["points"].cpu().numpy(), z["f%d_s_points" % frame], atol=1e-6)
        pasted, boxes, all_names, mask = sampler.paste(torch.from_numpy(pts).to(device), gt, names, s)
        ref = z["f%d_pasted_points" % frame]
        assert pasted.device.type == torch.device(device).type
        assert pasted.shape == ref.shape and np.allclose(pasted.cpu().numpy(), ref, atol=1e-6), frame
        assert len(boxes) == len(all_names) == len(mask) == len(gt) + len(s["gt_names"])
        # the sampled boxes collide neither with the frame's boxes nor with each other
        from det3d_amd.augment import _bev_corners, _box_collision

        b = torch.from_numpy(boxes.astype(np.float32))
        c = _bev_corners(b[:, 0:2], b[:, 3:5], b[:, -1])
        coll = _box_collision(c[:, None], c[None, :])
        coll.fill_diagonal_(False)
        assert not coll[len(gt):].any()


def test_filters_and_batch_sampler_follow_the_reference_rules():
    z = np.load(os.path.join(GOLDEN, "gt_sampler.npz"))
    db = _db(z)
    np.random.seed(0)
    s = DeviceGTSampler(db, GROUPS, PREP, device="cpu")
    for name, min_pts in (("Car", 5), ("Pedestrian", 8), ("Cyclist", 8)):
        kept = s.db_infos[name]
        assert all(i["num_points_in_gt"] >= min_pts and i["difficulty"] != -1 for i in kept)
        assert len(kept) == sum(1 for i in db[name] if i["num_points_in_gt"] >= min_pts and i["difficulty"] != -1)
    assert s.points.shape[0] == s.offsets_np[-1] == sum(i["num_points_in_gt"] for v in s.db_infos.values() for i in v)
    # a BatchSampler hands every index out once per pass, the short last slice included
    rng = np.random.RandomState(3)
    bs = BatchSampler(10, rng)
    got = np.concatenate([bs.sample(4), bs.sample(4), bs.sample(4)])
    assert sorted(got.tolist()) == list(range(10))
    with pytest.raises(NotImplementedError):
        DeviceGTSampler(db, [dict(Car=3, Cyclist=2)], device="cpu")


def test_nothing_to_sample_returns_none_and_paste_is_a_no_op():
    z = np.load(os.path.join(GOLDEN, "gt_sampler.npz"))
    np.random.seed(1)
    s = DeviceGTSampler(_db(z), [dict(Car=1)], PREP, device="cpu")
    gt, names = z["f0_gt"][:2], np.array(["Car", "Car"])
    assert s.sample_all(gt, names) is None
    pts = torch.from_numpy(z["f0_points"])
    out, boxes, all_names, mask = s.paste(pts, gt, names, None)
    assert out is pts and len(boxes) == 2 and mask.all()
    inside = points_in_rbbox(pts, torch.from_numpy(gt))
    assert inside.shape == (len(pts), 2)
