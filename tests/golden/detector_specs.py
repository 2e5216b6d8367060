"""Whole-detector golden cases shared by make_golden.py (runs the REFERENCE's detector classes) and the tests
(run the oracle model on CPU and the HIP product on the GPU).  Each case is one in-repo model config
(det3d_amd/configs/*.py -- field-for-field the reference's config, tests/test_models_cpu.py) on a CROPPED
point-cloud range, so the reference's CPU path finishes in seconds while every layer keeps its real width.
No reference imports here.
"""
import os

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))

CASES = {
    # name: config file, cropped range, voxel size, max points / voxel, points per frame
    "kitti_second": dict(config="kitti_car_second.py", range=[0, -6.4, -3.0, 12.8, 6.4, 1.0],
                         voxel=[0.05, 0.05, 0.1], T=5, C=4, points=5000),
    "nusc_cbgs": dict(config="nusc_cbgs.py", range=[-12.8, -12.8, -5.0, 12.8, 12.8, 3.0],
                      voxel=[0.1, 0.1, 0.2], T=4, C=5, points=6000),
    "lyft_cbgs": dict(config="lyft_cbgs.py", range=[-12.8, -12.8, -4.0, 12.8, 12.8, 2.0],
                      voxel=[0.1, 0.1, 0.15], T=4, C=4, points=6000),
    "nusc_pp": dict(config="nusc_point_pillars.py", range=[-12.8, -12.8, -5.0, 12.8, 12.8, 3.0],
                    voxel=[0.2, 0.2, 8.0], T=8, C=5, points=6000),
}
BATCH = 2
# arguments of oracle.model_ref.tame_head for the sane-head goldens (detectors_sane.pt): head magnitudes like a trained
# net's -- boxes within metres of their anchors, class scores on both sides of the config's score threshold
SANE_HEAD = {"kitti_second": dict(box_scale=0.1, cls_scale=0.5, cls_bias=0.5), "nusc_cbgs": {}, "lyft_cbgs": {}, "nusc_pp": {}}


def load_cfg(name):
    """the in-repo config with the case's cropped range applied (only PillarFeatureNet reads the range)."""
    from det3d_amd import Config

    case = CASES[name]
    cfg = Config.fromfile(os.path.join(ROOT, "det3d_amd", "configs", case["config"]))
    if cfg.model.reader.type == "PillarFeatureNet":
        cfg.model.reader.pc_range = tuple(case["range"])
        cfg.model.reader.voxel_size = tuple(case["voxel"])
    return cfg


def grid_of(case):
    r, v = np.array(case["range"], np.float64), np.array(case["voxel"], np.float64)
    return np.round((r[3:] - r[:3]) / v).astype(np.int64)  # (x, y, z)


def synth_points(case, seed):
    """ground slab + a few dense blobs, so that 3x3x3 neighbourhoods are populated"""
    rng = np.random.default_rng(seed)
    lo, hi = np.array(case["range"][:3]), np.array(case["range"][3:])
    n = case["points"]
    a = rng.uniform(lo, hi, (n // 2, 3))
    a[:, 2] = rng.normal(lo[2] + 0.35 * (hi[2] - lo[2]), 0.25, n // 2)
    centres = rng.uniform(lo[:2] + 2, hi[:2] - 2, (8, 2))
    which = rng.integers(0, 8, n - n // 2)
    b = np.concatenate([centres[which] + rng.normal(0, 0.7, (n - n // 2, 2)),
                        rng.uniform(lo[2] + 0.2 * (hi[2] - lo[2]), lo[2] + 0.7 * (hi[2] - lo[2]), (n - n // 2, 1))], 1)
    xyz = np.concatenate([a, b]).clip(lo + 1e-3, hi - 1e-3)
    extra = [rng.uniform(0, 1, (n, 1))]
    if case["C"] == 5:
        extra.append(rng.integers(0, 10, (n, 1)) * 0.05)
    pts = np.concatenate([xyz] + extra, 1).astype(np.float32)
    return pts[rng.permutation(n)]


def synth_targets(case, tasks, n_dim, code_size, fmap_hw, seed):
    """anchors on the feature-map grid (two rotations per class, like the reference's generators), random
    labels (80 % background, some ignored) and random regression targets -- the loss / decode arithmetic does
    not care where targets come from; target ASSIGNMENT has its own goldens."""
    g = torch.Generator().manual_seed(seed)
    H, W = fmap_hw
    lo, hi = case["range"][:3], case["range"][3:]
    anchors, labels, reg_targets = [], [], []
    for t in tasks:
        nc = len(t["class_names"])
        A = H * W * 2 * nc
        an = torch.zeros(BATCH, A, n_dim)
        ys, xs = torch.meshgrid(torch.arange(H).float(), torch.arange(W).float(), indexing="ij")
        base = torch.stack([lo[0] + (xs + 0.5) * (hi[0] - lo[0]) / W, lo[1] + (ys + 0.5) * (hi[1] - lo[1]) / H], -1)
        an[:, :, :2] = base.reshape(H * W, 1, 2).expand(H * W, 2 * nc, 2).reshape(1, A, 2)
        an[:, :, 2] = -1.0
        sizes = torch.tensor([[1.6 + 0.4 * c, 3.9 + 0.8 * c, 1.56] for c in range(nc)])
        an[:, :, 3:6] = sizes.repeat_interleave(2, 0).repeat(H * W, 1)
        an[:, :, -1] = torch.tensor([0.0, 1.57]).repeat(H * W * nc)
        lab = torch.randint(-1, nc + 1, (BATCH, A), generator=g, dtype=torch.int32)
        lab[torch.rand(BATCH, A, generator=g) < 0.8] = 0
        anchors.append(an)
        labels.append(lab)
        reg_targets.append(torch.randn(BATCH, A, code_size, generator=g) * 0.3)
    return anchors, labels, reg_targets


def param_fingerprint(module):
    """(name, shape) list: the product / oracle must expose exactly the reference's state-dict"""
    return [(k, tuple(v.shape)) for k, v in module.state_dict().items()]


DS_FACTOR = {"kitti_second": 8, "nusc_cbgs": 8, "lyft_cbgs": 8, "nusc_pp": 4}


def make_example(name, tasks, n_dim, code_size):
    """the case's `example` dict (collate_kitti keys, CPU tensors): oracle voxelizer (itself pinned to the
    reference's loop, tests/test_oracle_voxel_cpu.py) on the seeded clouds + synthetic targets.
    Returns (example, (H, W) of the head's feature map, voxels per frame)."""
    from oracle import clib

    case = CASES[name]
    grid = grid_of(case)
    vox, coors, nums, nv = [], [], [], []
    for b in range(BATCH):
        pts = synth_points(case, seed=100 + b)
        v, cc, n = clib.points_to_voxel(pts, case["voxel"], case["range"], case["T"], True, 20000)
        vox.append(v); nums.append(n); nv.append(len(cc))
        coors.append(np.pad(cc, ((0, 0), (1, 0)), mode="constant", constant_values=b))
    ds = DS_FACTOR[name]
    fmap = (int(grid[1]) // ds, int(grid[0]) // ds)
    anchors, labels, reg_targets = synth_targets(case, tasks, n_dim, code_size, fmap, seed=5)
    ex = dict(voxels=torch.from_numpy(np.concatenate(vox)), coordinates=torch.from_numpy(np.concatenate(coors)),
              num_points=torch.from_numpy(np.concatenate(nums)), num_voxels=np.array(nv, dtype=np.int64),
              shape=np.stack([grid] * BATCH), anchors=anchors, labels=labels, reg_targets=reg_targets,
              metadata=[dict(token=i) for i in range(BATCH)])
    return ex, fmap, nv


def check_input_sums(ex, sums):
    assert [int(v) for v in ex["num_voxels"]] == sums["num_voxels"]
    assert int(ex["coordinates"].long().sum()) == sums["coordinates"]
    assert int(ex["num_points"].long().sum()) == sums["num_points"]
    assert abs(float(ex["voxels"].double().sum()) - sums["voxels"]) <= 1e-6 * abs(sums["voxels"])
    for a, s in zip(ex["anchors"], sums["anchors"]):
        assert abs(float(a.double().sum()) - s) <= 1e-6 * abs(s) + 1e-6
    assert [int(l.long().sum()) for l in ex["labels"]] == sums["labels"]
    for t, s in zip(ex["reg_targets"], sums["reg_targets"]):
        assert abs(float(t.double().abs().sum()) - s) <= 1e-6 * s
