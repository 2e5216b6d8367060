"""GPU runs of the "next" rows that round 1 only checked on CPU tensors:
the device-side global augmentation against the reference's NumPy functions, and the BatchNorm-folded
inference model (sparse convs with the folded bias epilogue + folded RPN) against the unfolded one."""
import copy
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN, ROOT

pytestmark = pytest.mark.gpu

from test_preprocess_cpu import setup  # noqa: E402,F401  (fixture: synthetic GT database on disk + config factory)


@pytest.mark.parametrize("tag,flip,rot", [("kitti", "y", [-0.78539816, 0.78539816]), ("nusc", "both", [-0.3925, 0.3925])])
def test_device_augmentation_reproduces_the_reference_functions(tag, flip, rot):
    """f2: random_flip(_both) / global_rotation / global_scaling_v2 (core/sampler/preprocess.py:796-861) applied
    to a collated batch ON THE GPU, parameters drawn from the reference's seeded NumPy stream
    (tests/golden/augment.npz, produced by running the reference functions per frame)."""
    from det3d_amd import augment

    z = np.load(os.path.join(GOLDEN, "augment.npz"))
    frames = range(4)
    params = dict(flip_y=[], flip_x=[], rotation=[], scale=[])
    for f in frames:
        np.random.seed(7 + f)
        p = augment.sample_global_params(1, flip=flip, rotation=rot, scaling=(0.95, 1.05), rng=np.random)
        for k in params:
            params[k].append(p[k][0])
    params = {k: np.array(v) for k, v in params.items()}
    pts_in = [z["%s_f%d_points_in" % (tag, f)] for f in frames]
    off = torch.tensor(np.cumsum([0] + [len(p) for p in pts_in]), dtype=torch.int32, device="cuda")
    gt_in = torch.from_numpy(np.stack([z["%s_f%d_gt_in" % (tag, f)] for f in frames])).cuda()
    pts, gt = augment.apply_global_augmentation(torch.from_numpy(np.concatenate(pts_in)).cuda(), off, gt_in, params)
    assert pts.is_cuda and gt.is_cuda
    for f in frames:
        a, b = int(off[f]), int(off[f + 1])
        assert np.allclose(pts[a:b].cpu().numpy(), z["%s_f%d_points_out" % (tag, f)], atol=2e-5), f
        assert np.allclose(gt[f].cpu().numpy(), z["%s_f%d_gt_out" % (tag, f)], atol=2e-5), f
    # the augmented cloud goes straight into the device voxelizer: shuffle + voxelize stay consistent
    sh = augment.shuffle_points(pts, off, generator=torch.Generator(device="cuda").manual_seed(0))
    assert sh.shape == pts.shape and torch.allclose(sh.sum(0), pts.sum(0), rtol=1e-4, atol=1e-2)


@pytest.mark.parametrize("name", ["kitti_car_second.py", "nusc_cbgs.py"])
def test_bn_folded_detector_matches_the_unfolded_one_on_the_gpu(name):
    """f4: fuse_bn_recursively (the reference hints at it in tools/dist_test.py:164) on a whole detector --
    sparse conv + BN1d folded into weight + bias (bias epilogue of d3d_sparse_conv_fwd), Conv2d / ConvTranspose2d
    + BN2d folded -- must reproduce the eval-mode predictions of the unfolded model on the GPU."""
    from det3d_amd import Config, build_detector, checkpoint, synthetic
    from oracle import model_ref as M

    cfg = Config.fromfile(os.path.join(ROOT, "det3d_amd", "configs", name))
    model = build_detector(cfg.model, cfg.train_cfg, cfg.test_cfg)
    M.fill_deterministic(model)
    model = model.cuda().eval()
    folded = copy.deepcopy(model)
    n_bn = sum(isinstance(m, (torch.nn.BatchNorm1d, torch.nn.BatchNorm2d)) for m in folded.modules())
    assert checkpoint.fuse_bn_recursively(folded) == n_bn > 0
    assert not any(isinstance(m, (torch.nn.BatchNorm1d, torch.nn.BatchNorm2d)) for m in folded.modules())
    dev = torch.device("cuda")
    if name.startswith("kitti"):
        ds = synthetic.KittiCarSynthetic(n_points=8000, n_seeds=2)
        ex = synthetic.make_example(ds.device_batch(0, 2, dev), synthetic.KITTI_CAR)
    else:
        ds = synthetic.TaskGroupSynthetic(cfg, 30000, n_seeds=2)
        ex = ds.make_example(ds.device_batch(0, 2, dev), targets=False)
    data = dict(features=ex["voxel_mean"], num_voxels=ex["num_points"], coors=ex["coordinates"], batch_size=2,
                input_shape=ex["shape"][0])
    with torch.no_grad():
        want = model.bbox_head(model.extract_feat(data))
        got = folded.bbox_head(folded.extract_feat(data))
    for w, g in zip(want, got):
        for k in w:
            scale = float(w[k].abs().max())
            assert float((w[k] - g[k]).abs().max()) < 1e-4 * max(scale, 1.0), k


def test_per_object_noise_on_the_gpu_reproduces_the_reference_function():
    """f2 on the device: noise_per_object_v3_ (core/sampler/preprocess.py:587-709) on a collated CUDA batch -- same
    accepted candidates, boxes and points as the reference's numba functions (tests/golden/object_noise.npz)"""
    from test_augment_cpu import check_per_object_noise

    check_per_object_noise("cuda")


def test_gt_database_sampling_on_the_gpu_reproduces_the_reference_sampler():
    """f2 on the device: DataBaseSamplerV2 (core/sampler/sample_ops.py) with the database resident in HBM: same
    objects / survivors / pasted cloud over four consecutive frames (tests/golden/gt_sampler.npz)"""
    from test_gt_sampler_cpu import check_gt_sampler

    check_gt_sampler("cuda")


def test_preprocess_step_on_the_gpu_matches_the_reference_step(setup):
    """f2 on the device: the whole Preprocess step (datasets/pipelines/preprocess.py:215-217 shuffle included) with
    the cloud on the GPU: same boxes, names and points IN THE SAME ORDER (tests/golden/preprocess.npz)"""
    from test_preprocess_cpu import check_kitti_train_step

    check_kitti_train_step(setup, "cuda")


def test_preprocessed_frames_go_through_the_geometry_prefetcher(setup):
    """the device pipeline in front of the training step: Preprocess (sampling, noise, global augmentation, shuffle)
    on the side stream of GeometryPrefetcher, its output straight into the device voxelizer -- no host copy of the
    cloud between augmentation and voxelization"""
    from det3d_amd import pipelines, synthetic
    from det3d_amd.prefetch import GeometryPrefetcher

    z, tmp, cfg, _ = setup
    np.random.seed(7)
    step = pipelines.Preprocess(cfg=cfg("train"), device="cuda", root_path=str(tmp))
    dev = torch.device("cuda", 0)
    template = synthetic.KittiCarSynthetic(n_points=2000, n_seeds=1).device_batch(0, 1, dev)  # anchors of the config

    def make_example(f):
        res = dict(type="KittiDataset", metadata=dict(image_prefix=str(tmp), num_point_features=4),
                   lidar=dict(points=z["kitti%d_points" % f].copy(),
                              annotations=dict(boxes=z["kitti%d_gt" % f].copy(), names=z["kitti%d_names" % f].copy())))
        res, _ = step(res, {})
        pts = res["lidar"]["points"]
        assert pts.is_cuda
        an = res["lidar"]["annotations"]
        cars = torch.from_numpy(an["gt_boxes"][an["gt_names"] == "Car"].astype(np.float32)).to(dev)
        batch = dict(template, points=pts.float().contiguous(),
                     frame_offsets=torch.tensor([0, len(pts)], dtype=torch.int32, device=dev),
                     gt_boxes=cars[None].contiguous(), gt_count=torch.tensor([len(cars)], dtype=torch.int32, device=dev))
        return synthetic.make_example(batch, synthetic.KITTI_CAR, device_targets=True)

    pf = GeometryPrefetcher(make_example, None, dev)
    for f in range(2):
        ex = pf.prepare(f)
        torch.cuda.synchronize()
        assert ex["coordinates"].is_cuda and int(ex["coordinates"].shape[0]) > 0
        assert ex["labels"][0].shape[0] == 1 and int(ex["labels"][0].max()) >= 0
        assert torch.isfinite(ex["voxel_mean"]).all()
