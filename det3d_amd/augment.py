"""Global train-time augmentation of a collated batch on the device.

The reference applies these per frame with NumPy inside DataLoader workers, after GT sampling and before
voxelization (det3d/datasets/pipelines/preprocess.py:205-217 -> det3d/core/sampler/preprocess.py):
    random_flip :816-826 (KITTI / Lyft)  |  random_flip_both :829-854 (nuScenes)
    global_rotation :796-813  (rotation_points_single_angle axis 2, box_np_ops.py:394-416: clockwise-positive)
    global_scaling_v2 :857-861
    np.random.shuffle(points)  preprocess.py:215-217
With the voxelizer on the GPU the raw points are already resident, so the same transforms run there on the
whole batch at once: a per-frame 2x2 rotation-flip matrix and scale applied through a frame index, a handful of
element-wise launches over [sum N, 3] instead of B NumPy passes on the host.  `sample_global_params` draws the
parameters with the reference's own calls in the reference's order, so a seeded NumPy stream reproduces the
reference's augmentation exactly (tests/golden/augment.npz).
"""
import numpy as np
import torch


def sample_global_params(batch, flip="y", rotation=np.pi / 4, scaling=(0.95, 1.05), rng=np.random):
    """Per-frame draws, same calls and order as the reference: flip(s) -> rotation -> scale.
    flip: "y" = random_flip (mirror y), "both" = random_flip_both (then also mirror x), None = off."""
    rot = list(rotation) if isinstance(rotation, (list, tuple)) else [-rotation, rotation]
    out = dict(flip_y=np.zeros(batch, bool), flip_x=np.zeros(batch, bool),
               rotation=np.zeros(batch, np.float64), scale=np.ones(batch, np.float64))
    for b in range(batch):
        if flip in ("y", "both"):
            out["flip_y"][b] = rng.choice([False, True], replace=False, p=[0.5, 0.5])
        if flip == "both":
            out["flip_x"][b] = rng.choice([False, True], replace=False, p=[0.5, 0.5])
        out["rotation"][b] = rng.uniform(rot[0], rot[1])
        out["scale"][b] = rng.uniform(scaling[0], scaling[1])
    return out


def frame_index(frame_offsets, n_points):
    """[B+1] offsets of the collated cloud -> frame id of every point, on the offsets' device."""
    counts = (frame_offsets[1:] - frame_offsets[:-1]).long()
    return torch.repeat_interleave(torch.arange(counts.numel(), device=frame_offsets.device), counts,
                                   output_size=int(n_points))


def apply_global_augmentation(points, frame_offsets, gt_boxes, params):
    """points [sum N, C] (xyz first), gt_boxes [B, G, 7 | 9] padded (x, y, z, w, l, h, [vx, vy,] r).
    Returns new tensors; the arithmetic follows the reference statement by statement (float32)."""
    dev = points.device
    f32 = dict(dtype=torch.float32, device=dev)
    fy = torch.as_tensor(params["flip_y"], device=dev)
    fx = torch.as_tensor(params["flip_x"], device=dev)
    rot64 = np.asarray(params["rotation"], np.float64)
    ang = torch.as_tensor(rot64.astype(np.float32), **f32)
    sc = torch.as_tensor(np.asarray(params["scale"], np.float32), **f32)
    sy = torch.where(fy, -1.0, 1.0).to(**f32)   # y -> -y   (random_flip / first half of random_flip_both)
    sx = torch.where(fx, -1.0, 1.0).to(**f32)   # x -> -x   (second half of random_flip_both)
    # cos / sin in float64 on the host, then rounded to float32 -- what rotation_points_single_angle does
    cos = torch.as_tensor(np.cos(rot64).astype(np.float32), **f32)
    sin = torch.as_tensor(np.sin(rot64).astype(np.float32), **f32)

    def rot_scale(x, y, idx):
        x, y = x * sx[idx], y * sy[idx]
        return x * cos[idx] + y * sin[idx], -x * sin[idx] + y * cos[idx]   # points @ [[c,-s],[s,c]]

    pts = points.clone()
    fid = frame_index(frame_offsets, points.shape[0])
    px, py = rot_scale(points[:, 0], points[:, 1], fid)
    s_p = sc[fid]
    pts[:, 0], pts[:, 1], pts[:, 2] = px * s_p, py * s_p, points[:, 2] * s_p

    gt = gt_boxes.clone().float()
    b = gt.shape[0]
    bid = torch.arange(b, device=dev)[:, None].expand(b, gt.shape[1])
    gx, gy = rot_scale(gt[..., 0], gt[..., 1], bid)
    r = gt[..., -1]
    r = torch.where(fy[bid], -r + np.pi, r)
    r = torch.where(fx[bid], -r + 2 * np.pi, r)
    r = r + ang[bid]
    s_g = sc[bid]
    gt[..., 0], gt[..., 1] = gx * s_g, gy * s_g
    gt[..., 2:6] = gt_boxes[..., 2:6].float() * s_g[..., None]
    if gt.shape[-1] > 7:   # velocities flip and rotate with the frame; global_scaling_v2 scales gt[:, :-1], so them too
        vx, vy = rot_scale(gt_boxes[..., 6].float(), gt_boxes[..., 7].float(), bid)
        gt[..., 6], gt[..., 7] = vx * s_g, vy * s_g
    gt[..., -1] = r
    return pts, gt


def shuffle_points(points, frame_offsets, generator=None):
    """np.random.shuffle(points) per frame, on the device: a random permutation inside every frame (one sort of
    frame_id + U[0,1) keys).  The voxelizer's output depends on the point order (first point of a cell numbers
    the voxel, the first T points are kept), which is why the reference shuffles in train mode."""
    n = points.shape[0]
    fid = frame_index(frame_offsets, n)
    keys = fid.double() + torch.rand(n, device=points.device, generator=generator, dtype=torch.float64)
    return points[torch.argsort(keys)]
