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


# ----------------------------------------------------------------------------------------- per-object noise
# noise_per_object_v3_ (core/sampler/preprocess.py:587-709, called from pipelines/preprocess.py:183-192 right after
# GT sampling): every ground-truth box gets its own small rotation about z and translation, chosen as the FIRST of
# `num_try` random candidates whose bird's-eye rectangle does not collide with any other (already moved) box; the
# points inside the original box move with it.  The reference runs this per frame in numba (noise_per_box :239-267,
# box_collision_test :877-960, points_transform_ :451-468).  Here the candidates of one box are tested for all frames
# and all tries at once on the device; only the walk over the boxes is sequential, as the algorithm demands.
# Group ids and the per-box "global rotation" variant (noise_per_box_v2_) are not used by any shipped config
# (global_random_rot_range = [0, 0]) and are not built.


def sample_object_noise_params(gt_boxes, num_try=100, rotation_perturb=np.pi / 4, center_noise_std=1.0,
                               global_random_rot_range=(0.0, 0.0), rng=np.random):
    """The reference's draws for ONE frame, same calls, same order (:609-626): loc_noises [N, T, 3] (normal),
    rot_noises [N, T] (uniform), and the per-box global-rotation draws, which the reference makes even when that
    variant is off (they advance the stream).  gt_boxes: numpy [N, >= 7]."""
    n = gt_boxes.shape[0]
    if not isinstance(rotation_perturb, (list, tuple, np.ndarray)):
        rotation_perturb = [-rotation_perturb, rotation_perturb]
    if not isinstance(global_random_rot_range, (list, tuple, np.ndarray)):
        global_random_rot_range = [-global_random_rot_range, global_random_rot_range]
    if abs(global_random_rot_range[0] - global_random_rot_range[1]) >= 1e-3:
        raise NotImplementedError("per-box global rotation (noise_per_box_v2_) is not on the Det3D hot path")
    if not isinstance(center_noise_std, (list, tuple, np.ndarray)):
        center_noise_std = [center_noise_std] * 3
    std = np.array(center_noise_std, dtype=gt_boxes.dtype)
    loc = rng.normal(scale=std, size=[n, num_try, 3])
    rot = rng.uniform(rotation_perturb[0], rotation_perturb[1], size=[n, num_try])
    grots = np.arctan2(gt_boxes[:, 0], gt_boxes[:, 1])
    rng.uniform((global_random_rot_range[0] - grots)[..., np.newaxis],
                (global_random_rot_range[1] - grots)[..., np.newaxis], size=[n, num_try])
    return dict(loc_noises=loc, rot_noises=rot)


def _bev_corners(xy, wl, r):
    """box2d_to_corner_jit (box_np_ops / preprocess.py:215-236): corners (-,-) (-,+) (+,+) (+,-) times (w, l), times
    [[c, -s], [s, c]] from the right, plus the centre.  [..., 2], [..., 2], [...] -> [..., 4, 2]"""
    norm = torch.tensor([[-0.5, -0.5], [-0.5, 0.5], [0.5, 0.5], [0.5, -0.5]], dtype=xy.dtype, device=xy.device)
    c = wl[..., None, :] * norm
    cos, sin = torch.cos(r)[..., None], torch.sin(r)[..., None]
    x = c[..., 0] * cos + c[..., 1] * sin
    y = -c[..., 0] * sin + c[..., 1] * cos
    return torch.stack([x, y], -1) + xy[..., None, :]


def _box_collision(a, q):
    """box_collision_test (:877-960) for rectangles a [..., 4, 2] against q [..., 4, 2] (broadcast): stand-up
    boxes must overlap, then any pair of edges that properly cross, or one rectangle inside the other."""
    amin, amax = a.min(-2).values, a.max(-2).values
    qmin, qmax = q.min(-2).values, q.max(-2).values
    iw = torch.minimum(amax[..., 0], qmax[..., 0]) - torch.maximum(amin[..., 0], qmin[..., 0])
    ih = torch.minimum(amax[..., 1], qmax[..., 1]) - torch.maximum(amin[..., 1], qmin[..., 1])
    near = (iw > 0) & (ih > 0)
    nxt = [1, 2, 3, 0]
    A, B = a[..., :, None, :], a[..., nxt, :][..., :, None, :]      # edges of a:  [..., 4, 1, 2]
    C, D = q[..., None, :, :], q[..., nxt, :][..., None, :, :]      # edges of q:  [..., 1, 4, 2]

    def ccw(p1, p2, p3):  # (p3.y - p1.y) (p2.x - p1.x) > (p2.y - p1.y) (p3.x - p1.x)
        return (p3[..., 1] - p1[..., 1]) * (p2[..., 0] - p1[..., 0]) > (p2[..., 1] - p1[..., 1]) * (p3[..., 0] - p1[..., 0])

    cross = (ccw(A, C, D) != ccw(B, C, D)) & (ccw(A, B, C) != ccw(A, B, D))
    hit = cross.flatten(-2).any(-1)

    def inside(outer, inner):  # every corner of `inner` strictly on the inner side of every edge of `outer`
        vec = outer[..., nxt, :] - outer                                       # clockwise: -(o_k - o_{k+1})
        dx = outer[..., :, None, 0] - inner[..., None, :, 0]                   # [..., k, l]
        dy = outer[..., :, None, 1] - inner[..., None, :, 1]
        cr = vec[..., :, None, 1] * dx - vec[..., :, None, 0] * dy
        return (cr < 0).flatten(-2).all(-1)

    return near & (hit | inside(a, q) | inside(q, a))


def noise_per_object(points, frame_offsets, gt_boxes, gt_count, params, valid_mask=None):
    """points [sum N, C], frame_offsets [B+1]; gt_boxes [B, G, >= 7] padded (x, y, z, w, l, h, r at index 6, like the
    reference, which reads gt_boxes[:, 6] whatever the width); gt_count [B]; params: loc_noises [B, G, T, 3],
    rot_noises [B, G, T] (per-frame draws of sample_object_noise_params, zero-padded); valid_mask [B, G] bool: boxes
    allowed to move (all real boxes by default; the others still are obstacles).  Returns (points, gt_boxes,
    selected [B, G] int64 with -1 = no collision-free candidate)."""
    dev = points.device
    gt = gt_boxes.clone().float()
    b, g = gt.shape[:2]
    loc = torch.as_tensor(np.asarray(params["loc_noises"]), device=dev).to(gt.dtype)
    rot = torch.as_tensor(np.asarray(params["rot_noises"]), device=dev).to(gt.dtype)
    t = rot.shape[-1]
    real = torch.arange(g, device=dev)[None, :] < gt_count.to(dev)[:, None]
    valid = real if valid_mask is None else (valid_mask.to(dev) & real)
    xy, wl, r0 = gt[..., 0:2], gt[..., 3:5], gt[..., 6]
    corners = _bev_corners(xy, wl, r0)                                   # [B, G, 4, 2], updated as boxes move
    # boxes that do not exist must never collide: park them far away, each at its own spot
    far = 1e6 + 1e3 * torch.arange(g, device=dev, dtype=gt.dtype)[None, :, None, None]
    corners = torch.where(real[..., None, None], corners, corners * 0 + far)
    selected = torch.full((b, g), -1, dtype=torch.int64, device=dev)
    others = torch.ones((g,), dtype=torch.bool, device=dev)
    movable = valid.any(0).tolist()   # one host read for the whole walk
    for i in range(g):
        if not movable[i]:
            continue
        cur = corners[:, i] - xy[:, i, None, :]                          # [B, 4, 2]
        cos, sin = torch.cos(rot[:, i])[:, :, None], torch.sin(rot[:, i])[:, :, None]   # [B, T, 1]
        cx = cur[:, None, :, 0] * cos + cur[:, None, :, 1] * sin         # corners @ [[c, -s], [s, c]]
        cy = -cur[:, None, :, 0] * sin + cur[:, None, :, 1] * cos
        cand = torch.stack([cx, cy], -1) + (xy[:, i, None, None, :] + loc[:, i, :, None, :2])   # [B, T, 4, 2]
        coll = _box_collision(cand[:, :, None], corners[:, None])        # [B, T, G]
        mask = others.clone()
        mask[i] = False
        free = ~(coll & mask[None, None, :]).any(-1)                     # [B, T]
        first = torch.where(free.any(-1), free.int().argmax(-1), torch.full((b,), -1, device=dev))
        first = torch.where(valid[:, i], first, torch.full_like(first, -1))
        selected[:, i] = first
        moved = first >= 0
        pick = cand[torch.arange(b, device=dev), first.clamp(min=0)]    # [B, 4, 2]
        corners[:, i] = torch.where(moved[:, None, None], pick, corners[:, i])
    # _select_transform (:479-486): the chosen try's noise, zeros where none was found
    idx = selected.clamp(min=0)
    has = (selected >= 0)
    loc_t = torch.gather(loc, 2, idx[..., None, None].expand(b, g, 1, 3))[:, :, 0] * has[..., None]
    rot_t = torch.gather(rot, 2, idx[..., None])[..., 0] * has
    # points inside the ORIGINAL boxes (centre origin, rotation about z) move with the first valid box that holds them
    pts = points.clone()
    if points.shape[0] and g:
        fid = frame_index(frame_offsets.to(dev), points.shape[0])
        p = points[:, :3].float()
        d = p[:, None, :] - gt[fid][:, :, :3]                            # [P, G, 3]
        cr, sr = torch.cos(r0)[fid], torch.sin(r0)[fid]
        lx = d[..., 0] * cr - d[..., 1] * sr                             # into the box frame (inverse of @ [[c,-s],[s,c]])
        ly = d[..., 0] * sr + d[..., 1] * cr
        half = gt[fid][:, :, 3:6] * 0.5
        inside = (lx.abs() < half[..., 0]) & (ly.abs() < half[..., 1]) & (d[..., 2].abs() < half[..., 2]) & valid[fid]
        hit = inside.any(-1)
        j = inside.int().argmax(-1)                                      # first box, as points_transform_ breaks
        c_j = gt[fid, j][:, :3]
        a = rot_t[fid, j]
        ca, sa = torch.cos(a), torch.sin(a)
        q = p - c_j
        qx = q[:, 0] * ca + q[:, 1] * sa
        qy = -q[:, 0] * sa + q[:, 1] * ca
        new = torch.stack([qx, qy, q[:, 2]], -1) + c_j + loc_t[fid, j]
        pts[:, :3] = torch.where(hit[:, None], new.to(points.dtype), points[:, :3])
    # box3d_transform_ (:471-476)
    out = gt_boxes.clone().float()
    out[..., :3] = out[..., :3] + loc_t * valid[..., None]
    out[..., 6] = out[..., 6] + rot_t * valid
    return pts, out, selected
