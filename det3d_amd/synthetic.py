"""Seeded synthetic inputs for parity tests and the benchmark (no datasets offline).

  kitti_like_cloud     SURVEY.md 8d: 64-beam spinning lidar restricted to the camera FOV,
                       ground plane at sensor height 1.73 m, exponential obstacles, planted
                       car-sized boxes with surface points
  create_anchors_3d_range   restates det3d/core/bbox/box_np_ops.py:733-805
  assign_targets       restates the single-class path of create_target_np
                       (det3d/core/anchor/target_ops.py:28-222) with the nearest-IoU
                       similarity (region_similarity.py:73-93, box_np_ops.py:343-355,955-994)
  make_example         the `example` dict collate_kitti would hand to the detector
                       (det3d/torchie/parallel/collate.py:90-150), built from raw points by
                       the device voxelizer
Targets are "next" rows in SURVEY.md 8f: computed once per seed on the host (numpy) and
cached, so timed loops measure the hot path, not label assignment.
"""
import numpy as np
import torch

CAR_SIZE = (1.6, 3.9, 1.56)  # w, l, h

KITTI_CAR = dict(
    voxel_size=[0.05, 0.05, 0.1], pc_range=[0, -40.0, -3.0, 70.4, 40.0, 1.0], max_points=5,
    max_voxels=20000, anchor_ranges=[0, -40.0, -1.0, 70.4, 40.0, -1.0], sizes=list(CAR_SIZE),
    rotations=[0, 1.57], matched_threshold=0.6, unmatched_threshold=0.45, out_size_factor=8)


def kitti_like_cloud(seed, n_points=20000, n_cars=8):
    """float32 [n_points, 4] (x, y, z, intensity) + gt boxes [n_cars, 7] (x,y,z,w,l,h,r)."""
    rng = np.random.default_rng(seed)
    # planted cars on the ground, z = box centre
    gt = np.zeros((n_cars, 7), dtype=np.float32)
    gt[:, 0] = rng.uniform(6, 60, n_cars)
    gt[:, 1] = rng.uniform(-25, 25, n_cars)
    gt[:, 2] = -1.73 + CAR_SIZE[2] / 2
    gt[:, 3:6] = np.array(CAR_SIZE) * rng.uniform(0.9, 1.1, (n_cars, 3))
    gt[:, 6] = rng.uniform(-np.pi, np.pi, n_cars)
    n_car_pts = int(n_points * 0.15)
    n_bg = n_points - n_car_pts
    az = rng.uniform(-np.pi / 4, np.pi / 4, n_bg)
    beams = np.deg2rad(np.linspace(-24.8, 2.0, 64))
    el = beams[rng.integers(0, 64, n_bg)]
    with np.errstate(divide="ignore"):
        r_ground = np.where(el < 0, 1.73 / np.maximum(np.tan(-el), 1e-6), np.inf)
    r_obst = 3.0 + rng.exponential(25.0, n_bg)
    r = np.minimum(np.minimum(r_ground, r_obst), 80.0)
    x = r * np.cos(el) * np.cos(az)
    y = r * np.cos(el) * np.sin(az)
    z = r * np.sin(el) + rng.normal(0, 0.02, n_bg)
    bg = np.stack([x, y, z, rng.uniform(0, 1, n_bg)], 1)
    # car surface points: sample faces of each box, rotate (clockwise for +r like the reference)
    per = max(n_car_pts // max(n_cars, 1), 1)
    cars = []
    for b in gt:
        u = rng.uniform(-0.5, 0.5, (per, 3))
        face = rng.integers(0, 3, per)
        sign = rng.choice([-0.5, 0.5], per)
        u[np.arange(per), face] = sign
        loc = u * b[3:6]
        c, s = np.cos(b[6]), np.sin(b[6])
        px = loc[:, 0] * c + loc[:, 1] * s + b[0]
        py = -loc[:, 0] * s + loc[:, 1] * c + b[1]
        cars.append(np.stack([px, py, loc[:, 2] + b[2], rng.uniform(0, 1, per)], 1))
    pts = np.concatenate([bg] + cars)[:n_points]
    if len(pts) < n_points:
        pts = np.concatenate([pts, bg[: n_points - len(pts)]])
    rng.shuffle(pts)  # Preprocess shuffles points in train mode (preprocess.py:215-217)
    return pts.astype(np.float32), gt


def create_anchors_3d_range(feature_size, anchor_range, sizes=CAR_SIZE, rotations=(0, np.pi / 2),
                            dtype=np.float32, velocities=None):
    """feature_size [D, H, W] (zyx) -> anchors [D, H, W, n_sizes, n_rots, 7], or [..., 9] =
    (x, y, z, w, l, h, vx, vy, r) when `velocities` is given (box_np_ops.py:733-805)."""
    anchor_range = np.array(anchor_range, dtype)
    stride = (anchor_range[3] - anchor_range[0]) / feature_size[2]
    zc = np.linspace(anchor_range[2], anchor_range[5], feature_size[0], dtype=dtype)
    yc = np.linspace(anchor_range[1], anchor_range[4], feature_size[1], endpoint=False, dtype=dtype) + stride / 2
    xc = np.linspace(anchor_range[0], anchor_range[3], feature_size[2], endpoint=False, dtype=dtype) + stride / 2
    rotations = np.array(rotations, dtype=dtype)
    sizes = np.reshape(np.array(sizes, dtype=dtype), [-1, 3])
    X, Y, Z, R = np.meshgrid(xc, yc, zc, rotations, indexing="ij")  # [W, H, D, nrot]
    ns = sizes.shape[0]
    parts = [np.repeat(a[:, :, :, None, :], ns, axis=3)[..., None] for a in (X, Y, Z)]
    if velocities is not None:
        sizes = np.hstack([sizes, np.array(velocities, dtype=dtype).reshape(-1, 2)])
    sz = np.broadcast_to(sizes.reshape(1, 1, 1, ns, 1, -1), parts[0].shape[:-1] + (sizes.shape[1],))
    rot = np.repeat(R[:, :, :, None, :], ns, axis=3)[..., None]
    ret = np.concatenate(parts + [sz, rot], axis=-1).astype(dtype)
    return np.transpose(ret, [2, 1, 0, 3, 4, 5])


def _limit_period(val, offset=0.5, period=np.pi):
    return val - np.floor(val / period + offset) * period


def _near_bbox(rb):
    """rbbox2d_to_near_bbox: [N,5] (x,y,w,l,r) -> axis-aligned [N,4]"""
    rots = np.abs(_limit_period(rb[:, -1], 0.5, np.pi))
    cond = (rots > np.pi / 4)[:, None]
    ctr = np.where(cond, rb[:, [0, 1, 3, 2]], rb[:, :4])
    return np.concatenate([ctr[:, :2] - ctr[:, 2:] / 2, ctr[:, :2] + ctr[:, 2:] / 2], axis=-1)


def _iou_aligned(a, b):
    """iou_jit(eps=0): [N,4] x [K,4] -> [N,K]"""
    iw = np.minimum(a[:, None, 2], b[None, :, 2]) - np.maximum(a[:, None, 0], b[None, :, 0])
    ih = np.minimum(a[:, None, 3], b[None, :, 3]) - np.maximum(a[:, None, 1], b[None, :, 1])
    inter = np.clip(iw, 0, None) * np.clip(ih, 0, None)
    ua = ((a[:, 2] - a[:, 0]) * (a[:, 3] - a[:, 1]))[:, None] + \
        ((b[:, 2] - b[:, 0]) * (b[:, 3] - b[:, 1]))[None, :] - inter
    return np.where((iw > 0) & (ih > 0), inter / ua, 0.0).astype(np.float32)


def _encode(boxes, anchors, encode_angle_vector=False):
    """second_box_encode (box_np_ops.py:52-113): 7-dim boxes or 9-dim (+ vx, vy before the angle),
    log sizes, plain velocity differences, angle difference or (cos, sin) differences."""
    nd = anchors.shape[-1]
    a = np.split(anchors, nd, axis=-1)
    g = np.split(boxes, nd, axis=-1)
    (xa, ya, za, wa, la, ha), ra = a[:6], a[-1]
    (xg, yg, zg, wg, lg, hg), rg = g[:6], g[-1]
    diag = np.sqrt(la ** 2 + wa ** 2)
    ret = [(xg - xa) / diag, (yg - ya) / diag, (zg - za) / ha, np.log(wg / wa), np.log(lg / la), np.log(hg / ha)]
    if nd == 9:
        ret += [g[6] - a[6], g[7] - a[7]]
    ret += [np.cos(rg) - np.cos(ra), np.sin(rg) - np.sin(ra)] if encode_angle_vector else [rg - ra]
    return np.concatenate(ret, axis=-1)


def assign_targets(anchors, gt_boxes, matched_threshold=0.6, unmatched_threshold=0.45, gt_classes=None,
                   encode_angle_vector=False):
    """create_target_np (target_ops.py:28-222) without subsampling: anchors [A,nd], gt [G,nd] ->
    labels [A] i32 (class / 0 neg / -1 ignore), reg_targets [A,code] f32, reg_weights [A] f32."""
    A, nd = anchors.shape
    code = nd - 1 + (2 if encode_angle_vector else 1)
    labels = np.full((A,), -1, dtype=np.int32)
    targets = np.zeros((A, code), dtype=np.float32)
    if len(gt_boxes) == 0:
        labels[:] = 0
        return labels, targets, np.zeros((A,), np.float32)
    cls = np.ones((len(gt_boxes),), np.int32) if gt_classes is None else np.asarray(gt_classes, np.int32)
    bev = [0, 1, 3, 4, nd - 1]
    ov = _iou_aligned(_near_bbox(anchors[:, bev]), _near_bbox(gt_boxes[:, bev]))
    a2g = ov.argmax(axis=1)
    a2g_max = ov[np.arange(A), a2g]
    g2a_max = ov.max(axis=0)
    g2a_max[g2a_max == 0] = -1
    force = np.where(ov == g2a_max[None, :])[0]
    labels[force] = cls[a2g[force]]
    pos = a2g_max >= matched_threshold
    labels[pos] = cls[a2g[pos]]
    bg = a2g_max < unmatched_threshold
    labels[bg] = 0
    labels[force] = cls[a2g[force]]
    fg = np.where(labels > 0)[0]
    targets[fg] = _encode(gt_boxes[a2g[fg]], anchors[fg], encode_angle_vector)
    weights = np.zeros((A,), np.float32)
    weights[fg] = 1.0
    return labels, targets, weights


def task_anchors(feature_size, generators):
    """TargetAssigner.generate_anchors (target_assigner.py:144-166) for one task group: `generators` =
    the task's anchor_generator dicts (class order).  Returns anchors [A, nd] in the concatenated layout
    (anchor a -> class (a // R) % K), K, R, matched [K], unmatched [K]."""
    per_class = []
    for g in generators:
        anc = create_anchors_3d_range(feature_size, g["anchor_ranges"], g["sizes"], g["rotations"],
                                      velocities=g.get("velocities"))
        per_class.append(anc.reshape(*anc.shape[:3], -1, anc.shape[-1]))
    r = per_class[0].shape[3]
    assert all(p.shape[3] == r for p in per_class), "classes of a task must have equally many anchors per cell"
    anchors = np.concatenate(per_class, axis=-2)
    return (np.ascontiguousarray(anchors.reshape(-1, anchors.shape[-1])), len(generators), r,
            np.array([g["matched_threshold"] for g in generators], np.float32),
            np.array([g["unmatched_threshold"] for g in generators], np.float32))


def assign_targets_task(anchors, num_classes, per_class, matched, unmatched, gt_boxes, gt_classes,
                        encode_angle_vector=False):
    """TargetAssigner.assign_v2 (target_assigner.py:68-142) on the concatenated anchor layout: class k's
    anchors against the gt boxes of class k + 1 (ids within the task), NaN gt entries read as 0."""
    A, nd = anchors.shape
    code = nd - 1 + (2 if encode_angle_vector else 1)
    cls_of = (np.arange(A) // per_class) % num_classes
    labels = np.zeros((A,), np.int32)
    targets = np.zeros((A, code), np.float32)
    weights = np.zeros((A,), np.float32)
    gt_boxes = np.nan_to_num(np.asarray(gt_boxes, np.float32).reshape(-1, nd))
    gt_classes = np.asarray(gt_classes, np.int32).reshape(-1)
    for k in range(num_classes):
        sel = np.where(cls_of == k)[0]
        m = gt_classes == k + 1
        lab, tgt, w = assign_targets(anchors[sel], gt_boxes[m], matched[k], unmatched[k], gt_classes[m],
                                     encode_angle_vector)
        labels[sel], targets[sel], weights[sel] = lab, tgt, w
    return labels, targets, weights


def assign_targets_device(anchors, gt, gt_count, matched_threshold=0.6, unmatched_threshold=0.45,
                          gt_classes=None):
    """d3d_assign_targets: anchors [A,7], gt [B,Gmax,7] (padded), gt_count [B] int32, all CUDA ->
    labels [B,A] i32, reg_targets [B,A,7] f32, reg_weights [B,A] f32 (same rule as assign_targets)."""
    from . import _lib

    if not anchors.is_cuda:
        raise _lib.D3dError("assign_targets_device needs device tensors (numpy path: assign_targets)")
    L = _lib.lib()
    anchors = anchors.float().contiguous()
    gt = gt.float().contiguous()
    gt_count = gt_count.int().contiguous()
    B, gmax = gt.shape[0], gt.shape[1]
    A = anchors.shape[0]
    dev = anchors.device
    from . import arena as mem  # (the calling thread's arena if it has one: a prefetched build's ring slot)

    labels = mem.empty((B, A), torch.int32, dev)
    targets = mem.empty((B, A, 7), torch.float32, dev)
    weights = mem.empty((B, A), torch.float32, dev)
    wsb = L.d3d_assign_targets_workspace_bytes(A, B, gmax)
    ws = mem.empty((wsb,), torch.uint8, dev)
    cls = gt_classes.int().contiguous() if gt_classes is not None else None
    rc = L.d3d_assign_targets(anchors.data_ptr(), A, gt.data_ptr(), _lib.ptr(cls), gt_count.data_ptr(), B,
                              gmax, float(matched_threshold), float(unmatched_threshold), labels.data_ptr(),
                              targets.data_ptr(), weights.data_ptr(), ws.data_ptr(), wsb, _lib.stream_ptr(dev))
    _lib.check(rc, "d3d_assign_targets")
    return labels, targets, weights


def assign_targets_task_device(anchors, num_classes, per_class, matched, unmatched, gt, gt_classes, gt_count,
                                encode_angle_vector=False, limit_gt_angle=False):
    """d3d_assign_targets_task: anchors [A,nd] (layout of task_anchors), gt [B,Gmax,nd] padded, gt_classes
    [B,Gmax] (1-based within the task), gt_count [B] -- all CUDA -> labels [B,A] i32, reg_targets
    [B,A,code] f32, reg_weights [B,A] f32.  limit_gt_angle applies AssignTarget's limit_period(r, 0.5, 2 pi)
    (preprocess.py:415-419) to the gt yaw first."""
    from . import _lib

    if not anchors.is_cuda:
        raise _lib.D3dError("assign_targets_task_device needs device tensors (numpy path: assign_targets_task)")
    L = _lib.lib()
    anchors = anchors.float().contiguous()
    gt = gt.float().contiguous()
    if limit_gt_angle:
        gt = gt.clone()
        r = gt[..., -1]
        gt[..., -1] = r - torch.floor(r / (2 * np.pi) + 0.5) * (2 * np.pi)
    gt_count = gt_count.int().contiguous()
    gt_classes = gt_classes.int().contiguous() if gt_classes is not None else None
    (A, nd), (B, gmax) = anchors.shape, gt.shape[:2]
    code = nd - 1 + (2 if encode_angle_vector else 1)
    dev = anchors.device
    from . import arena as mem

    labels = mem.empty((B, A), torch.int32, dev)
    targets = mem.empty((B, A, code), torch.float32, dev)
    weights = mem.empty((B, A), torch.float32, dev)
    wsb = L.d3d_assign_targets_workspace_bytes(A, B, gmax)
    ws = mem.empty((wsb,), torch.uint8, dev)
    rc = L.d3d_assign_targets_task(anchors.data_ptr(), A, nd, int(num_classes), int(per_class),
                                   _lib.f32_array(matched), _lib.f32_array(unmatched), int(encode_angle_vector),
                                   gt.data_ptr(), _lib.ptr(gt_classes), gt_count.data_ptr(), B, gmax,
                                   labels.data_ptr(), targets.data_ptr(), weights.data_ptr(), ws.data_ptr(), wsb,
                                   _lib.stream_ptr(dev))
    _lib.check(rc, "d3d_assign_targets_task")
    return labels, targets, weights


class _SeededFrames:
    """frames[s] = make(s), generated on first use: a rank of an N-GPU job only pays for the seeds it reads"""

    def __init__(self, n, make):
        self.n, self.make, self.cache = int(n), make, {}

    def __len__(self):
        return self.n

    def __getitem__(self, s):
        s = int(s) % self.n
        f = self.cache.get(s)
        if f is None:
            f = self.cache[s] = self.make(s)
        return f


class KittiCarSynthetic:
    """Cycles seeded frames of the KITTI-car SECOND geometry (BASELINE.json configs[1])."""

    def __init__(self, n_points=20000, n_seeds=16, cfg=None, n_cars=8):
        self.cfg = dict(KITTI_CAR if cfg is None else cfg)
        from .point_cloud_ops import grid_size_of

        self.grid = grid_size_of(self.cfg["voxel_size"], self.cfg["pc_range"])  # x, y, z
        f = self.cfg["out_size_factor"]
        self.feature_size = [1, int(self.grid[1]) // f, int(self.grid[0]) // f]
        anc = create_anchors_3d_range(self.feature_size, self.cfg["anchor_ranges"], self.cfg["sizes"],
                                      self.cfg["rotations"])
        self.anchors = anc.reshape(-1, 7)

        def make(s):
            pts, gt = kitti_like_cloud(s, n_points, n_cars)
            lab, tgt, w = assign_targets(self.anchors, gt, self.cfg["matched_threshold"],
                                         self.cfg["unmatched_threshold"])
            return dict(points=pts, gt=gt, labels=lab, reg_targets=tgt, reg_weights=w)

        self.frames = _SeededFrames(n_seeds, make)

    def host_batch(self, step, batch_size, rank=0, world=1):
        idx = [(step * batch_size * world + rank * batch_size + i) % len(self.frames) for i in range(batch_size)]
        fr = [self.frames[i] for i in idx]
        off = np.cumsum([0] + [len(f["points"]) for f in fr]).astype(np.int32)
        return dict(points=np.concatenate([f["points"] for f in fr]), frame_offsets=off,
                    anchors=np.stack([self.anchors] * batch_size),
                    labels=np.stack([f["labels"] for f in fr]),
                    reg_targets=np.stack([f["reg_targets"] for f in fr]),
                    reg_weights=np.stack([f["reg_weights"] for f in fr]),
                    gt=[f["gt"] for f in fr], metadata=[dict(token=int(i)) for i in idx])

    def device_batch(self, step, batch_size, device, rank=0, world=1):
        hb = self.host_batch(step, batch_size, rank, world)
        t = lambda a: torch.from_numpy(a).to(device, non_blocking=True)
        gmax = max(max(len(g) for g in hb["gt"]), 1)
        gt_pad = np.zeros((batch_size, gmax, 7), np.float32)
        for i, g in enumerate(hb["gt"]):
            gt_pad[i, : len(g)] = g
        gt_count = np.array([len(g) for g in hb["gt"]], np.int32)
        return dict(points=t(hb["points"]), frame_offsets=t(hb["frame_offsets"]),
                    anchors=[t(hb["anchors"])], labels=[t(hb["labels"])],
                    reg_targets=[t(hb["reg_targets"])], reg_weights=[t(hb["reg_weights"])],
                    metadata=hb["metadata"], gt=hb["gt"], gt_boxes=t(gt_pad), gt_count=t(gt_count),
                    anchors_flat=t(self.anchors))


def make_example(batch, cfg, fused_mean=True, want_voxels=False, device_targets=False):
    """raw device points -> the collate_kitti-style `example` (device voxelizer, 1 sync).
    device_targets=True recomputes labels / reg_targets from the frame's gt boxes with
    d3d_assign_targets instead of using the cached host assignment."""
    from .point_cloud_ops import grid_size_of, voxelize_batch

    vb = voxelize_batch(batch["points"], batch["frame_offsets"], cfg["voxel_size"], cfg["pc_range"],
                        cfg["max_points"], cfg["max_voxels"], want_voxels=want_voxels or not fused_mean,
                        mean_channels=batch["points"].shape[1] if fused_mean else 0)
    out = vb.sync()
    grid = grid_size_of(cfg["voxel_size"], cfg["pc_range"])
    bsz = batch["frame_offsets"].numel() - 1
    ex = dict(coordinates=out["coordinates"], num_points=out["num_points"], num_voxels=out["num_voxels"],
              shape=np.stack([grid] * bsz), anchors=batch["anchors"], metadata=batch.get("metadata", []))
    if device_targets:
        lab, tgt, w = assign_targets_device(batch["anchors_flat"], batch["gt_boxes"], batch["gt_count"],
                                            cfg["matched_threshold"], cfg["unmatched_threshold"])
        ex["labels"], ex["reg_targets"], ex["reg_weights"] = [lab], [tgt], [w]
    else:
        for k in ("labels", "reg_targets", "reg_weights"):
            if k in batch:
                ex[k] = batch[k]
    if "voxels" in out:
        ex["voxels"] = out["voxels"]
    if fused_mean:
        ex["voxel_mean"] = out["mean"]
    return ex


# ------------------------------------------------------------------ multi-group workloads (BASELINE configs[2..4])
def spinning_lidar_cloud(seed, n_points, pc_range, channels=5, beams=32, sweeps=10, sensor_h=1.84,
                         boxes=None):
    """360-degree spinning lidar: `beams` elevations between -30.7 and +10.7 degrees,
    ground-plane hits at sensor height, exponential obstacles, clipped to the range; `sweeps` > 1 adds the
    time-lag channel (0, 0.05, ... as in nuScenes' 10-sweep clouds, pipelines/loading.py:126).  Surface points
    of the planted `boxes` [G, >=7] (x, y, z, w, l, h, ..., r) take 15 % of the budget.
    Returns float32 [n_points, channels] (x, y, z, intensity[, dt])."""
    rng = np.random.default_rng(seed)
    lo, hi = np.array(pc_range[:3], np.float64), np.array(pc_range[3:], np.float64)
    n_box = int(n_points * 0.15) if boxes is not None and len(boxes) else 0
    n_bg = n_points - n_box
    az = rng.uniform(-np.pi, np.pi, n_bg)
    el = np.deg2rad(np.linspace(-30.7, 10.7, beams))[rng.integers(0, beams, n_bg)]
    with np.errstate(divide="ignore"):
        r_ground = np.where(el < 0, sensor_h / np.maximum(np.tan(-el), 1e-6), np.inf)
    r = np.minimum(np.minimum(r_ground, 2.0 + rng.exponential(0.35 * hi[0], n_bg)), 1.4 * hi[0])
    xyz = np.stack([r * np.cos(el) * np.cos(az), r * np.cos(el) * np.sin(az),
                    r * np.sin(el) + rng.normal(0, 0.02, n_bg)], 1)
    parts = [xyz]
    if n_box:
        per = max(n_box // len(boxes), 1)
        for b in boxes:
            u = rng.uniform(-0.5, 0.5, (per, 3))
            u[np.arange(per), rng.integers(0, 3, per)] = rng.choice([-0.5, 0.5], per)
            loc = u * b[3:6]
            c, s = np.cos(b[-1]), np.sin(b[-1])
            parts.append(np.stack([loc[:, 0] * c + loc[:, 1] * s + b[0], -loc[:, 0] * s + loc[:, 1] * c + b[1],
                                   loc[:, 2] + b[2]], 1))
    xyz = np.concatenate(parts)[:n_points]
    if len(xyz) < n_points:
        xyz = np.concatenate([xyz, parts[0][: n_points - len(xyz)]])
    xyz = np.clip(xyz, lo + 1e-3, hi - 1e-3)
    cols = [xyz, rng.uniform(0, 1, (n_points, 1))]
    if channels >= 5:
        cols.append(rng.integers(0, sweeps, (n_points, 1)) * 0.05)
    pts = np.concatenate(cols, 1).astype(np.float32)[:, :channels]
    rng.shuffle(pts)
    return pts


class TaskGroupSynthetic:
    """Seeded frames for a multi-group config (nuScenes CBGS / nuScenes PointPillars / Lyft CBGS): clouds from
    spinning_lidar_cloud with planted boxes of every class, per-task anchors from the config's anchor
    generators (target_assigner.py:144-166), padded per-task gt tensors for d3d_assign_targets_task."""

    def __init__(self, cfg, n_points, n_seeds=8, channels=None, boxes_per_class=3, beams=32, sweeps=10):
        from .point_cloud_ops import grid_size_of

        self.cfg = cfg
        vg = cfg.voxel_generator
        self.vox = dict(voxel_size=list(vg.voxel_size), pc_range=list(vg.range), max_points=vg.max_points_in_voxel,
                        max_voxels=vg.max_voxel_num)
        self.grid = grid_size_of(vg.voxel_size, vg.range)
        f = cfg.assigner.out_size_factor
        self.feature_size = [1, int(self.grid[1]) // f, int(self.grid[0]) // f]
        self.vec = bool(cfg.box_coder.encode_angle_vector)
        self.nd = int(cfg.box_coder.n_dim)
        self.channels = channels or (5 if self.nd == 9 else 4)
        gens_all = {g.class_name: g for g in cfg.target_assigner.anchor_generators}
        self.tasks = []
        for task in cfg.tasks:
            gens = [gens_all[c].to_dict() if hasattr(gens_all[c], "to_dict") else dict(gens_all[c])
                    for c in task.class_names]
            anchors, k, r, hi_t, lo_t = task_anchors(self.feature_size, gens)
            self.tasks.append(dict(gens=gens, anchors=anchors, k=k, r=r, matched=hi_t, unmatched=lo_t))
        lo, hi = np.array(vg.range[:3]), np.array(vg.range[3:])

        def make(s):
            rng = np.random.default_rng(1000 + s)
            per_task, all_boxes = [], []
            for t in self.tasks:
                g = np.zeros((t["k"] * boxes_per_class, self.nd), np.float32)
                c = np.zeros((t["k"] * boxes_per_class,), np.int32)
                for j in range(len(g)):
                    ci = j % t["k"]
                    gen = t["gens"][ci]
                    c[j] = ci + 1
                    g[j, :2] = rng.uniform(0.8 * lo[:2], 0.8 * hi[:2])
                    g[j, 2] = gen["anchor_ranges"][2]
                    g[j, 3:6] = np.array(gen["sizes"]) * rng.uniform(0.9, 1.1, 3)
                    if self.nd == 9:
                        g[j, 6:8] = rng.normal(0, 2, 2)
                    g[j, -1] = rng.uniform(-np.pi, np.pi)
                per_task.append((g, c))
                all_boxes.append(g)
            pts = spinning_lidar_cloud(s, n_points, vg.range, self.channels, beams, sweeps,
                                       boxes=np.concatenate(all_boxes))
            return dict(points=pts, gt=per_task)

        self.frames = _SeededFrames(n_seeds, make)

    def device_batch(self, step, batch_size, device, rank=0, world=1):
        idx = [(step * batch_size * world + rank * batch_size + i) % len(self.frames) for i in range(batch_size)]
        fr = [self.frames[i] for i in idx]
        t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(device, non_blocking=True)
        off = np.cumsum([0] + [len(f["points"]) for f in fr]).astype(np.int32)
        out = dict(points=t(np.concatenate([f["points"] for f in fr])), frame_offsets=t(off),
                   metadata=[dict(token=int(i)) for i in idx], anchors=[], anchors_flat=[], gt_boxes=[],
                   gt_classes=[], gt_count=[])
        for ti, task in enumerate(self.tasks):
            gmax = max(len(f["gt"][ti][0]) for f in fr)
            gt = np.zeros((batch_size, gmax, self.nd), np.float32)
            cls = np.zeros((batch_size, gmax), np.int32)
            for b, f in enumerate(fr):
                g, c = f["gt"][ti]
                gt[b, : len(g)], cls[b, : len(g)] = g, c
            a = t(task["anchors"])
            out["anchors_flat"].append(a)
            out["anchors"].append(a[None].expand(batch_size, -1, -1).contiguous())
            out["gt_boxes"].append(t(gt))
            out["gt_classes"].append(t(cls))
            out["gt_count"].append(t(np.array([len(f["gt"][ti][0]) for f in fr], np.int32)))
        return out

    def make_example(self, batch, want_voxels=False, targets=True):
        """device voxelizer (+ fused mean for VoxelNet readers) and one d3d_assign_targets_task call per task
        group -> the collate_kitti-style `example`."""
        from .point_cloud_ops import voxelize_batch

        v = self.vox
        pillars = self.cfg.model.reader.type == "PillarFeatureNet"
        mean_c = 0 if pillars else int(self.cfg.model.reader.num_input_features)
        vb = voxelize_batch(batch["points"], batch["frame_offsets"], v["voxel_size"], v["pc_range"],
                            v["max_points"], v["max_voxels"], want_voxels=want_voxels or pillars,
                            mean_channels=mean_c)
        out = vb.sync()
        bsz = batch["frame_offsets"].numel() - 1
        ex = dict(coordinates=out["coordinates"], num_points=out["num_points"], num_voxels=out["num_voxels"],
                  shape=np.stack([self.grid] * bsz), anchors=batch["anchors"], metadata=batch["metadata"])
        if "voxels" in out:
            ex["voxels"] = out["voxels"]
        if mean_c:
            ex["voxel_mean"] = out["mean"]
        if targets:
            ex["labels"], ex["reg_targets"], ex["reg_weights"] = [], [], []
            for ti, task in enumerate(self.tasks):
                lab, tgt, w = assign_targets_task_device(
                    batch["anchors_flat"][ti], task["k"], task["r"], task["matched"], task["unmatched"],
                    batch["gt_boxes"][ti], batch["gt_classes"][ti], batch["gt_count"][ti],
                    encode_angle_vector=self.vec, limit_gt_angle=True)
                ex["labels"].append(lab)
                ex["reg_targets"].append(tgt)
                ex["reg_weights"].append(w)
        return ex
