"""Ground-truth database sampling ("GT-AUG") with the object point clouds resident in HBM.

Reference: det3d/core/sampler/sample_ops.py:13-298 (DataBaseSamplerV2), BatchSampler and the database filters in
det3d/core/sampler/preprocess.py:19-110, built by det3d/builder.py:367-396, used by the train pipeline at
det3d/datasets/pipelines/preprocess.py:152-181.  Per training frame it draws up to `max - present` objects of
every class from a database of cropped ground-truth objects, drops the ones whose bird's-eye rectangles collide with
the frame's boxes or with an earlier sample, reads each survivor's points from its own small file, moves them to
the box centre, removes the scene points inside the pasted boxes and puts the pasted points in front of the cloud.

Here the database is loaded ONCE: all object clouds packed into one `[sum P, C]` tensor on the device with an offset
table (KITTI's car database is ~50 MB, nuScenes' a few GB -- nothing next to 288 GB of HBM), so pasting is a gather
instead of 15 file reads per frame.  The selection logic (which objects, in which order, which survive) is the
reference's, statement by statement, driven by the same NumPy random stream: a seeded run picks the same objects
(tests/golden/gt_sampler.npz, made by running the reference class).  Group sampling, per-object global rotation
(`global_random_rotation_range_per_object`, [0, 0] in every shipped config) and KITTI's frustum `random_crop` are
not on the hot path and not built.
"""
import copy

import numpy as np
import torch

from .augment import _bev_corners, _box_collision


class BatchSampler:
    """preprocess.py:19-54: a shuffled index list handed out in slices, reshuffled when it runs out"""

    def __init__(self, n, rng=np.random):
        self._rng = rng
        self._indices = np.arange(n)
        rng.shuffle(self._indices)
        self._idx = 0
        self._n = n

    def sample(self, num):
        if self._idx + num >= self._n:
            ret = self._indices[self._idx:].copy()
            self._rng.shuffle(self._indices)
            self._idx = 0
        else:
            ret = self._indices[self._idx: self._idx + num].copy()  # (the reference returns a view; its callers use it at once)
            self._idx += num
        return ret


def filter_db_infos(db_infos, db_prep_steps):
    """det3d/builder.py:67-80 + preprocess.py:66-98: `filter_by_min_num_points` / `filter_by_difficulty` steps"""
    for step in db_prep_steps or []:
        if "filter_by_difficulty" in step:
            removed = step["filter_by_difficulty"]
            db_infos = {k: [i for i in v if i["difficulty"] not in removed] for k, v in db_infos.items()}
        elif "filter_by_min_num_points" in step:
            for name, min_num in step["filter_by_min_num_points"].items():
                if min_num > 0:
                    db_infos[name] = [i for i in db_infos[name] if i["num_points_in_gt"] >= min_num]
        else:
            raise ValueError("unknown database preprocessing step: %r" % (step,))
    return db_infos


class DeviceGTSampler:
    def __init__(self, db_infos, sample_groups, db_prep_steps=None, rate=1.0, device="cuda", root_path=None,
                 num_point_features=4, rng=np.random):
        """db_infos: {class name: [info, ...]} as in the reference's dbinfos pickle; an info carries `box3d_lidar`,
        `name`, `difficulty`, `num_points_in_gt` and either `points` (array [P, C]) or `path` (float32 file under
        root_path, as written by create_groundtruth_database).  sample_groups: [{class: max objects per frame}, ...]."""
        if any(len(g) > 1 for g in sample_groups):
            raise NotImplementedError("group sampling (several classes per group) is not on the Det3D hot path")
        db_infos = filter_db_infos({k: list(v) for k, v in db_infos.items()}, db_prep_steps)
        self.db_infos = db_infos
        self._rate = rate
        self._rng = rng
        self._sample_classes, self._sample_max_nums = [], []
        for g in sample_groups:
            self._sample_classes += list(g.keys())
            self._sample_max_nums += list(g.values())
        # ---- the database, packed: boxes [n, 7+], points [sum P, C], offsets [n + 1]; class k owns a slice of it
        self.device = torch.device(device)
        self._base, boxes, clouds, off = {}, [], [], [0]
        for name, infos in db_infos.items():
            self._base[name] = len(boxes)
            for info in infos:
                if "points" in info:
                    p = np.asarray(info["points"], np.float32)
                else:
                    import pathlib

                    p = np.fromfile(str(pathlib.Path(root_path) / info["path"]), dtype=np.float32)
                    p = p.reshape(-1, num_point_features)
                boxes.append(np.asarray(info["box3d_lidar"], np.float32))
                clouds.append(p)
                off.append(off[-1] + len(p))
        self.n_objects = len(boxes)
        width = clouds[0].shape[1] if clouds else num_point_features
        self.boxes_np = np.stack(boxes) if boxes else np.zeros((0, 7), np.float32)
        self.points = torch.from_numpy(np.concatenate(clouds) if clouds else np.zeros((0, width), np.float32)).to(self.device)
        self.offsets_np = np.asarray(off, np.int64)
        # BatchSamplers in the reference's construction order (dict order of the filtered infos): each one shuffles
        self._samplers = {k: BatchSampler(len(v), rng) for k, v in db_infos.items()}

    # ------------------------------------------------------------------ selection (host, a few dozen boxes)
    def _sample_class(self, name, num, avoid_boxes):
        """sample_class_v2 (:253-298) without the per-object global rotation"""
        idx = self._samplers[name].sample(num)
        infos = [self.db_infos[name][i] for i in idx]
        num_gt, num_sampled = avoid_boxes.shape[0], len(infos)
        if num_sampled == 0:
            return [], []
        sp_boxes = np.stack([i["box3d_lidar"] for i in infos], axis=0).astype(avoid_boxes.dtype)
        boxes = torch.from_numpy(np.concatenate([avoid_boxes, sp_boxes], axis=0))
        bv = _bev_corners(boxes[:, 0:2], boxes[:, 3:5], boxes[:, -1])
        coll = _box_collision(bv[:, None], bv[None, :]).numpy()
        np.fill_diagonal(coll, False)
        keep = []
        for i in range(num_gt, num_gt + num_sampled):
            if coll[i].any():
                coll[i] = False
                coll[:, i] = False
            else:
                keep.append(i - num_gt)
        return [copy.deepcopy(infos[k]) for k in keep], [self._base[name] + int(idx[k]) for k in keep]

    def sample_all(self, gt_boxes, gt_names):
        """One frame.  gt_boxes: numpy [N, 7+]; gt_names: sequence of N class names.  Returns None or a dict with the
        reference's keys (:213-231); `points` is a device tensor gathered from the resident database, `db_index` the
        positions of the pasted objects in it."""
        gt_boxes = np.asarray(gt_boxes)
        nums = []
        for cls, max_num in zip(self._sample_classes, self._sample_max_nums):
            n = int(max_num - np.sum([x == cls for x in gt_names]))
            nums.append(np.round(self._rate * n).astype(np.int64))
        sampled, rows, sampled_boxes = [], [], []
        avoid = gt_boxes
        for cls, n in zip(self._sample_classes, nums):
            if n > 0:
                infos, idx = self._sample_class(cls, n, avoid)
                sampled += infos
                rows += idx
                if infos:
                    box = np.stack([s["box3d_lidar"] for s in infos], axis=0)
                    sampled_boxes.append(box)
                    avoid = np.concatenate([avoid, box], axis=0)
        if not sampled:
            return None
        sampled_boxes = np.concatenate(sampled_boxes, axis=0)
        return {"gt_names": np.array([s["name"] for s in sampled]),
                "difficulty": np.array([s["difficulty"] for s in sampled]),
                "gt_boxes": sampled_boxes,
                "points": self.gather_points(rows, sampled_boxes),
                "gt_masks": np.ones((len(sampled),), dtype=np.bool_),
                "group_ids": np.arange(gt_boxes.shape[0], gt_boxes.shape[0] + len(sampled)),
                "db_index": np.asarray(rows, np.int64)}

    # ------------------------------------------------------------------ pasting (device)
    def gather_points(self, rows, boxes):
        """the clouds of database objects `rows`, each moved to its box centre (s_points[:, :3] += box[:3], :188)"""
        lo, hi = self.offsets_np[rows], self.offsets_np[np.asarray(rows) + 1]
        counts = hi - lo
        total = int(counts.sum())
        if total == 0:
            return self.points.new_zeros((0, self.points.shape[1]))
        obj = np.repeat(np.arange(len(rows)), counts)
        src = np.concatenate([np.arange(a, b) for a, b in zip(lo, hi)])
        pts = self.points[torch.from_numpy(src).to(self.device)].clone()
        shift = torch.from_numpy(np.asarray(boxes, np.float32)[:, :3][obj]).to(self.device)
        pts[:, :3] += shift
        return pts

    def paste(self, points, gt_boxes, gt_names, sampled, remove_points_after_sample=True):
        """pipelines/preprocess.py:163-181 for one frame: append boxes / names, drop the scene points inside the pasted
        boxes (box_np_ops.points_in_rbbox), put the pasted points in front.  points: device tensor [N, C]."""
        if sampled is None:
            return points, gt_boxes, np.asarray(gt_names), np.ones((len(gt_names),), np.bool_)
        names = np.concatenate([np.asarray(gt_names), sampled["gt_names"]], axis=0)
        boxes = np.concatenate([np.asarray(gt_boxes), sampled["gt_boxes"]])
        mask = np.concatenate([np.ones((len(gt_names),), np.bool_), sampled["gt_masks"]], axis=0)
        if remove_points_after_sample and points.shape[0]:
            inside = points_in_rbbox(points, torch.from_numpy(sampled["gt_boxes"].astype(np.float32)).to(points.device))
            points = points[~inside.any(-1)]
        return torch.cat([sampled["points"].to(points.dtype), points], dim=0), boxes, names, mask


def points_in_rbbox(points, boxes):
    """box_np_ops.points_in_rbbox (centre origin, rotation about z by the LAST box column):
    [N, 3+] x [M, 7+] -> bool [N, M]"""
    d = points[:, None, :3].float() - boxes[None, :, :3]
    c, s = torch.cos(boxes[:, -1])[None], torch.sin(boxes[:, -1])[None]
    lx = d[..., 0] * c - d[..., 1] * s
    ly = d[..., 0] * s + d[..., 1] * c
    half = boxes[None, :, 3:6] * 0.5
    return (lx.abs() < half[..., 0]) & (ly.abs() < half[..., 1]) & (d[..., 2].abs() < half[..., 2])
