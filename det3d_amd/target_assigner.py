"""`AssignTarget` of the reference pipeline (det3d/datasets/pipelines/preprocess.py:306-483) on top of the device
kernel: same constructor (`AssignTarget(cfg=train_cfg.assigner)`), same per-task outputs (`anchors`, `labels`,
`reg_targets`, `reg_weights`), but working on the collated batch that is already on the GPU instead of frame by
frame in DataLoader workers.

  per task group t (target_assigner.tasks[t], its classes are consecutive in `anchor_generators`):
    anchors_t   TargetAssigner.generate_anchors (target_assigner.py:144-166) -- cached per feature-map size
    gt of task  gt_classes == class index within all classes, re-based to 1..K_t (preprocess.py:383-413)
    yaw         limit_period(r, 0.5, 2 pi)                                   (preprocess.py:415-419)
    targets     TargetAssigner.assign_v2 per class                           (target_assigner.py:68-142)

`__call__(res, info)` keeps the reference's single-frame NumPy interface (numpy restatement in synthetic.py, pinned
to the reference's own AssignTarget by tests/golden/assign_pipeline.npz); `assign_batch` is the device path
(d3d_assign_targets_task, one launch pair per task group for the whole batch).
"""
import numpy as np
import torch

from . import synthetic


def _get(cfg, key):
    return cfg[key] if isinstance(cfg, dict) else getattr(cfg, key)


class AssignTarget:
    def __init__(self, **kwargs):
        cfg = kwargs["cfg"]
        ta = _get(cfg, "target_assigner")
        self.tasks = [dict(num_class=_get(t, "num_class"), class_names=list(_get(t, "class_names")))
                      for t in _get(ta, "tasks")]
        gens = [dict(g) if isinstance(g, dict) else g.to_dict() for g in _get(ta, "anchor_generators")]
        for g in gens:
            if g.get("type", "anchor_generator_range") != "anchor_generator_range":
                raise NotImplementedError("AssignTarget: only anchor_generator_range is used by the shipped configs")
        self.class_names = [g["class_name"] for g in gens]
        coder = _get(cfg, "box_coder")
        self.encode_angle_vector = bool(_get(coder, "encode_angle_vector"))
        self.n_dim = int(_get(coder, "n_dim"))
        if _get(ta, "sample_positive_fraction") >= 0:
            raise NotImplementedError("AssignTarget: positive/negative subsampling is off in every shipped config")
        if _get(ta, "pos_area_threshold") >= 0:
            raise NotImplementedError("AssignTarget: anchors_mask (pos_area_threshold >= 0) is not supported")
        self.out_size_factor = _get(cfg, "out_size_factor")
        self._gens_by_task, flag = [], 0
        for t in self.tasks:
            self._gens_by_task.append(gens[flag: flag + t["num_class"]])
            flag += t["num_class"]
        self._anchor_cache = {}

    # ------------------------------------------------------------------ anchors
    def feature_map_size(self, grid_size):
        fm = np.asarray(grid_size)[:2] // self.out_size_factor
        return [1, int(fm[1]), int(fm[0])]  # [*fm, 1][::-1]

    def task_anchors(self, grid_size):
        """list over tasks of (anchors [A_t, nd] f32, K_t, R_t, matched [K_t], unmatched [K_t])"""
        key = tuple(self.feature_map_size(grid_size))
        if key not in self._anchor_cache:
            self._anchor_cache[key] = [synthetic.task_anchors(list(key), g) for g in self._gens_by_task]
        return self._anchor_cache[key]

    def split_by_task(self, gt_boxes, gt_classes):
        """global 1-based class ids -> per task (boxes, ids within the task), class-major like the reference"""
        out, flag = [], 0
        gt_boxes = np.asarray(gt_boxes, np.float32).reshape(-1, self.n_dim)
        gt_classes = np.asarray(gt_classes).reshape(-1)
        for t in self.tasks:
            boxes, cls = [], []
            for i in range(t["num_class"]):
                m = np.where(gt_classes == i + 1 + flag)[0]
                boxes.append(gt_boxes[m])
                cls.append(gt_classes[m] - flag)
            b = np.concatenate(boxes, axis=0).astype(np.float32)
            b[:, -1] = b[:, -1] - np.floor(b[:, -1] / (2 * np.pi) + 0.5) * (2 * np.pi)  # limit_period(., 0.5, 2 pi)
            out.append((b, np.concatenate(cls).astype(np.int32)))
            flag += t["num_class"]
        return out

    # ------------------------------------------------------------------ reference interface (one frame, NumPy)
    def __call__(self, res, info):
        grid_size = res["lidar"]["voxels"]["shape"]
        per_task = self.task_anchors(grid_size)
        example = {"anchors": [a for a, _, _, _, _ in per_task]}
        if res["mode"] == "train":
            gt = res["lidar"]["annotations"]
            split = self.split_by_task(gt["gt_boxes"], gt["gt_classes"])
            names = []
            for t, (_, cls) in zip(self.tasks, split):
                names.append(np.array([t["class_names"][c - 1] for c in cls]))
            gt["gt_boxes"] = [b for b, _ in split]
            gt["gt_classes"] = [c for _, c in split]
            gt["gt_names"] = names
            labels, targets, weights = [], [], []
            for (anc, k, r, hi, lo), (boxes, cls) in zip(per_task, split):
                lab, tgt, w = synthetic.assign_targets_task(anc, k, r, hi, lo, boxes, cls, self.encode_angle_vector)
                labels.append(lab)
                targets.append(tgt)
                weights.append(w)
            example.update(labels=labels, reg_targets=targets, reg_weights=weights)
        res["lidar"]["targets"] = example
        return res, info

    # ------------------------------------------------------------------ device path (whole batch)
    def assign_batch(self, grid_size, gt_boxes, gt_classes, gt_count):
        """gt_boxes [B, G, nd] padded, gt_classes [B, G] global 1-based ids, gt_count [B]: CUDA tensors.
        Returns dict(anchors, labels, reg_targets, reg_weights), each a list over tasks of [B, ...] tensors."""
        dev = gt_boxes.device
        per_task = self.task_anchors(grid_size)
        b, g = gt_classes.shape
        valid = torch.arange(g, device=dev)[None, :] < gt_count[:, None].to(dev)
        out = dict(anchors=[], labels=[], reg_targets=[], reg_weights=[])
        flag = 0
        for t, (anc, k, r, hi, lo) in zip(self.tasks, per_task):
            key = ("dev", id(anc), str(dev))
            if key not in self._anchor_cache:
                self._anchor_cache[key] = torch.from_numpy(anc).to(dev)
            a = self._anchor_cache[key]
            # boxes of other tasks are masked out by class id 0 (no class of the kernel matches it)
            cls = torch.where(valid & (gt_classes > flag) & (gt_classes <= flag + k), gt_classes - flag,
                              torch.zeros_like(gt_classes)).int()
            if k == 1:  # the single-class kernel path takes every listed gt: compact the task's boxes to the front
                order = torch.argsort((cls == 0).int(), dim=1, stable=True)
                boxes = torch.gather(gt_boxes, 1, order[..., None].expand(-1, -1, gt_boxes.shape[-1]))
                cls_t = torch.gather(cls, 1, order)
                cnt = (cls_t > 0).sum(1).int()
            else:
                boxes, cls_t, cnt = gt_boxes, cls, gt_count.to(dev).int()
            lab, tgt, w = synthetic.assign_targets_task_device(a, k, r, hi, lo, boxes, cls_t, cnt,
                                                               encode_angle_vector=self.encode_angle_vector,
                                                               limit_gt_angle=True)
            out["anchors"].append(a[None].expand(b, -1, -1))
            out["labels"].append(lab)
            out["reg_targets"].append(tgt)
            out["reg_weights"].append(w)
            flag += k
        return out
