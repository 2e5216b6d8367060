"""Batch layout of the path (seam B3b): `collate_kitti`
(det3d/torchie/parallel/collate.py:90-150) and `example_to_device` (det3d/torchie/trainer/trainer.py:34-57),
same keys, dtypes and shapes, plus the adapter that feeds a reference-style collated batch to the device
voxelizer (the reference uploads `example["points"]` and never uses it; here it is the voxelizer's input).
"""
import collections

import numpy as np
import torch

_CONCAT = ("voxels", "num_points", "num_gt", "voxel_labels", "num_voxels")
_PER_TASK = ("anchors", "anchors_mask", "reg_targets", "reg_weights", "labels")
_TO_DEVICE = ("voxels", "bev_map", "coordinates", "num_points", "points", "num_voxels")


def collate_kitti(batch_list, samples_per_gpu=1):
    merged = collections.defaultdict(list)
    for example in batch_list:
        for k, v in example.items():
            merged[k].append(v)
    batch_size = len(batch_list)
    ret = {}
    for key, elems in merged.items():
        if key in _CONCAT:
            ret[key] = torch.tensor(np.concatenate(elems, axis=0))
        elif key == "gt_boxes":  # per task: zero-padded [B, max_gt, 7] float64 arrays
            res = []
            for task_id in range(len(elems[0])):
                max_gt = max(len(elems[k][task_id]) for k in range(batch_size))
                padded = np.zeros((batch_size, max_gt, 7))
                for i in range(batch_size):
                    padded[i, : len(elems[i][task_id]), :] = elems[i][task_id]
                res.append(padded)
            ret[key] = res
        elif key == "metadata":
            ret[key] = elems
        elif key == "calib":
            calib = {}
            for elem in elems:
                for k1, v1 in elem.items():
                    calib.setdefault(k1, []).append(v1)
            ret[key] = {k1: torch.tensor(np.stack(v1, axis=0)) for k1, v1 in calib.items()}
        elif key in ("coordinates", "points"):  # prepend the frame index column
            ret[key] = torch.tensor(np.concatenate(
                [np.pad(c, ((0, 0), (1, 0)), mode="constant", constant_values=i) for i, c in enumerate(elems)], axis=0))
        elif key in _PER_TASK:  # list over tasks of [B, ...] tensors
            per_task = collections.defaultdict(list)
            for elem in elems:
                for idx, ele in enumerate(elem):
                    per_task[idx].append(torch.tensor(ele))
            ret[key] = [torch.stack(per_task[idx]) for idx in sorted(per_task)]
        else:
            ret[key] = np.stack(elems, axis=0)
    return ret


def example_to_device(example, device, non_blocking=False):
    out = {}
    for k, v in example.items():
        if k in _PER_TASK:
            out[k] = [t.to(device, non_blocking=non_blocking) for t in v]
        elif k in _TO_DEVICE:
            out[k] = v.to(device, non_blocking=non_blocking)
        elif k == "calib":
            out[k] = {k1: v1.to(device, non_blocking=non_blocking) for k1, v1 in v.items()}
        else:
            out[k] = v
    return out


def points_for_device_voxelizer(example):
    """collated `example["points"]` [sum N, 1 + C] (frame index first, frames in order) ->
    (points [sum N, C] float32, frame_offsets [B + 1] int32) for point_cloud_ops.voxelize_batch."""
    pts = example["points"]
    batch = len(example["metadata"]) if "metadata" in example else int(pts[:, 0].max().item()) + 1
    counts = torch.bincount(pts[:, 0].long(), minlength=batch)
    offsets = torch.zeros(batch + 1, dtype=torch.int32, device=pts.device)
    offsets[1:] = torch.cumsum(counts, 0).int()
    return pts[:, 1:].float().contiguous(), offsets
