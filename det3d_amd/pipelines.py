"""Pipeline steps of the reference on either side of the hot path, with the reference's `(res, info)` interface
(det3d/datasets/pipelines/preprocess.py), for callers that keep the reference's dataset code:

  Voxelization   :258-304  range filter of the gt boxes + VoxelGenerator.generate -> res["lidar"]["voxels"];
                           the voxelizer is the HIP kernel (point_cloud_ops.VoxelGenerator), so this step needs the GPU
  AssignTarget   :306-483  see target_assigner.py

In the MI355X training loop these steps run on the collated batch instead (point_cloud_ops.voxelize_batch,
AssignTarget.assign_batch); the per-frame classes exist so that an unmodified reference pipeline list
(`train_pipeline = [..., dict(type="Voxelization", cfg=voxel_generator), dict(type="AssignTarget", ...)]`) builds.
"""
import numpy as np

from .point_cloud_ops import VoxelGenerator
from .target_assigner import AssignTarget  # noqa: F401  (re-exported under the reference's name)


def box_bev_corners(gt_boxes):
    """center_to_corner_box2d(xy, (w, l), r) of box_np_ops.py:477-497 -> [N, 4, 2]; corners (-,-) (-,+) (+,+) (+,-)
    times the dims, rotated clockwise for positive r (rotation_2d :419-432)."""
    gt_boxes = np.asarray(gt_boxes)
    dims = gt_boxes[:, [3, 4]]
    norm = np.array([[-0.5, -0.5], [-0.5, 0.5], [0.5, 0.5], [0.5, -0.5]], dtype=dims.dtype)
    corners = dims[:, None, :] * norm[None]
    s, c = np.sin(gt_boxes[:, -1]), np.cos(gt_boxes[:, -1])
    x = corners[..., 0] * c[:, None] + corners[..., 1] * s[:, None]
    y = -corners[..., 0] * s[:, None] + corners[..., 1] * c[:, None]
    return np.stack([x, y], -1) + gt_boxes[:, None, :2]


def filter_gt_box_outside_range(gt_boxes, limit_range):
    """core/sampler/preprocess.py:133-147: keep a box when at least one BEV corner lies strictly inside
    (x0, y0, x1, y1) -- the convex-polygon test there rejects points on the border (cross >= 0)."""
    x0, y0, x1, y1 = (float(v) for v in np.asarray(limit_range).reshape(-1)[:4])
    c = box_bev_corners(gt_boxes)
    inside = (c[..., 0] > x0) & (c[..., 0] < x1) & (c[..., 1] > y0) & (c[..., 1] < y1)
    return inside.any(axis=1)


def _dict_select(d, mask):
    for k, v in d.items():
        d[k] = _dict_select(v, mask) if isinstance(v, dict) else v[mask]
    return d


class Voxelization:
    def __init__(self, **kwargs):
        cfg = kwargs.get("cfg", None)
        get = (lambda k: cfg[k]) if isinstance(cfg, dict) else (lambda k: getattr(cfg, k))
        self.range = get("range")
        self.voxel_size = get("voxel_size")
        self.max_points_in_voxel = get("max_points_in_voxel")
        self.max_voxel_num = get("max_voxel_num")
        self.voxel_generator = VoxelGenerator(voxel_size=self.voxel_size, point_cloud_range=self.range,
                                              max_num_points=self.max_points_in_voxel, max_voxels=self.max_voxel_num)

    def __call__(self, res, info):
        pc_range = self.voxel_generator.point_cloud_range
        grid_size = self.voxel_generator.grid_size
        if res["mode"] == "train":
            gt_dict = res["lidar"]["annotations"]
            mask = filter_gt_box_outside_range(gt_dict["gt_boxes"], pc_range[[0, 1, 3, 4]])
            _dict_select(gt_dict, mask)
            res["lidar"]["annotations"] = gt_dict
        voxels, coordinates, num_points = self.voxel_generator.generate(res["lidar"]["points"])
        res["lidar"]["voxels"] = dict(voxels=voxels, coordinates=coordinates, num_points=num_points,
                                      num_voxels=np.array([voxels.shape[0]], dtype=np.int64), shape=grid_size)
        return res, info
