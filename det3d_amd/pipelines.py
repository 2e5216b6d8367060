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

from .loading import LoadPointCloudFromFile
from .point_cloud_ops import VoxelGenerator
from .registry import PIPELINES, build_from_cfg
from .target_assigner import AssignTarget  # (re-exported under the reference's name)

PIPELINES.register_module(LoadPointCloudFromFile)
PIPELINES.register_module(AssignTarget)


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


@PIPELINES.register_module
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


def build_dbsampler(cfg, device="cuda", root_path=None, num_point_features=4):
    """det3d/builder.py:378-396: the database infos pickle + groups + filters -> a sampler; here the one that keeps
    the object clouds resident on the device (gt_sampler.DeviceGTSampler)."""
    import pickle

    from .gt_sampler import DeviceGTSampler

    get = (lambda k, d=None: cfg.get(k, d)) if hasattr(cfg, "get") else (lambda k, d=None: getattr(cfg, k, d))
    grot = list(get("global_random_rotation_range_per_object", []) or [])
    if len(grot) == 2 and abs(grot[0] - grot[1]) >= 1e-3:
        raise NotImplementedError("per-object global rotation of sampled boxes is not on the Det3D hot path")
    with open(get("db_info_path"), "rb") as f:
        db_infos = pickle.load(f)
    return DeviceGTSampler(db_infos, [dict(g) for g in get("sample_groups")], [dict(s) for s in get("db_prep_steps")],
                           rate=get("rate", 1.0), device=device, root_path=root_path,
                           num_point_features=num_point_features)


@PIPELINES.register_module
class Preprocess:
    """The train / val preprocessing step (det3d/datasets/pipelines/preprocess.py:29-255) for ONE frame with the
    points on the device: class filtering, GT-database sampling + pasting, per-object noise, global flip / rotation /
    scaling, shuffle.  Every random draw is made with the reference's NumPy call at the reference's place in the
    stream, so a seeded run yields the same boxes, the same points and the same point ORDER (the voxelizer is order
    sensitive); the geometry runs on the device (augment.py, gt_sampler.py).  Branches of the reference that cannot
    run there either (they use undefined names: remove_environment, min_points_in_gt) or need camera data
    (add_rgb_to_points, reference_detections, remove_outside_points, random_crop) are rejected at construction."""

    def __init__(self, cfg=None, device="cuda", root_path=None, **kwargs):
        get = (lambda k, d=None: cfg.get(k, d)) if hasattr(cfg, "get") else (lambda k, d=None: getattr(cfg, k, d))
        for flag in ("remove_environment", "add_rgb_to_points", "remove_outside_points", "random_crop", "random_select"):
            if get(flag, False):
                raise NotImplementedError("Preprocess(%s=True) is not on the Det3D hot path" % flag)
        if get("reference_detections", None) is not None or get("min_points_in_gt", -1) > 0:
            raise NotImplementedError("Preprocess: reference_detections / min_points_in_gt are not supported")
        self.mode = get("mode")
        self.device = device
        self.shuffle_points = get("shuffle_points")
        self.remove_unknown = get("remove_unknown_examples")
        self.symmetry_intensity = get("symmetry_intensity", False)
        if self.mode == "train":
            self.gt_rotation_noise = get("gt_rot_noise")
            self.gt_loc_noise_std = get("gt_loc_noise")
            self.global_rotation_noise = get("global_rot_noise")
            self.global_scaling_noise = get("global_scale_noise")
            self.global_random_rot_range = get("global_rot_per_obj_range")
            self.remove_points_after_sample = get("remove_points_after_sample")
            self.class_names = list(get("class_names"))
            db = get("db_sampler")
            self.db_sampler = db if (db is None or hasattr(db, "sample_all")) else \
                build_dbsampler(db, device=device, root_path=root_path)

    def __call__(self, res, info):
        import torch

        from . import augment

        res["mode"] = self.mode
        points = res["lidar"]["combined"] if res["type"] == "NuScenesDataset" else res["lidar"]["points"]
        if not torch.is_tensor(points):
            points = torch.from_numpy(np.ascontiguousarray(points))
        points = points.to(self.device)
        if self.mode == "train":
            anno = res["lidar"]["annotations"]
            gt = {"gt_boxes": np.asarray(anno["boxes"]), "gt_names": np.array(anno["names"]).reshape(-1)}
            gt["difficulty"] = np.asarray(anno["difficulty"]) if "difficulty" in anno else \
                np.zeros([gt["gt_boxes"].shape[0]], dtype=np.int32)
            keep = np.array([i for i, x in enumerate(gt["gt_names"]) if x not in ("DontCare", "ignore")], dtype=np.int64)
            _dict_select(gt, keep)
            if self.remove_unknown:
                _dict_select(gt, np.logical_not(gt["difficulty"] == -1))
            gt.pop("difficulty")
            mask = np.array([n in self.class_names for n in gt["gt_names"]], dtype=np.bool_)
            if self.db_sampler:
                s = self.db_sampler.sample_all(gt["gt_boxes"], gt["gt_names"])
                if s is not None:
                    points, gt["gt_boxes"], gt["gt_names"], _ = self.db_sampler.paste(
                        points, gt["gt_boxes"], gt["gt_names"], s, self.remove_points_after_sample)
                    mask = np.concatenate([mask, s["gt_masks"]], axis=0)
            boxes = np.asarray(gt["gt_boxes"])
            n = boxes.shape[0]
            # per-object noise: draws first (reference order), geometry on the device as a batch of one frame
            p = augment.sample_object_noise_params(boxes, 100, self.gt_rotation_noise, self.gt_loc_noise_std,
                                                   self.global_random_rot_range)
            off = torch.tensor([0, points.shape[0]], dtype=torch.int32, device=points.device)
            bt = torch.from_numpy(boxes.astype(np.float32))[None].to(points.device)
            if n:
                points, bt, _ = augment.noise_per_object(
                    points, off, bt, torch.tensor([n], device=points.device),
                    dict(loc_noises=p["loc_noises"][None], rot_noises=p["rot_noises"][None]),
                    valid_mask=torch.from_numpy(mask)[None].to(points.device))
            sel = torch.from_numpy(np.nonzero(mask)[0]).to(points.device)
            bt = bt[:, sel]
            gt["gt_names"] = gt["gt_names"][mask]
            gt["gt_classes"] = np.array([self.class_names.index(x) + 1 for x in gt["gt_names"]], dtype=np.int32)
            g = augment.sample_global_params(1, flip="both" if res["type"] == "NuScenesDataset" else "y",
                                             rotation=self.global_rotation_noise, scaling=tuple(self.global_scaling_noise))
            points, bt = augment.apply_global_augmentation(points, off, bt, g)
            gt["gt_boxes"] = bt[0].cpu().numpy().astype(boxes.dtype if boxes.size else np.float32)
        if self.shuffle_points:
            order = np.arange(points.shape[0])
            np.random.shuffle(order)  # the same permutation np.random.shuffle(points) applies to the rows
            points = points[torch.from_numpy(order).to(points.device)]
        if self.symmetry_intensity:
            points = points.clone()
            points[:, -1] -= 0.5
        res["lidar"]["points"] = points
        if self.mode == "train":
            res["lidar"]["annotations"] = gt
        return res, info


@PIPELINES.register_module
class LoadPointCloudAnnotations:
    """datasets/pipelines/loading.py:166-222: the frame's ground truth as `res["lidar"]["annotations"]`.  nuScenes /
    Lyft infos already hold lidar boxes; KITTI boxes come in camera coordinates with the origin at the bottom face:
    DontCare rows dropped, (location, l h w, rotation_y) -> lidar (x y z, w l h, r) through the inverse of
    R0_rect @ Tr_velo_to_cam, then moved to the box centre (box_np_ops.py:909-930, :1346-1349)."""

    def __init__(self, with_bbox=True, **kwargs):
        pass

    def __call__(self, res, info):
        if res["type"] in ("NuScenesDataset", "LyftDataset") and "gt_boxes" in info:
            res["lidar"]["annotations"] = {"boxes": info["gt_boxes"].astype(np.float32), "names": info["gt_names"],
                                           "tokens": info["gt_boxes_token"],
                                           "velocities": info["gt_boxes_velocity"].astype(np.float32)}
        elif res["type"] == "KittiDataset":
            calib = info["calib"]
            res["calib"] = {"rect": calib["R0_rect"], "Trv2c": calib["Tr_velo_to_cam"], "P2": calib["P2"]}
            if "annos" in info:
                annos = info["annos"]
                keep = [i for i, x in enumerate(annos["name"]) if x != "DontCare"]
                annos = {k: v[keep] for k, v in annos.items()}
                cam = np.concatenate([annos["location"], annos["dimensions"], annos["rotation_y"][..., np.newaxis]],
                                     axis=1).astype(np.float32)
                xyz1 = np.concatenate([cam[:, 0:3], np.ones((cam.shape[0], 1))], axis=-1)
                xyz = (xyz1 @ np.linalg.inv((calib["R0_rect"] @ calib["Tr_velo_to_cam"]).T))[:, :3]
                boxes = np.concatenate([xyz, cam[:, 5:6], cam[:, 3:4], cam[:, 4:5], cam[:, 6:7]], axis=1)  # w, l, h
                boxes[:, :3] += boxes[:, 3:6] * (np.array([0.5, 0.5, 0.5], boxes.dtype) - np.array([0.5, 0.5, 0], boxes.dtype))
                res["lidar"]["annotations"] = {"boxes": boxes, "names": annos["name"]}
                res["cam"]["annotations"] = {"boxes": annos["bbox"], "names": annos["name"]}
        else:
            raise NotImplementedError(res["type"])
        return res, info


@PIPELINES.register_module
class Reformat:
    """datasets/pipelines/formating.py:14-61: the per-frame `res` tree flattened into the example dict that
    collate_kitti batches (collate.py)."""

    def __init__(self, **kwargs):
        pass

    def __call__(self, res, info):
        vox = res["lidar"]["voxels"]
        out = dict(metadata=res["metadata"], points=res["lidar"]["points"], voxels=vox["voxels"], shape=vox["shape"],
                   num_points=vox["num_points"], num_voxels=vox["num_voxels"], coordinates=vox["coordinates"],
                   anchors=res["lidar"]["targets"]["anchors"])
        if res.get("calib", None):
            out["calib"] = res["calib"]
        if res["mode"] != "test":
            out["annos"] = res["lidar"]["annotations"]
        if res["mode"] == "train":
            if res["lidar"].get("ground_plane", None):
                out["ground_plane"] = res["lidar"]["ground_plane"]
            t = res["lidar"]["targets"]
            out.update(labels=t["labels"], reg_targets=t["reg_targets"], reg_weights=t["reg_weights"])
        return out, info


@PIPELINES.register_module
class Compose:
    """datasets/pipelines/compose.py:7-27: the steps of a config's `train_pipeline` / `test_pipeline` list in order;
    a step returning None drops the frame."""

    def __init__(self, transforms):
        self.transforms = []
        for t in transforms:
            if isinstance(t, dict):
                t = build_from_cfg(t, PIPELINES)
            elif not callable(t):
                raise TypeError("transform must be callable or a dict")
            self.transforms.append(t)

    def __call__(self, res, info):
        for t in self.transforms:
            res, info = t(res, info)
            if res is None:
                return None
        return res, info
