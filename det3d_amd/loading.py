"""Point-cloud loading step of the input pipeline: det3d/datasets/pipelines/loading.py:17-163.

The reference reads the key frame and its sweeps file by file in a DataLoader worker and, for nuScenes, runs three
NumPy passes per sweep (drop the points close to the sensor, 4x4 transform into the key frame, time-lag column).  File
reads stay on the host here as well (they are IO); everything after them runs ONCE on the concatenated sweeps on the
device: one mask, one batched 4x4 transform selected through a sweep index (in float64, as the reference's float64
matrices make NumPy do), one concatenation.  `LoadPointCloudFromFile` keeps the reference's `(res, info)` step
interface and result keys (`points`, `times`, `combined`), with device tensors as values.
"""
import os

import numpy as np
import torch

MIN_DISTANCE = 1.0  # read_sweep: remove_close(points_sweep, 1.0) in the sweep's own frame


def read_bin(path, width=5, keep=4):
    """a float32 file of `width` columns -> [N, keep]; a trailing partial row is dropped (read_file :17-32)"""
    raw = np.fromfile(str(path), dtype=np.float32)
    raw = raw[: raw.size - raw.size % width]
    return raw.reshape(-1, width)[:, :keep]


def assemble_sweeps(key_points, sweeps, device):
    """key_points [N0, 4] numpy; sweeps: list of (points [Ni, 4] numpy, transform_matrix 4x4 or None, time_lag).
    -> (points [M, 4], times [M, 1]) on `device`, rows in the reference's order: key frame, then the sweeps in the
    given order with the points closer than MIN_DISTANCE (|x| and |y| both below it) removed."""
    clouds = [key_points] + [s[0] for s in sweeps]
    sid = np.repeat(np.arange(len(clouds)), [len(c) for c in clouds])
    pts = torch.from_numpy(np.concatenate(clouds, axis=0)).to(device)
    sid_t = torch.from_numpy(sid).to(device)
    mats = np.stack([np.eye(4)] + [np.eye(4) if s[1] is None else np.asarray(s[1], np.float64) for s in sweeps])
    lags = np.array([0.0] + [float(s[2]) for s in sweeps])
    close = (pts[:, 0].abs() < MIN_DISTANCE) & (pts[:, 1].abs() < MIN_DISTANCE) & (sid_t > 0)
    keep = ~close
    pts, sid_t = pts[keep], sid_t[keep]
    m = torch.from_numpy(mats).to(device)[sid_t]                              # [M, 4, 4] float64
    xyz1 = torch.cat([pts[:, :3].double(), torch.ones((pts.shape[0], 1), dtype=torch.float64, device=pts.device)], 1)
    out = pts.clone()
    out[:, :3] = torch.einsum("nij,nj->ni", m, xyz1)[:, :3].to(pts.dtype)
    times = torch.from_numpy(lags).to(device)[sid_t].to(pts.dtype)[:, None]
    return out, times


class LoadPointCloudFromFile:
    def __init__(self, dataset="KittiDataset", device="cuda", **kwargs):
        self.type = dataset
        self.device = device

    def __call__(self, res, info):
        res["type"] = self.type
        if self.type == "KittiDataset":
            rel = info["point_cloud"]["velodyne_path"]
            path = rel if os.path.isabs(rel) else os.path.join(res["metadata"]["image_prefix"], rel)
            folder, name = os.path.split(path)
            reduced = os.path.join(os.path.dirname(folder), os.path.basename(folder) + "_reduced", name)
            if os.path.exists(reduced):  # the camera-frustum-cropped copy, when create_reduced_point_cloud made one
                path = reduced
            c = res["metadata"]["num_point_features"]
            res["lidar"]["points"] = torch.from_numpy(np.fromfile(path, dtype=np.float32).reshape(-1, c)).to(self.device)
        elif self.type == "NuScenesDataset":
            nsweeps = res["lidar"]["nsweeps"]
            key = read_bin(info["lidar_path"])
            if nsweeps - 1 > len(info["sweeps"]):
                raise AssertionError("nsweeps {} should not greater than list length {}.".format(nsweeps, len(info["sweeps"])))
            chosen = np.random.choice(len(info["sweeps"]), nsweeps - 1, replace=False)  # the reference's draw
            sweeps = [(read_bin(info["sweeps"][i]["lidar_path"]), info["sweeps"][i]["transform_matrix"],
                       info["sweeps"][i]["time_lag"]) for i in chosen]
            points, times = assemble_sweeps(key, sweeps, self.device)
            res["lidar"]["points"], res["lidar"]["times"] = points, times
            res["lidar"]["combined"] = torch.cat([points, times], dim=1)
        elif self.type == "LyftDataset":
            res["lidar"]["points"] = torch.from_numpy(read_bin(info["ref_info"]["LIDAR_TOP"]["lidar_path"])).to(self.device)
        else:
            raise NotImplementedError(self.type)
        return res, info
