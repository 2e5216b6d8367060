"""Host side of the HIP voxelizer.

Mirrors det3d/ops/point_cloud/point_cloud_ops.py:112-184 (`points_to_voxel`) and
det3d/core/input/voxel_generator.py:5-43 (`VoxelGenerator`) -- same names, argument
meaning and return layout -- on top of the C ABI `d3d_voxelize` (include/d3d.h).
`voxelize_batch` is the MI355X-native entry: a whole collated batch
([sum N, C] points already in HBM) in one enqueue, no host sync.
"""
import numpy as np
import torch

from . import _lib, arena as mem


def grid_size_of(voxel_size, coors_range):
    """round((hi - lo) / vs) in float32 (point_cloud_ops.py:26-29, voxel_generator.py:10-11)."""
    vs = np.asarray(voxel_size, dtype=np.float32)
    rg = np.asarray(coors_range, dtype=np.float32)
    return np.round((rg[3:] - rg[:3]) / vs).astype(np.int64)


class VoxelBatch:
    """Result of `voxelize_batch`.  Buffers are sized for batch*max_voxels rows;
    `num_voxels` ([batch+1] int32, device) holds per-frame counts and the total.
    `.sync()` does the one D2H read and returns views trimmed to the total."""

    def __init__(self, coors, num_points, voxels, mean, num_voxels):
        self.coors_cap = coors
        self.num_points_cap = num_points
        self.voxels_cap = voxels
        self.mean_cap = mean
        self.num_voxels = num_voxels
        self._host_counts = None

    def counts(self):
        if self._host_counts is None:
            self._host_counts = self.num_voxels.cpu().numpy()
        return self._host_counts

    def sync(self):
        m = int(self.counts()[-1])
        out = dict(coordinates=self.coors_cap[:m], num_points=self.num_points_cap[:m],
                   num_voxels=self.counts()[:-1].astype(np.int64))
        if self.voxels_cap is not None:
            out["voxels"] = self.voxels_cap[:m]
        if self.mean_cap is not None:
            out["mean"] = self.mean_cap[:m]
        return out


def voxelize_batch(points, frame_offsets, voxel_size, coors_range, max_points, max_voxels,
                   want_voxels=True, mean_channels=0):
    """points: CUDA float32 [sum N, C]; frame_offsets: CUDA int32 [B+1].

    Enqueues d3d_voxelize on the current stream and returns a VoxelBatch.
    """
    if not points.is_cuda:
        raise _lib.D3dError("voxelize_batch needs device points (no CPU path in det3d_amd)")
    L = _lib.lib()
    points = points.contiguous()
    assert points.dtype == torch.float32 and points.dim() == 2
    frame_offsets = frame_offsets.to(device=points.device, dtype=torch.int32).contiguous()
    batch = frame_offsets.numel() - 1
    n, c = points.shape
    grid = grid_size_of(voxel_size, coors_range)
    rows = batch * max_voxels
    dev = points.device
    # (from the calling thread's arena when it has one -- the geometry ring slot of a prefetched build, arena.py -- else
    # torch.empty: the direct API never sees an arena)
    coors = mem.empty((rows, 4), torch.int32, dev)
    num_points = mem.empty((rows,), torch.int32, dev)
    voxels = mem.empty((rows, max_points, c), torch.float32, dev) if want_voxels else None
    mean = mem.empty((rows, mean_channels), torch.float32, dev) if mean_channels else None
    num_voxels = mem.empty((batch + 1,), torch.int32, dev)
    ws_bytes = L.d3d_voxelize_workspace_bytes(n, batch, max_points, max_voxels)
    ws = mem.empty((ws_bytes,), torch.uint8, dev)
    rc = L.d3d_voxelize(points.data_ptr(), n, c, frame_offsets.data_ptr(), batch,
                        _lib.f32_array(np.asarray(voxel_size, dtype=np.float32)),
                        _lib.f32_array(np.asarray(coors_range, dtype=np.float32)),
                        _lib.i32_array(grid), int(max_points), int(max_voxels),
                        coors.data_ptr(), num_points.data_ptr(), _lib.ptr(voxels), _lib.ptr(mean),
                        int(mean_channels), num_voxels.data_ptr(), ws.data_ptr(), ws_bytes,
                        _lib.stream_ptr(dev))
    _lib.check(rc, "d3d_voxelize")
    return VoxelBatch(coors, num_points, voxels, mean, num_voxels)


def points_to_voxel(points, voxel_size, coors_range, max_points=35, reverse_index=True,
                    max_voxels=20000, device="cuda"):
    """Drop-in for det3d.ops.point_cloud.point_cloud_ops.points_to_voxel.

    numpy in -> numpy out (voxels [M,T,C], coordinates [M,3] zyx int32,
    num_points_per_voxel [M] int32); a CUDA tensor in -> CUDA tensors out.
    Runs on the GPU either way; a numpy call pays H2D + D2H.
    """
    if not reverse_index:
        raise NotImplementedError("only reverse_index=True (zyx) is on the Det3D hot path")
    is_np = isinstance(points, np.ndarray)
    pts = torch.from_numpy(np.ascontiguousarray(points, dtype=np.float32)).to(device) if is_np \
        else points.float()
    off = torch.tensor([0, pts.shape[0]], dtype=torch.int32, device=pts.device)
    vb = voxelize_batch(pts, off, voxel_size, coors_range, max_points, max_voxels).sync()
    voxels, coors, num = vb["voxels"], vb["coordinates"][:, 1:], vb["num_points"]
    if is_np:
        return voxels.cpu().numpy(), coors.cpu().numpy(), num.cpu().numpy()
    return voxels, coors.contiguous(), num


class VoxelGenerator:
    """det3d/core/input/voxel_generator.py:5-43, same constructor and properties."""

    def __init__(self, voxel_size, point_cloud_range, max_num_points, max_voxels=20000):
        point_cloud_range = np.array(point_cloud_range, dtype=np.float32)
        voxel_size = np.array(voxel_size, dtype=np.float32)
        self._voxel_size = voxel_size
        self._point_cloud_range = point_cloud_range
        self._max_num_points = max_num_points
        self._max_voxels = max_voxels
        self._grid_size = grid_size_of(voxel_size, point_cloud_range)

    def generate(self, points, max_voxels=20000):
        # like the reference (:19-27) the max_voxels argument is ignored
        return points_to_voxel(points, self._voxel_size, self._point_cloud_range,
                               self._max_num_points, True, self._max_voxels)

    def generate_batch(self, points, frame_offsets, want_voxels=True, mean_channels=0):
        return voxelize_batch(points, frame_offsets, self._voxel_size, self._point_cloud_range,
                              self._max_num_points, self._max_voxels, want_voxels, mean_channels)

    @property
    def voxel_size(self):
        return self._voxel_size

    @property
    def max_num_points_per_voxel(self):
        return self._max_num_points

    @property
    def point_cloud_range(self):
        return self._point_cloud_range

    @property
    def grid_size(self):
        return self._grid_size
