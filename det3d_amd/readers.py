"""Readers (voxel feature encoders).  Same registry names, constructor kwargs and forward
signatures as the reference:
  VoxelFeatureExtractorV3   det3d/models/readers/voxel_encoder.py:197-211
  PFNLayer, PillarFeatureNet det3d/models/readers/pillar_encoder.py:17-155
On the fused MI355X path the mean VFE is already produced by d3d_voxelize
(`mean_feats`); VoxelFeatureExtractorV3 accepts those [M, C] rows and passes them on.
"""
import torch
from torch import nn
from torch.nn import functional as F

from . import _lib, arena as mem
from .nn_utils import build_norm_layer
from .registry import READERS

FUSE_PFN = True  # single-layer PillarFeatureNet on CUDA goes through d3d_pfn_fwd / d3d_pfn_bwd


class _PFNFn(torch.autograd.Function):
    """decorate + Linear + BatchNorm1d + ReLU + max over T in three launches (d3d_pfn_fwd: statistics slabs, fixed-order fold, apply)."""

    @staticmethod
    @_lib.ctx_forward
    def forward(ctx, voxels, num_points, coors, weight, gamma, beta, running_mean, running_var, momentum,
                eps, training, geom):
        L = _lib.lib()
        voxels = voxels.contiguous()
        num_points = num_points.int().contiguous()
        coors = coors.int().contiguous()
        m, t, c = voxels.shape
        f = weight.shape[0]
        dev = voxels.device
        lease = mem.Lease()  # the pillar count follows the batch: step arena, if the caller runs under one (arena.py)
        out = mem.empty((m, f), torch.float32, dev, lease.arena)
        argmax = mem.empty((m, f), torch.int32, dev, lease.arena)
        save = torch.empty((2, f), dtype=torch.float32, device=dev)
        ws = mem.empty((L.d3d_pfn_ws_floats(m, c, f),), torch.float32, dev, lease.arena)  # slabs of the fixed-order sums
        w = weight.detach().float().contiguous()
        vx, vy, xo, yo = geom
        rc = L.d3d_pfn_fwd(voxels.data_ptr(), num_points.data_ptr(), coors.data_ptr(), m, t, c, vx, vy, xo, yo,
                           w.data_ptr(), f, _lib.ptr(gamma), _lib.ptr(beta), _lib.ptr(running_mean),
                           _lib.ptr(running_var), float(momentum), float(eps), int(training), out.data_ptr(),
                           argmax.data_ptr(), save[0].data_ptr(), save[1].data_ptr(), ws.data_ptr(), ws.numel(),
                           _lib.stream_ptr(dev))
        _lib.check(rc, "d3d_pfn_fwd")
        ctx.save_for_backward(voxels, num_points, coors, w, gamma, save, out, argmax)
        ctx.geom = geom
        ctx.training = training
        ctx.lease = lease
        return out

    @staticmethod
    @_lib.ctx_backward
    def backward(ctx, dout):
        if not ctx.training:
            raise _lib.D3dError("fused PFN backward is implemented for training mode only")
        L = _lib.lib()
        voxels, num_points, coors, w, gamma, save, out, argmax = ctx.saved_tensors
        m, t, c = voxels.shape
        f = w.shape[0]
        vx, vy, xo, yo = ctx.geom
        dout = dout.float().contiguous()
        dw = torch.empty_like(w)
        g = torch.empty((2, f), dtype=torch.float32, device=w.device)
        ws = mem.empty((L.d3d_pfn_ws_floats(m, c, f),), torch.float32, w.device, ctx.lease.check("PillarFeatureNet"))
        rc = L.d3d_pfn_bwd(voxels.data_ptr(), num_points.data_ptr(), coors.data_ptr(), m, t, c, vx, vy, xo, yo,
                           w.data_ptr(), f, _lib.ptr(gamma), save[0].data_ptr(), save[1].data_ptr(),
                           out.data_ptr(), dout.data_ptr(), argmax.data_ptr(), dw.data_ptr(), g[0].data_ptr(),
                           g[1].data_ptr(), ws.data_ptr(), ws.numel(), _lib.stream_ptr(w.device))
        _lib.check(rc, "d3d_pfn_bwd")
        return None, None, None, dw, g[0], g[1], None, None, None, None, None, None


@READERS.register_module
class VoxelFeatureExtractorV3(nn.Module):
    def __init__(self, num_input_features=4, norm_cfg=None, name="VoxelFeatureExtractorV3"):
        super().__init__()
        self.name = name
        self.num_input_features = num_input_features

    def forward(self, features, num_voxels, coors=None):
        if features.dim() == 2:
            # already-averaged rows from the fused voxelizer (d3d_voxelize mean_feats)
            return features[:, : self.num_input_features].contiguous()
        points_mean = features[:, :, : self.num_input_features].sum(dim=1, keepdim=False) / \
            num_voxels.type_as(features).view(-1, 1)
        return points_mean.contiguous()


def decorate_pillars(points, counts, coors, cell, origin, with_distance=False):
    """[M, T, C] padded pillars -> [M, T, C + 5 (+1)] rows: the point, its offset from the pillar's mean point (x, y,
    z), its offset from the pillar's cell centre (x, y), optionally its range; padded slots zeroed.
    (what PillarFeatureNet.forward computes in front of its layers, pillar_encoder.py:115-150)
    cell = (vx, vy); origin = centre of cell (0, 0)."""
    m, t, _ = points.shape
    real = (torch.arange(t, device=points.device).unsqueeze(0) < counts.view(m, 1)).unsqueeze(-1)
    xyz = points[..., :3]
    centroid = xyz.sum(1, keepdim=True) / counts.to(points.dtype).view(m, 1, 1)
    ij = torch.stack([coors[:, 3], coors[:, 2]], 1).to(points.dtype)                 # (x index, y index) of the pillar
    centre = ij * points.new_tensor(cell) + points.new_tensor(origin)
    cols = [points, xyz - centroid, points[..., :2] - centre.unsqueeze(1)]
    if with_distance:
        cols.append(xyz.norm(dim=2, keepdim=True))
    return torch.cat(cols, 2) * real.to(points.dtype)


class PFNLayer(nn.Module):
    """Linear -> BatchNorm over all M*T point rows -> ReLU -> max over the T slots of a pillar
    (registry / state-dict names of pillar_encoder.py:17-60: `linear`, `norm`)."""

    def __init__(self, in_channels, out_channels, norm_cfg=None, last_layer=False):
        super().__init__()
        self.name = "PFNLayer"
        self.last_vfe = last_layer
        self.units = out_channels if last_layer else out_channels // 2   # inner layers concatenate [x, max x]
        self.norm_cfg = norm_cfg if norm_cfg is not None else dict(type="BN1d", eps=1e-3, momentum=0.01)
        self.linear = nn.Linear(in_channels, self.units, bias=False)
        self.norm = build_norm_layer(self.norm_cfg, self.units)[1]

    def forward(self, inputs):
        m, t, _ = inputs.shape
        y = self.linear(inputs)
        y = F.relu(self.norm(y.transpose(1, 2)).transpose(1, 2))   # the norm layer sees [M, C, T]: per-channel statistics
        top = y.amax(1, keepdim=True)
        if self.last_vfe:
            return top
        return torch.cat([y, top.expand(m, t, self.units)], 2)


@READERS.register_module
class PillarFeatureNet(nn.Module):
    def __init__(self, num_input_features=4, num_filters=(64,), with_distance=False,
                 voxel_size=(0.2, 0.2, 4), pc_range=(0, -40, -3, 70.4, 40, 1), norm_cfg=None):
        super().__init__()
        self.name = "PillarFeatureNet"
        assert len(num_filters) > 0
        self.num_input = num_input_features
        num_input_features += 5
        if with_distance:
            num_input_features += 1
        self._with_distance = with_distance
        num_filters = [num_input_features] + list(num_filters)
        layers = []
        for i in range(len(num_filters) - 1):
            layers.append(PFNLayer(num_filters[i], num_filters[i + 1], norm_cfg=norm_cfg,
                                   last_layer=i >= len(num_filters) - 2))
        self.pfn_layers = nn.ModuleList(layers)
        self.vx = voxel_size[0]
        self.vy = voxel_size[1]
        self.x_offset = self.vx / 2 + pc_range[0]
        self.y_offset = self.vy / 2 + pc_range[1]

    def _fusable(self, features):
        if not (FUSE_PFN and features.is_cuda and features.dtype == torch.float32):
            return False
        if len(self.pfn_layers) != 1 or self._with_distance or features.shape[0] < 1:
            return False
        pfn = self.pfn_layers[0]
        bn = pfn.norm
        ok_bn = type(bn) is nn.BatchNorm1d and bn.track_running_stats and bn.momentum is not None \
            and bn.weight is not None
        if not bn.training and torch.is_grad_enabled() and pfn.linear.weight.requires_grad:
            return False  # eval-mode backward: PyTorch path
        return ok_bn and pfn.units <= 64 and 3 <= features.shape[2] <= 5

    def forward(self, features, num_voxels, coors):
        if self._fusable(features):
            pfn = self.pfn_layers[0]
            bn = pfn.norm
            out = _PFNFn.apply(features, num_voxels, coors, pfn.linear.weight, bn.weight, bn.bias,
                               bn.running_mean, bn.running_var, bn.momentum, bn.eps, bn.training,
                               (float(self.vx), float(self.vy), float(self.x_offset), float(self.y_offset)))
            return out.squeeze()
        x = decorate_pillars(features, num_voxels, coors, (self.vx, self.vy), (self.x_offset, self.y_offset),
                             self._with_distance)
        for layer in self.pfn_layers:
            x = layer(x)
        return x.squeeze()
