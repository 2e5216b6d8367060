"""Backbones of the hot path, registry names / kwargs / forward signatures as in the
reference:
  SpMiddleFHD        det3d/models/backbones/scn.py:92-197   (KITTI SECOND)
  SpMiddleResNetFHD  det3d/models/backbones/scn.py:308-370  (nuScenes / Lyft CBGS)
  SparseBasicBlock   det3d/models/backbones/scn.py:46-89
  PointPillarsScatter det3d/models/readers/pillar_encoder.py:158-211
Module nesting (and therefore state-dict keys `backbone.middle_conv.<i>.*`) is identical.
The sparse stacks run on the HIP kernels behind det3d_amd.spconv; `forward` plans all
rulebooks of the stack with one host sync before running the layers.
"""
import numpy as np
import torch
from torch import nn

from . import spconv
from .nn_utils import build_norm_layer
from .registry import BACKBONES
from .spconv import SparseConv3d, SubMConv3d


def conv3x3(in_planes, out_planes, stride=1, indice_key=None, bias=True):
    return spconv.SubMConv3d(in_planes, out_planes, kernel_size=3, stride=stride, padding=1,
                             bias=bias, indice_key=indice_key)


def conv1x1(in_planes, out_planes, stride=1, indice_key=None, bias=True):
    return spconv.SubMConv3d(in_planes, out_planes, kernel_size=1, stride=stride, padding=1,
                             bias=bias, indice_key=indice_key)


def _bn_act(bn, feats, relu_module=None):
    """norm [+ ReLU] on sparse features outside the fused BatchNorm1d path: the cross-rank SyncBN runs on its split
    sweeps with the ReLU folded in, anything else is the plain module call"""
    if type(bn).__name__ == "DistributedSyncBN" and feats.shape[0] > 0:
        from .syncbn import sync_bn_relu

        y = sync_bn_relu(bn, feats, relu_module is not None)
        if y is not None:
            return y
    y = bn(feats)
    return relu_module(y) if relu_module is not None else y


class SparseBasicBlock(spconv.SparseModule):
    expansion = 1

    def __init__(self, inplanes, planes, stride=1, norm_cfg=None, downsample=None, indice_key=None):
        super().__init__()
        if norm_cfg is None:
            norm_cfg = dict(type="BN1d", eps=1e-3, momentum=0.01)
        bias = norm_cfg is not None  # always True after defaulting, as in the reference (:60-63)
        self.conv1 = conv3x3(inplanes, planes, stride, indice_key=indice_key, bias=bias)
        self.bn1 = build_norm_layer(norm_cfg, planes)[1]
        self.relu = nn.ReLU()
        self.conv2 = conv3x3(planes, planes, indice_key=indice_key, bias=bias)
        self.bn2 = build_norm_layer(norm_cfg, planes)[1]
        self.downsample = downsample
        self.stride = stride

    def conv_layers(self):
        return [self.conv1, self.conv2]

    def forward(self, x):
        from . import sparse_block

        rb = sparse_block.rulebook_of(spconv, self, x)
        if rb is not None:  # the whole block as one autograd node, residual add + ReLU inside the BatchNorm sweeps
            return sparse_block.run(spconv, self, x, rb)
        identity = x
        n = x.indices.shape[0]
        if spconv.FUSE_BN_RELU and n != 0 and spconv._fusable_bn(self.bn1, x.features, channels=self.conv1.out_channels):
            # statistics of bn1 / bn2 from the conv epilogues; conv2 is the only consumer of bn1's output, so
            # its data-gradient launch also reduces bn1's backward sums
            out, link = spconv.conv_bn_relu(self.conv1, self.bn1, True, x, want_link=True)
        else:
            out, link = self.conv1(x), None
            out.features = _bn_act(self.bn1, out.features, self.relu)
        if spconv.FUSE_BN_RELU and n != 0 and spconv._fusable_bn(self.bn2, out.features, channels=self.conv2.out_channels):
            out, _ = spconv.conv_bn_relu(self.conv2, self.bn2, False, out, link_in=link)
        else:
            out = self.conv2(out)
            out.features = self.bn2(out.features)
        if self.downsample is not None:
            identity = self.downsample(x)
        out.features = out.features + identity.features
        out.features = self.relu(out.features)
        return out


class _SparseMiddle(nn.Module):
    """shared forward of the two sparse backbones (scn.py:184-197 / :357-370)."""

    def plan_geometry(self, coors, batch_size, input_shape):
        """Every rulebook of the sparse stack for these voxel coordinates (one host sync on the CURRENT stream),
        as the dict `forward(..., plan=)` accepts: lets a caller build the geometry of the next batch on a second
        stream while the current batch trains (prefetch.GeometryPrefetcher)."""
        sparse_shape = np.array(input_shape[::-1]) + [1, 0, 0]
        coors = coors.int()
        probe = spconv.SparseConvTensor(coors.new_empty((coors.shape[0], 0), dtype=torch.float32), coors,
                                        sparse_shape.tolist(), batch_size)
        return dict(coors=coors, indice_dict=self.middle_conv.plan(probe))

    def _run(self, voxel_features, coors, batch_size, input_shape, plan=None):
        sparse_shape = np.array(input_shape[::-1]) + [1, 0, 0]
        coors = coors.int()
        if torch.is_autocast_enabled():
            voxel_features = voxel_features.to(torch.get_autocast_dtype("cuda"))
        ret = spconv.SparseConvTensor(voxel_features, coors, sparse_shape.tolist(), batch_size)
        if plan is not None and plan["coors"].data_ptr() == coors.data_ptr() and plan["coors"].shape == coors.shape:
            ticket = plan.get("ring_ticket")
            if ticket is not None and ticket[0].generation != ticket[1]:
                from ._lib import D3dError

                raise D3dError("the rulebooks of this example lived in slot '%s' of a GeometryPrefetcher ring that has been "
                               "rewound since (generation %d -> %d): the example was kept for more than ring - 1 further "
                               "builds; use GeometryPrefetcher(ring=0) for such a loop" % (ticket[0].name, ticket[1], ticket[0].generation))
            ret.indice_dict = plan["indice_dict"]  # built ahead of time for exactly these coordinates
        else:
            self.middle_conv.plan(ret)  # every rulebook of the stack, one host sync
        spconv.ARENA.reset(voxel_features.device)  # rewind the fold workspaces of this pass's conv epilogues (no launch)
        ret = self.middle_conv(ret)
        self.last_indice_dict = ret.indice_dict
        if voxel_features.is_cuda and self.training:
            from . import prefetch

            prefetch.mark_sparse_forward(voxel_features.device)  # (runtime "geometry_gate": where the next build may start)
        # [B, C, D, H, W] -> [B, C*D, H, W]; on the GPU as a channels-last tensor written directly by the scatter
        # kernel: the layout the dense conv kernels (dense_stack.py) and the BatchNorm sweeps work on
        return ret.dense_2d(channels_last=getattr(self, "dense_channels_last", voxel_features.is_cuda))

    def init_weights(self, pretrained=None):
        pass

    def forward(self, voxel_features, coors, batch_size, input_shape, plan=None):
        return self._run(voxel_features, coors, batch_size, input_shape, plan)


@BACKBONES.register_module
class SpMiddleFHD(_SparseMiddle):
    def __init__(self, num_input_features=128, norm_cfg=None, name="SpMiddleFHD", **kwargs):
        super().__init__()
        self.name = name
        self.dcn = None
        self.zero_init_residual = False
        if norm_cfg is None:
            norm_cfg = dict(type="BN1d", eps=1e-3, momentum=0.01)

        def bn(c):
            return build_norm_layer(norm_cfg, c)[1]

        layers = [SubMConv3d(num_input_features, 16, 3, bias=False, indice_key="subm0"), bn(16), nn.ReLU(),
                  SubMConv3d(16, 16, 3, bias=False, indice_key="subm0"), bn(16), nn.ReLU(),
                  SparseConv3d(16, 32, 3, 2, padding=1, bias=False), bn(32), nn.ReLU()]
        for _ in range(2):
            layers += [SubMConv3d(32, 32, 3, indice_key="subm1", bias=False), bn(32), nn.ReLU()]
        layers += [SparseConv3d(32, 64, 3, 2, padding=1, bias=False), bn(64), nn.ReLU()]
        for _ in range(3):
            layers += [SubMConv3d(64, 64, 3, indice_key="subm2", bias=False), bn(64), nn.ReLU()]
        layers += [SparseConv3d(64, 64, 3, 2, padding=[0, 1, 1], bias=False), bn(64), nn.ReLU()]
        for _ in range(3):
            layers += [SubMConv3d(64, 64, 3, indice_key="subm3", bias=False), bn(64), nn.ReLU()]
        layers += [SparseConv3d(64, 64, (3, 1, 1), (2, 1, 1), bias=False), bn(64), nn.ReLU()]
        self.middle_conv = spconv.SparseSequential(*layers)


@BACKBONES.register_module
class SpMiddleResNetFHD(_SparseMiddle):
    def __init__(self, num_input_features=128, norm_cfg=None, name="SpMiddleResNetFHD", **kwargs):
        super().__init__()
        self.name = name
        self.dcn = None
        self.zero_init_residual = False
        if norm_cfg is None:
            norm_cfg = dict(type="BN1d", eps=1e-3, momentum=0.01)

        def bn(c):
            return build_norm_layer(norm_cfg, c)[1]

        def block(c, key):
            return SparseBasicBlock(c, c, norm_cfg=norm_cfg, indice_key=key)

        self.middle_conv = spconv.SparseSequential(
            SubMConv3d(num_input_features, 16, 3, bias=False, indice_key="res0"), bn(16), nn.ReLU(),
            block(16, "res0"), block(16, "res0"),
            SparseConv3d(16, 32, 3, 2, padding=1, bias=False), bn(32), nn.ReLU(),
            block(32, "res1"), block(32, "res1"),
            SparseConv3d(32, 64, 3, 2, padding=1, bias=False), bn(64), nn.ReLU(),
            block(64, "res2"), block(64, "res2"),
            SparseConv3d(64, 128, 3, 2, padding=[0, 1, 1], bias=False), bn(128), nn.ReLU(),
            block(128, "res3"), block(128, "res3"),
            SparseConv3d(128, 128, (3, 1, 1), (2, 1, 1), bias=False), bn(128), nn.ReLU())


@BACKBONES.register_module
class PointPillarsScatter(nn.Module):
    def __init__(self, num_input_features=64, norm_cfg=None, name="PointPillarsScatter", **kwargs):
        super().__init__()
        self.name = "PointPillarsScatter"
        self.nchannels = num_input_features

    def forward(self, voxel_features, coords, batch_size, input_shape):
        self.nx = int(input_shape[0])
        self.ny = int(input_shape[1])
        if voxel_features.is_cuda and voxel_features.dtype in (torch.float32, torch.bfloat16):
            # pillars are a one-cell-deep sparse tensor: d3d_dense_scatter (HIP) with D = 1
            idx = torch.stack([coords[:, 0], torch.zeros_like(coords[:, 0]), coords[:, 2], coords[:, 3]], 1)
            st = spconv.SparseConvTensor(voxel_features, idx.int(), [1, self.ny, self.nx], batch_size)
            # the canvas written channels-last by the kernel, like SpMiddleFHD's dense map: the neck's BatchNorm2d then
            # runs on the library's sweeps, its 128 / 256-channel stride-1 convs on the library's implicit GEMM, and
            # MIOpen's own kernels for the rest take NHWC as it is (an NCHW canvas cost the nuScenes PointPillars step
            # 64 layout transposes = 1.5 ms of 17.5, and kept every layer of the neck on MIOpen)
            if getattr(self, "dense_channels_last", True):
                return st.dense_2d(channels_last=True).view(batch_size, self.nchannels, self.ny, self.nx)
            return st.dense().view(batch_size, self.nchannels, self.ny, self.nx)
        canvas = torch.zeros(batch_size, self.nchannels, self.nx * self.ny, dtype=voxel_features.dtype,
                             device=voxel_features.device)
        cell = (coords[:, 2] * self.nx + coords[:, 3]).long()
        canvas[coords[:, 0].long(), :, cell] = voxel_features
        return canvas.view(batch_size, self.nchannels, self.ny, self.nx)
