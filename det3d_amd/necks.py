"""RPN neck (registry name `RPN`), behaviour of det3d/models/necks/rpn.py:22-159.

Stage i:  ZeroPad2d(1) + Conv3x3(stride s_i) + BN + ReLU, then n_i x (Conv3x3(pad 1) + BN + ReLU),
all bias-free.  The last len(us_layer_strides) stages each feed a "deblock": ConvTranspose2d
(kernel = stride = u) when u > 1, otherwise a Conv2d with kernel = stride = round(1/u); + BN + ReLU.
Deblock outputs are concatenated along C.  Child modules are numbered exactly like the reference
(`blocks.<i>.<j>`, `deblocks.<i>.<j>`) so checkpoints load.  On the GPU the conv -> BN -> ReLU runs of the
Sequentials execute on the library's own kernels (det3d_amd/dense_stack.py: csrc/d3d_conv2d.hip); what those do not
cover (kernel == stride deblocks of other shapes, CPU tensors) goes through PyTorch-ROCm / MIOpen.
"""
import logging
from functools import reduce
from operator import mul

import torch
from torch import nn

from .nn_utils import Sequential, build_norm_layer
from .registry import NECKS

_DEFAULT_NORM = dict(type="BN", eps=1e-3, momentum=0.01)


def _prod(xs):
    return reduce(mul, xs, 1)


def _run_joined(stage, deblock, x):
    """stage and deblock as one run of det3d_amd.dense_stack, or None when the library's kernels do not cover both"""
    if not (torch.is_tensor(x) and x.is_cuda) or not isinstance(stage, Sequential) or not isinstance(deblock, Sequential):
        return None
    from . import dense_stack

    mods = list(stage._modules.values()) + list(deblock._modules.values())
    layers, nxt = dense_stack.collect(mods, 0, x)
    if not layers or nxt != len(mods):
        return None
    return dense_stack.run(layers, x)


@NECKS.register_module
class RPN(nn.Module):
    def __init__(self, layer_nums, ds_layer_strides, ds_num_filters, us_layer_strides, us_num_filters,
                 num_input_features, norm_cfg=None, name="rpn", logger=None, **kwargs):
        super().__init__()
        n_stage, n_up = len(layer_nums), len(us_layer_strides)
        if not (len(ds_layer_strides) == len(ds_num_filters) == n_stage):
            raise AssertionError("layer_nums, ds_layer_strides and ds_num_filters must have equal length")
        if len(us_num_filters) != n_up:
            raise AssertionError("us_layer_strides and us_num_filters must have equal length")
        self._norm_cfg = dict(norm_cfg) if norm_cfg is not None else dict(_DEFAULT_NORM)
        self._layer_nums = list(layer_nums)
        self._layer_strides = list(ds_layer_strides)
        self._num_filters = list(ds_num_filters)
        self._upsample_strides = list(us_layer_strides)
        self._num_upsample_filters = list(us_num_filters)
        self._num_input_features = num_input_features
        self._upsample_start_idx = first_up = n_stage - n_up

        # every deblock must bring its stage back to the same resolution
        scale = [us_layer_strides[j] / _prod(ds_layer_strides[: j + first_up + 1]) for j in range(n_up)]
        if any(v != scale[0] for v in scale):
            raise AssertionError("inconsistent up/down-sampling strides: %s" % scale)

        widths_in = [num_input_features] + list(ds_num_filters[:-1])
        self.blocks = nn.ModuleList(
            self._make_layer(widths_in[i], ds_num_filters[i], layer_nums[i], stride=ds_layer_strides[i])[0]
            for i in range(n_stage))
        self.deblocks = nn.ModuleList(
            self._make_deblock(ds_num_filters[first_up + j], us_num_filters[j], us_layer_strides[j])
            for j in range(n_up))
        (logger if logger is not None else logging.getLogger("RPN")).info("Finish RPN Initialization")

    def _norm(self, channels):
        return build_norm_layer(self._norm_cfg, channels)[1]

    def _make_layer(self, inplanes, planes, num_blocks, stride=1):
        seq = Sequential(nn.ZeroPad2d(1), nn.Conv2d(inplanes, planes, 3, stride=stride, bias=False),
                         self._norm(planes), nn.ReLU())
        for _ in range(num_blocks):
            for layer in (nn.Conv2d(planes, planes, 3, padding=1, bias=False), self._norm(planes), nn.ReLU()):
                seq.add(layer)
        return seq, planes

    def _make_deblock(self, cin, cout, stride):
        if stride > 1:
            up = nn.ConvTranspose2d(cin, cout, stride, stride=stride, bias=False)
        else:
            k = int(round(1.0 / stride))
            up = nn.Conv2d(cin, cout, k, stride=k, bias=False)
        return Sequential(up, self._norm(cout), nn.ReLU())

    @property
    def downsample_factor(self):
        f = _prod(self._layer_strides)
        return f / self._upsample_strides[-1] if self._upsample_strides else f

    def init_weights(self):
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.xavier_uniform_(m.weight)

    def forward(self, x):
        outs = []
        for i, stage in enumerate(self.blocks):
            j = i - self._upsample_start_idx
            if j >= 0 and i + 1 == len(self.blocks):
                # the last stage's map has one reader, its deblock: where that is a plain 1x1 conv -> BN -> ReLU (stride-1
                # "upsampling", rpn.py:85-93) the two Sequentials run as ONE conv stack -- the stage's last BatchNorm
                # writes the operand planes the deblock's conv reads (no f32 copy of the map, no split sweep over it:
                # 25 us of the KITTI step) and the deblock's weight gradient joins the second stream
                y = _run_joined(stage, self.deblocks[j], x)
                if y is not None:
                    outs.append(y)
                    break
            x = stage(x)
            if j >= 0:
                outs.append(self.deblocks[j](x))
        if len(outs) == 1:
            return outs[0]  # (the reference concatenates the single map with itself to no effect: a 72 MB copy)
        return torch.cat(outs, dim=1) if outs else x
