"""RPN neck, det3d/models/necks/rpn.py:22-159: per block ZeroPad2d(1) + Conv3x3(stride) + BN
+ ReLU then n x (Conv3x3 + BN + ReLU); deblocks are ConvTranspose2d (stride > 1) or a
strided Conv2d (stride < 1); outputs concatenated on C.  Dense convs go to MIOpen/rocBLAS
through PyTorch-ROCm (bf16 under autocast)."""
import logging

import numpy as np
import torch
from torch import nn

from .nn_utils import Sequential, build_norm_layer
from .registry import NECKS


@NECKS.register_module
class RPN(nn.Module):
    def __init__(self, layer_nums, ds_layer_strides, ds_num_filters, us_layer_strides, us_num_filters,
                 num_input_features, norm_cfg=None, name="rpn", logger=None, **kwargs):
        super().__init__()
        self._layer_strides = ds_layer_strides
        self._num_filters = ds_num_filters
        self._layer_nums = layer_nums
        self._upsample_strides = us_layer_strides
        self._num_upsample_filters = us_num_filters
        self._num_input_features = num_input_features
        if norm_cfg is None:
            norm_cfg = dict(type="BN", eps=1e-3, momentum=0.01)
        self._norm_cfg = norm_cfg
        assert len(self._layer_strides) == len(self._layer_nums)
        assert len(self._num_filters) == len(self._layer_nums)
        assert len(self._num_upsample_filters) == len(self._upsample_strides)
        self._upsample_start_idx = len(self._layer_nums) - len(self._upsample_strides)
        ratios = [self._upsample_strides[i] / np.prod(self._layer_strides[: i + self._upsample_start_idx + 1])
                  for i in range(len(self._upsample_strides))]
        assert all(r == ratios[0] for r in ratios)

        in_filters = [self._num_input_features, *self._num_filters[:-1]]
        blocks, deblocks = [], []
        for i, layer_num in enumerate(self._layer_nums):
            block, planes = self._make_layer(in_filters[i], self._num_filters[i], layer_num,
                                             stride=self._layer_strides[i])
            blocks.append(block)
            j = i - self._upsample_start_idx
            if j >= 0:
                stride = self._upsample_strides[j]
                out_c = self._num_upsample_filters[j]
                if stride > 1:
                    up = nn.ConvTranspose2d(planes, out_c, stride, stride=stride, bias=False)
                else:
                    s = int(np.round(1 / stride).astype(np.int64))
                    up = nn.Conv2d(planes, out_c, s, stride=s, bias=False)
                deblocks.append(Sequential(up, build_norm_layer(self._norm_cfg, out_c)[1], nn.ReLU()))
        self.blocks = nn.ModuleList(blocks)
        self.deblocks = nn.ModuleList(deblocks)
        (logger or logging.getLogger("RPN")).info("Finish RPN Initialization")

    @property
    def downsample_factor(self):
        factor = np.prod(self._layer_strides)
        if len(self._upsample_strides) > 0:
            factor /= self._upsample_strides[-1]
        return factor

    def _make_layer(self, inplanes, planes, num_blocks, stride=1):
        block = Sequential(nn.ZeroPad2d(1), nn.Conv2d(inplanes, planes, 3, stride=stride, bias=False),
                           build_norm_layer(self._norm_cfg, planes)[1], nn.ReLU())
        for _ in range(num_blocks):
            block.add(nn.Conv2d(planes, planes, 3, padding=1, bias=False))
            block.add(build_norm_layer(self._norm_cfg, planes)[1])
            block.add(nn.ReLU())
        return block, planes

    def init_weights(self):
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.xavier_uniform_(m.weight)

    def forward(self, x):
        ups = []
        for i, blk in enumerate(self.blocks):
            x = blk(x)
            if i - self._upsample_start_idx >= 0:
                ups.append(self.deblocks[i - self._upsample_start_idx](x))
        if len(ups) > 0:
            x = torch.cat(ups, dim=1)
        return x
