"""TEST INFRASTRUCTURE -- a `spconv`-shaped module over oracle/spconv_ref.py, so that the
REFERENCE'S OWN backbone files (det3d/models/backbones/scn.py) can be executed in the build
container, where the real (un-vendored, un-pinned) spconv package is absent.

tests/golden/make_golden.py installs this module as `sys.modules["spconv"]` before importing the
reference: module composition (layer list, residual add, bias flags, BN placement, dense() + view)
is then the reference's code, while the sparse-conv arithmetic is oracle/spconv_ref.py, which is
itself pinned to torch.nn.functional.conv3d (tests/test_oracle_spconv_cpu.py).  Only the API
surface scn.py touches exists here (seam B3):
  SparseConvTensor(features, indices, spatial_shape, batch_size) with mutable .features, .dense()
  SparseSequential, SparseModule, SubMConv3d, SparseConv3d
Never imported by the product.
"""
from collections import OrderedDict

import numpy as np
import torch
from torch import nn

from . import spconv_ref as R


class SparseConvTensor:
    def __init__(self, features, indices, spatial_shape, batch_size, grid=None):
        self.features = features
        self.indices = indices
        self.spatial_shape = [int(s) for s in spatial_shape]
        self.batch_size = int(batch_size)
        self.indice_dict = {}
        self.grid = grid

    def dense(self, channels_first=True):
        out = R.dense(self.features, self.indices.cpu().numpy(), self.batch_size, self.spatial_shape)
        return out if channels_first else out.permute(0, 2, 3, 4, 1).contiguous()


class SparseModule(nn.Module):
    pass


class _Conv(SparseModule):
    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, dilation=1, groups=1,
                 bias=True, indice_key=None, subm=False, **kw):
        super().__init__()
        t = lambda v: [int(q) for q in v] if isinstance(v, (list, tuple)) else [int(v)] * 3
        self.kernel_size, self.stride, self.padding = t(kernel_size), t(stride), t(padding)
        self.subm = subm
        self.indice_key = indice_key
        self.weight = nn.Parameter(torch.empty(*self.kernel_size, in_channels, out_channels))
        nn.init.kaiming_uniform_(self.weight, a=5 ** 0.5)
        if bias:
            self.bias = nn.Parameter(torch.zeros(out_channels))
        else:
            self.register_parameter("bias", None)

    def forward(self, x):
        key = self.indice_key if self.indice_key is not None else ("_anon", id(self))
        if key not in x.indice_dict:
            stride = [1, 1, 1] if self.subm else self.stride
            padding = [0, 0, 0] if self.subm else self.padding
            x.indice_dict[key] = R.get_indice_pairs(x.indices.cpu().numpy(), x.batch_size, x.spatial_shape,
                                                    self.kernel_size, stride, padding, self.subm)
        out_idx, out_shape, pairs = x.indice_dict[key]
        feats = R.indice_conv(x.features, self.weight, pairs, len(out_idx), self.bias)
        out = SparseConvTensor(feats, torch.from_numpy(np.ascontiguousarray(out_idx)), out_shape, x.batch_size)
        out.indice_dict = x.indice_dict
        return out


class SubMConv3d(_Conv):
    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, dilation=1, groups=1,
                 bias=True, indice_key=None, **kw):
        super().__init__(in_channels, out_channels, kernel_size, 1, 0, dilation, groups, bias, indice_key, subm=True)


class SparseConv3d(_Conv):
    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, dilation=1, groups=1,
                 bias=True, indice_key=None, **kw):
        super().__init__(in_channels, out_channels, kernel_size, stride, padding, dilation, groups, bias,
                         indice_key, subm=False)


class SparseSequential(SparseModule):
    def __init__(self, *args, **kwargs):
        super().__init__()
        if len(args) == 1 and isinstance(args[0], OrderedDict):
            for k, m in args[0].items():
                self.add_module(k, m)
        else:
            for i, m in enumerate(args):
                self.add_module(str(i), m)
        for k, m in kwargs.items():
            self.add_module(k, m)

    def forward(self, x):
        for m in self._modules.values():
            if isinstance(m, SparseModule):
                x = m(x)
            elif isinstance(x, SparseConvTensor):
                if x.indices.shape[0] != 0:
                    x.features = m(x.features)
            else:
                x = m(x)
        return x
