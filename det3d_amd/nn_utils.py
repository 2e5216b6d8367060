"""Small nn helpers the reference model files lean on:
  build_norm_layer / norm table   det3d/models/utils/norm.py:60-112
  Sequential (.add), Empty        det3d/models/utils/misc.py:40-113
  get_paddings_indicator          det3d/models/utils/misc.py:180-202
"""
import os
from collections import OrderedDict

import torch
from torch import nn


from .syncbn import DistributedSyncBN  # noqa: E402  (reference semantics, det3d/ops/syncbn/syncbn.py)


norm_cfg = {
    "BN": ("bn", nn.BatchNorm2d),
    "BN1d": ("bn1d", nn.BatchNorm1d),
    "GN": ("gn", nn.GroupNorm),
    "SyncBN": ("bn", DistributedSyncBN),
}


def build_norm_layer(cfg, num_features, postfix=""):
    assert isinstance(cfg, dict) and "type" in cfg
    cfg_ = dict(cfg)
    layer_type = cfg_.pop("type")
    if layer_type not in norm_cfg:
        raise KeyError("Unrecognized norm type {}".format(layer_type))
    abbr, norm_layer = norm_cfg[layer_type]
    assert isinstance(postfix, (int, str))
    name = abbr + str(postfix)
    requires_grad = cfg_.pop("requires_grad", True)
    cfg_.setdefault("eps", 1e-5)
    if layer_type != "GN":
        layer = norm_layer(num_features, **cfg_)
    else:
        assert "num_groups" in cfg_
        layer = norm_layer(num_channels=num_features, **cfg_)
    for p in layer.parameters():
        p.requires_grad = requires_grad
    return name, layer


# The RPN's Conv2d -> BatchNorm2d -> ReLU triples (channels-last maps) go through the fused HIP BatchNorm+ReLU
# sweeps of the sparse layers (d3d_bn.hip): 4 launches per layer and direction pair instead of MIOpen's 3 + 3
# BatchNorm kernels plus the separate clamp / threshold-backward kernels.  Round 1 had this off (MIOpen was faster
# on the 36 MB KITTI maps, 438 vs 398 frames/s); with the large-input variants of the reduction sweeps it is
# 0.3 ms of GPU time per step ahead (profiles/round2_kernel_stats_v4.txt).  D3D_FUSE_BN2D=0 turns it off.
FUSE_BN2D_RELU = os.environ.get("D3D_FUSE_BN2D", "1") == "1"


def _pad_folds_into(pad, conv):
    """ZeroPad2d(p) followed by a zero-padded-by-nothing Conv2d == that Conv2d with padding p"""
    if type(conv) is not nn.Conv2d or conv.padding_mode != "zeros":
        return False
    p = pad.padding
    p = (p,) * 4 if isinstance(p, int) else tuple(p)
    return len(set(p)) == 1 and p[0] >= 0 and tuple(conv.padding) == (0, 0)


class Sequential(nn.Module):
    """nn.Sequential with `.add()`; children are named "0", "1", ... so state-dict keys
    match the reference (`neck.blocks.0.1.weight`, ...)."""

    def __init__(self, *args, **kwargs):
        super().__init__()
        if len(args) == 1 and isinstance(args[0], OrderedDict):
            for key, module in args[0].items():
                self.add_module(key, module)
        else:
            for idx, module in enumerate(args):
                self.add_module(str(idx), module)
        for name, module in kwargs.items():
            if name in self._modules:
                raise ValueError("name exists.")
            self.add_module(name, module)

    def __getitem__(self, idx):
        if not (-len(self) <= idx < len(self)):
            raise IndexError("index {} is out of range".format(idx))
        if idx < 0:
            idx += len(self)
        return list(self._modules.values())[idx]

    def __len__(self):
        return len(self._modules)

    def add(self, module, name=None):
        if name is None:
            name = str(len(self._modules))
            if name in self._modules:
                raise KeyError("name exists")
        self.add_module(name, module)

    def forward(self, input):
        mods = list(self._modules.values())
        i = 0
        while i < len(mods):
            m = mods[i]
            if type(m) in (nn.ZeroPad2d, nn.Conv2d) and torch.is_tensor(input) and input.is_cuda:
                # runs of [pad] conv -> BN [-> ReLU] on the library's own MFMA conv kernels, one autograd node
                from . import dense_stack

                layers, nxt = dense_stack.collect(mods, i, input)
                if layers:
                    input = dense_stack.run(layers, input)
                    i = nxt
                    continue
            if type(m) is nn.ZeroPad2d and i + 1 < len(mods) and _pad_folds_into(m, mods[i + 1]) and \
                    torch.is_tensor(input) and input.is_cuda:
                # rpn.py:64-68 pads with a separate ZeroPad2d(1) and convolves without padding: the same sums as a
                # padding=1 convolution, minus a 36 MB padded copy of the map (and its backward) per step
                conv = mods[i + 1]
                input = nn.functional.conv2d(input, conv.weight, conv.bias, conv.stride, (m.padding if isinstance(m.padding, int) else m.padding[0]), conv.dilation,
                                             conv.groups)
                i += 2
                continue
            if FUSE_BN2D_RELU and type(m) is nn.BatchNorm2d and torch.is_tensor(input) and input.is_cuda:
                from . import spconv  # the fused [N, C] BatchNorm(+ReLU) kernels (d3d_bn_relu_*)

                if spconv.fusable_bn2d(m, input):
                    relu = i + 1 < len(mods) and type(mods[i + 1]) is nn.ReLU
                    input = spconv.bn_relu_2d(m, input, relu)
                    i += 2 if relu else 1
                    continue
            input = m(input)
            i += 1
        return input


class Empty(nn.Module):
    def __init__(self, *args, **kwargs):
        super().__init__()

    def forward(self, *args, **kwargs):
        if len(args) == 1:
            return args[0]
        if len(args) == 0:
            return None
        return args


def get_paddings_indicator(actual_num, max_num, axis=0):
    actual_num = torch.unsqueeze(actual_num, axis + 1)
    shape = [1] * len(actual_num.shape)
    shape[axis + 1] = -1
    rng = torch.arange(max_num, dtype=torch.int, device=actual_num.device).view(shape)
    return actual_num.int() > rng
