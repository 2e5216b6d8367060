"""Small nn helpers with the reference's names and call signatures (API surface only; the bodies are this repo's):
  build_norm_layer(cfg, n) -> (name, layer)   det3d/models/utils/norm.py:60-112   (config key "type": BN | BN1d | GN | SyncBN)
  Sequential(.add) with "0", "1", ... child names, Empty                          det3d/models/utils/misc.py:40-113
  get_paddings_indicator(actual_num, max_num)                                      det3d/models/utils/misc.py:180-202
`Sequential.forward` is where the RPN's conv -> BN -> ReLU runs are handed to the library's kernels (dense_stack.py).
"""
import os
from collections import OrderedDict

import torch
from torch import nn


from .syncbn import DistributedSyncBN  # noqa: E402  (reference semantics, det3d/ops/syncbn/syncbn.py)


# config "type" -> (name prefix the reference derives attribute names from, constructor)
norm_cfg = {
    "BN": ("bn", nn.BatchNorm2d),
    "BN1d": ("bn1d", nn.BatchNorm1d),
    "GN": ("gn", nn.GroupNorm),
    "SyncBN": ("bn", DistributedSyncBN),
}


def build_norm_layer(cfg, num_features, postfix=""):
    """(name, module) for a norm config dict: {"type": ..., "eps", "momentum", "requires_grad", "num_groups" (GN)}.
    eps defaults to 1e-5; every other key goes to the constructor."""
    if not isinstance(cfg, dict) or "type" not in cfg:
        raise TypeError("norm config must be a dict with a 'type' key, got %r" % (cfg,))
    kwargs = {k: v for k, v in cfg.items() if k not in ("type", "requires_grad")}
    kwargs.setdefault("eps", 1e-5)
    try:
        prefix, ctor = norm_cfg[cfg["type"]]
    except KeyError:
        raise KeyError("Unrecognized norm type {}".format(cfg["type"])) from None
    if ctor is nn.GroupNorm:
        layer = ctor(num_channels=num_features, **kwargs)  # needs num_groups in the config
    else:
        layer = ctor(num_features, **kwargs)
    layer.requires_grad_(bool(cfg.get("requires_grad", True)))
    return "%s%s" % (prefix, postfix), layer


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
    """nn.Sequential with `.add()`.  Children are registered under "0", "1", ... (or the given names), so state-dict
    keys match the reference (`neck.blocks.0.1.weight`, ...).  Accepts modules, one OrderedDict, and named modules as
    keywords, like the reference's container."""

    def __init__(self, *args, **kwargs):
        super().__init__()
        named = args[0].items() if len(args) == 1 and isinstance(args[0], OrderedDict) else \
            ((str(i), m) for i, m in enumerate(args))
        for name, module in list(named) + list(kwargs.items()):
            self.add(module, name)

    def __len__(self):
        return len(self._modules)

    def __getitem__(self, idx):
        n = len(self)
        if not -n <= idx < n:
            raise IndexError("index {} is out of range".format(idx))
        return list(self._modules.values())[idx % n]

    def add(self, module, name=None):
        name = str(len(self._modules)) if name is None else name
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
            if type(m).__name__ == "DistributedSyncBN" and torch.is_tensor(input) and input.is_cuda and input.dim() == 4:
                # a cross-rank BN outside a conv run (behind a deblock's kernel == stride layer): the HIP sweeps split at
                # its packed all-reduce, the ReLU that follows folded in
                from .syncbn import sync_bn_relu

                relu = i + 1 < len(mods) and type(mods[i + 1]) is nn.ReLU
                y = sync_bn_relu(m, input, relu)
                if y is not None:
                    input = y
                    i += 2 if relu else 1
                    continue
            if type(m) in (nn.Conv2d, nn.ConvTranspose2d) and m.kernel_size == m.stride and m.kernel_size[0] > 1 and \
                    torch.is_tensor(input) and input.is_cuda:
                from . import dense_conv  # the deblocks' kernel == stride layers as 1x1 convolutions (dense_conv.ks_conv)

                if dense_conv.ks_eligible(m, input):
                    input = dense_conv.ks_conv(m, input)
                    i += 1
                    continue
            input = m(input)
            i += 1
        return input


class Empty(nn.Module):
    """placeholder module: returns its positional inputs (one -> itself, none -> None, several -> the tuple)"""

    def __init__(self, *args, **kwargs):
        super().__init__()

    def forward(self, *args, **kwargs):
        return args[0] if len(args) == 1 else (args or None)


def get_paddings_indicator(actual_num, max_num, axis=0):
    """bool mask of the real entries: out[..., j, ...] = j < actual_num[...], the new axis inserted at axis + 1"""
    ramp = torch.arange(max_num, dtype=torch.int32, device=actual_num.device)
    ramp = ramp.view([-1 if d == axis + 1 else 1 for d in range(actual_num.dim() + 1)])
    return ramp < actual_num.int().unsqueeze(axis + 1)
