"""Checkpoint import / export and BatchNorm folding for inference.

  load_state_dict / load_checkpoint / save_checkpoint   det3d/torchie/trainer/checkpoint.py:42-215
      same tolerant matching (unexpected / missing / shape-mismatched keys reported, `num_batches_tracked`
      ignored, a leading "module." stripped, {"state_dict": ...} or a bare OrderedDict accepted).  Module and
      parameter names of det3d_amd match the reference one to one (`backbone.middle_conv.N.*`,
      `neck.blocks.i.j.*`, `bbox_head.tasks.t.conv_{box,cls,dir}.*`, sparse weights `[kD,kH,kW,Cin,Cout]`), so
      released Det3D `.pth` files load unchanged.  URL schemes are not supported (no network here).
  fuse_bn_recursively   the reference only hints at it (tools/dist_test.py:164, commented out): fold every
      eval-mode BatchNorm into the convolution in front of it -- sparse conv + BatchNorm1d in
      SparseSequential / SparseBasicBlock, Conv2d / ConvTranspose2d + BatchNorm2d in the RPN -- so inference
      runs one kernel per layer instead of conv + normalise.
"""
import logging
import os
from collections import OrderedDict

import torch
from torch import nn


def load_state_dict(module, state_dict, strict=False, logger=None):
    unexpected, mismatched = [], []
    own = module.state_dict()
    for name, param in state_dict.items():
        if name not in own:
            unexpected.append(name)
            continue
        if isinstance(param, nn.Parameter):
            param = param.data
        if param.size() != own[name].size():
            mismatched.append((name, tuple(own[name].size()), tuple(param.size())))
            continue
        own[name].copy_(param)
    missing = [k for k in set(own) - set(state_dict) if "num_batches_tracked" not in k]
    msg = []
    if unexpected:
        msg.append("unexpected key in source state_dict: %s" % ", ".join(unexpected))
    if missing:
        msg.append("missing keys in source state_dict: %s" % ", ".join(sorted(missing)))
    if mismatched:
        msg.append("these keys have mismatched shape:\n" + "\n".join(
            "  %s: expected %s, loaded %s" % m for m in mismatched))
    if msg:
        text = "The model and loaded state dict do not match exactly\n" + "\n".join(msg)
        if strict:
            raise RuntimeError(text)
        (logger or logging.getLogger("det3d_amd.checkpoint")).warning(text)
    return dict(unexpected=unexpected, missing=sorted(missing), mismatched=mismatched)


def load_checkpoint(model, filename, map_location=None, strict=False, logger=None):
    if filename.startswith(("modelzoo://", "torchvision://", "open-mmlab://", "http://", "https://")):
        raise IOError("%s: URL checkpoints are not supported (offline build); download the file first" % filename)
    if not os.path.isfile(filename):
        raise IOError("{} is not a checkpoint file".format(filename))
    checkpoint = torch.load(filename, map_location=map_location, weights_only=False)
    if isinstance(checkpoint, OrderedDict):
        state_dict = checkpoint
    elif isinstance(checkpoint, dict) and "state_dict" in checkpoint:
        state_dict = checkpoint["state_dict"]
    else:
        raise RuntimeError("No state_dict found in checkpoint file {}".format(filename))
    if state_dict and next(iter(state_dict)).startswith("module."):
        state_dict = OrderedDict((k[7:], v) for k, v in state_dict.items())
    target = model.module if hasattr(model, "module") else model
    load_state_dict(target, state_dict, strict, logger)
    return checkpoint


def save_checkpoint(model, filename, optimizer=None, meta=None):
    if meta is None:
        meta = {}
    elif not isinstance(meta, dict):
        raise TypeError("meta must be a dict or None, but got {}".format(type(meta)))
    d = os.path.dirname(filename)
    if d:
        os.makedirs(d, exist_ok=True)
    if hasattr(model, "module"):
        model = model.module
    checkpoint = {"meta": meta, "state_dict": OrderedDict((k, v.cpu()) for k, v in model.state_dict().items())}
    if optimizer is not None:
        checkpoint["optimizer"] = optimizer.state_dict()
    torch.save(checkpoint, filename)


# ----------------------------------------------------------------------------- BN folding
def _bn_scale_shift(bn):
    invstd = torch.rsqrt(bn.running_var.float() + bn.eps)
    gamma = bn.weight.float() if bn.weight is not None else torch.ones_like(invstd)
    beta = bn.bias.float() if bn.bias is not None else torch.zeros_like(invstd)
    scale = gamma * invstd
    return scale, beta - bn.running_mean.float() * scale


def _foldable(bn):
    return isinstance(bn, nn.modules.batchnorm._BatchNorm) and bn.track_running_stats and bn.running_mean is not None


@torch.no_grad()
def fold_bn_into_conv(conv, bn):
    """y = BN(conv(x)) in eval mode  ==  conv'(x) with W' = W * scale[cout], b' = b * scale + shift."""
    from . import spconv

    scale, shift = _bn_scale_shift(bn)
    w = conv.weight
    if isinstance(conv, spconv.SparseConvolution):
        w.mul_(scale.to(w.dtype))                                   # [kD,kH,kW,Cin,Cout]: Cout is last
    elif isinstance(conv, nn.ConvTranspose2d):
        if conv.groups != 1:
            raise ValueError("fold_bn_into_conv: grouped ConvTranspose2d is not supported")
        w.mul_(scale.to(w.dtype).view(1, -1, 1, 1))                 # [Cin, Cout, kH, kW]
    elif isinstance(conv, nn.Conv2d):
        w.mul_(scale.to(w.dtype).view(-1, 1, 1, 1))                 # [Cout, Cin/g, kH, kW]
    else:
        raise TypeError("fold_bn_into_conv: unsupported layer %s" % type(conv).__name__)
    bias = conv.bias.float() * scale + shift if conv.bias is not None else shift
    conv.bias = nn.Parameter(bias.to(w.dtype), requires_grad=w.requires_grad)
    return conv


def _fold_sequence(container, names):
    """fold (conv, bn) neighbours among the children `names` of `container`; BN -> Identity"""
    from . import spconv

    folded = 0
    prev_name = None
    for name in names:
        m = container._modules[name]
        if prev_name is not None and _foldable(m):
            prev = container._modules[prev_name]
            if isinstance(prev, (spconv.SparseConvolution, nn.Conv2d, nn.ConvTranspose2d)):
                fold_bn_into_conv(prev, m)
                container._modules[name] = nn.Identity()
                folded += 1
                prev_name = None
                continue
        prev_name = name
    return folded


def fuse_bn_recursively(model):
    """In place; the model must be in eval mode afterwards (folded layers have no batch statistics).
    Returns the number of BatchNorm layers folded away."""
    from . import backbones

    model.eval()
    folded = 0
    for mod in model.modules():
        if isinstance(mod, backbones.SparseBasicBlock):
            for conv, bn in (("conv1", "bn1"), ("conv2", "bn2")):
                if _foldable(getattr(mod, bn)):
                    fold_bn_into_conv(getattr(mod, conv), getattr(mod, bn))
                    setattr(mod, bn, nn.Identity())
                    folded += 1
        elif not isinstance(mod, nn.ModuleList) and len(mod._modules) > 1 and all(k.isdigit() for k in mod._modules):
            folded += _fold_sequence(mod, list(mod._modules))
    return folded
