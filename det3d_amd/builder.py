"""Builders reachable from the reference configs and detectors:
  build_box_coder        det3d/builder.py:399-433 (called INSIDE config files)
  get_downsample_factor  det3d/utils/config_tool.py:39-48 (idem)
  build_reader/backbone/neck/head/loss/detector   det3d/models/builder.py:15-53
"""
import numpy as np
from torch import nn

from . import box_ops
from .registry import (BACKBONES, DETECTORS, HEADS, LOSSES, NECKS, READERS, build_from_cfg)


class GroundBox3dCoderTorch:
    """det3d/core/bbox/box_coders.py:32-62,100-109"""

    def __init__(self, linear_dim=False, vec_encode=False, n_dim=7, norm_velo=False):
        self.linear_dim = linear_dim
        self.vec_encode = vec_encode
        self.norm_velo = norm_velo
        self.n_dim = n_dim

    @property
    def code_size(self):
        return self.n_dim + 1 if self.vec_encode else self.n_dim

    def encode_torch(self, boxes, anchors):
        return box_ops.second_box_encode(boxes, anchors, self.vec_encode, self.linear_dim)

    def decode_torch(self, boxes, anchors):
        # like the reference, smooth_dim is passed in the bin_loss slot's successor:
        # second_box_decode(boxes, anchors, vec_encode, linear_dim) -> 4th positional is
        # `bin_loss`, so linear_dim is effectively ignored there (box_coders.py:106-109)
        return box_ops.second_box_decode(boxes, anchors, self.vec_encode, self.linear_dim)

    def encode(self, boxes, anchors):
        import torch

        return self.encode_torch(torch.as_tensor(boxes), torch.as_tensor(anchors)).numpy()

    def decode(self, encodings, anchors):
        import torch

        return self.decode_torch(torch.as_tensor(encodings), torch.as_tensor(anchors)).numpy()


def build_box_coder(box_coder_config):
    kind = box_coder_config["type"]
    cfg = box_coder_config
    if kind == "ground_box3d_coder":
        return GroundBox3dCoderTorch(cfg["linear_dim"], cfg["encode_angle_vector"],
                                     n_dim=cfg.get("n_dim", 9), norm_velo=cfg.get("norm_velo", False))
    raise ValueError("unknown box_coder type")


def get_downsample_factor(model_config):
    neck_cfg = model_config["neck"]
    f = np.prod(neck_cfg.get("ds_layer_strides", [1]))
    if len(neck_cfg.get("us_layer_strides", [])) > 0:
        f /= neck_cfg.get("us_layer_strides", [])[-1]
    f *= model_config["backbone"]["ds_factor"]
    f = int(f)
    assert f > 0
    return f


def build(cfg, registry, default_args=None):
    if isinstance(cfg, list):
        return nn.Sequential(*[build_from_cfg(c, registry, default_args) for c in cfg])
    return build_from_cfg(cfg, registry, default_args)


def build_reader(cfg):
    return build(cfg, READERS)


def build_backbone(cfg):
    return build(cfg, BACKBONES)


def build_neck(cfg):
    return build(cfg, NECKS)


def build_head(cfg):
    return build(cfg, HEADS)


def build_loss(cfg):
    return build(cfg, LOSSES)


def build_detector(cfg, train_cfg=None, test_cfg=None):
    from . import detectors  # noqa: F401  (registers VoxelNet / PointPillars and their parts)

    return build(cfg, DETECTORS, dict(train_cfg=train_cfg, test_cfg=test_cfg))
