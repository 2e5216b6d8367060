"""MultiGroupHead, det3d/models/bbox_heads/mg_head.py:
  Head.forward            :221-230   1x1 convs, NHWC permute
  MultiGroupHead.__init__ :385-500   per-task heads, 2 anchors (rotations) per class
  prepare_loss_weights    :535-578
  loss                    :580-695   focal + smooth-L1(sin-difference) + direction softmax
  predict                 :697-803   decode, per-task detections, merge with label offsets
  get_task_detections     :805-1085  sigmoid, score threshold, NMS, direction flip, range mask
Same constructor kwargs, attribute names (`tasks.<t>.conv_{box,cls,dir}`) and output dict
keys.  NMS goes to the HIP kernel (box_ops.rotate_nms); the multi-class NMS branch of the
reference references undefined attributes (:911-913) and is not reachable from any shipped
config, so it raises here.
"""
import logging
import os
import weakref
from collections import defaultdict

import numpy as np
import torch
from torch import nn

from . import _lib, box_ops
from . import losses  # noqa: F401  (registers the loss classes)
from .builder import build_loss
from .registry import HEADS


# MultiGroupHead.loss through d3d_head_loss (one fused forward+gradient kernel per task group instead of
# ~200 element-wise launches).  D3D_FUSE_LOSS=0 keeps the PyTorch-op formulation (also the CPU path).
FUSE_LOSS = os.environ.get("D3D_FUSE_LOSS", "1") != "0"
# MultiGroupHead.predict through d3d_decode_filter_topk + d3d_rotate_nms_batched (5 launches per task group for the
# whole batch, one host sync per batch).  D3D_FUSE_PREDICT=0 keeps the per-frame PyTorch-op path.
FUSE_PREDICT = os.environ.get("D3D_FUSE_PREDICT", "1") != "0"


class _HeadLossFn(torch.autograd.Function):
    """loss of one task group; returns the [D3D_LOSS_SCALARS] vector (element 0 is the loss)."""

    @staticmethod
    @_lib.ctx_forward
    def forward(ctx, box_preds, cls_preds, dir_preds, labels, reg_targets, anchors, cfg):
        from . import _lib

        L = _lib.lib()
        dev = box_preds.device
        dt = box_preds.dtype
        cls_preds = cls_preds.to(dt).contiguous()
        box_preds = box_preds.contiguous()
        if dir_preds is not None:
            dir_preds = dir_preds.to(dt).contiguous()
        b, a, code = box_preds.shape
        nc = cls_preds.shape[-1]
        g_box, g_cls = torch.empty_like(box_preds), torch.empty_like(cls_preds)
        g_dir = torch.empty_like(dir_preds) if dir_preds is not None else None
        scalars = torch.empty((_lib.D3D_LOSS_SCALARS,), dtype=torch.float32, device=dev)
        wsb = L.d3d_head_loss_workspace_bytes(b)
        ws = torch.empty((wsb,), dtype=torch.uint8, device=dev)
        per_batch = anchors is not None and anchors.dim() == 3
        rc = L.d3d_head_loss(
            box_preds.data_ptr(), cls_preds.data_ptr(), _lib.ptr(dir_preds),
            _lib.D3D_BF16 if dt == torch.bfloat16 else _lib.D3D_F32, labels.data_ptr(), reg_targets.data_ptr(),
            _lib.ptr(anchors), int(per_batch), b, a, code, nc, anchors.shape[-1] if anchors is not None else 0,
            int(cfg["sin_diff"]), float(cfg["dir_offset"]), float(cfg["pos_w"]), float(cfg["neg_w"]),
            float(cfg["alpha"]), float(cfg["gamma"]), float(cfg["sigma"]), float(cfg["w_loc"]),
            float(cfg["w_cls"]), float(cfg["w_dir"]), g_box.data_ptr(), g_cls.data_ptr(), _lib.ptr(g_dir),
            scalars.data_ptr(), ws.data_ptr(), wsb, _lib.stream_ptr(dev))
        _lib.check(rc, "d3d_head_loss")
        ctx.save_for_backward(g_box, g_cls, *([g_dir] if g_dir is not None else []))
        ctx.has_dir = g_dir is not None
        # the loss also as an output of its own (a 0-dim view of element 0): a caller that differentiates THAT instead of
        # `scalars[0]` spares autograd the select-backward pair (zero fill + copy of an 18-float vector) every step
        ctx.set_materialize_grads(False)
        return scalars, scalars[0]

    @staticmethod
    @_lib.ctx_backward
    def backward(ctx, gs, gl):
        saved = ctx.saved_tensors
        # only the loss is differentiable (element 0 of the vector / the second output); the rest are logging sums
        g = gl if gs is None else (gs[0] if gl is None else gs[0] + gl)
        if g is None:
            return (None,) * 7
        # one multi-tensor launch for the two or three prediction gradients (they share a dtype: see forward)
        scaled = torch._foreach_mul(list(saved), g.to(saved[0].dtype))
        return scaled[0], scaled[1], (scaled[2] if ctx.has_dir else None), None, None, None, None


def _eager_task_loss(head, preds, labels, reg_targets, anchors, num_class, batch):
    """One task group's loss in PyTorch ops -- the CPU plumbing path (BASELINE configs[0]) and the unfused side of
    tests/test_loss_gpu.py; on the GPU `d3d_head_loss` computes the same quantities in two launches.
    Definitions (det3d/models/bbox_heads/mg_head.py:562-668): focal classification loss on one-hot targets without
    the background column, smooth-L1 on the box code with the yaw residual replaced by sin(a - b) = sin a cos b -
    cos a sin b split over prediction and target, optional direction classifier on (yaw_gt - offset) mod 2 pi > 0,
    each summed and divided by the batch size."""
    box = preds["box_preds"].float().reshape(batch, -1, head.box_n_dim - 2 if head.bev_only else head.box_n_dim)
    ncol = num_class if head.encode_background_as_zeros else num_class + 1
    cls = preds["cls_preds"].float().reshape(batch, -1, ncol)
    cls_w, reg_w, cared = head.prepare_loss_weights(labels, loss_norm=head.loss_norm, dtype=torch.float32)
    hot = torch.nn.functional.one_hot((labels * cared.type_as(labels)).long(), num_class + 1).to(box.dtype)
    if head.encode_background_as_zeros:
        hot = hot[..., 1:]
    tgt = reg_targets
    if head.encode_rad_error_by_sin:
        a, b = box[..., -1:], tgt[..., -1:]
        box = torch.cat([box[..., :-1], a.sin() * b.cos()], -1)
        tgt = torch.cat([tgt[..., :-1], a.cos() * b.sin()], -1)
    loc = head.loss_reg(box, tgt, weights=reg_w)
    cl = head.loss_cls(cls, hot, weights=cls_w)
    loc_red = loc.sum() / batch * head.loss_reg._loss_weight
    cls_red = cl.sum() / batch * head.loss_cls._loss_weight
    if cl.dim() == 2 or cl.shape[-1] == 1:  # one column: split by the label of the anchor
        flat = cl.reshape(batch, -1)
        pos, neg = (flat * (labels > 0)).sum() / batch, (flat * (labels == 0)).sum() / batch
    else:
        pos, neg = cl[..., 1:].sum() / batch, cl[..., 0].sum() / batch
    total = loc_red + cls_red
    dir_loss = None
    if head.use_direction_classifier:
        yaw = reg_targets[..., -1] + anchors.reshape(batch, -1, anchors.shape[-1])[..., -1] - head.direction_offset
        yaw = yaw - torch.floor(yaw / (2 * np.pi) + 0.5) * (2 * np.pi)
        dir_t = torch.nn.functional.one_hot((yaw > 0).long(), 2).to(anchors.dtype)
        w = (labels > 0).type_as(box)
        w = w / w.sum(-1, keepdim=True).clamp(min=1.0)
        logits = preds["dir_cls_preds"].float().reshape(batch, -1, 2)
        dir_loss = head.loss_aux(logits, dir_t, weights=w).sum() / batch
        total = total + dir_loss * head.loss_aux._loss_weight
    # logged scalars stay on the device (the reference reads ~10 of them to the host per task group, :657-666)
    return {"loss": total,
            "cls_pos_loss": (pos / head.loss_norm["pos_cls_weight"]).detach(),
            "cls_neg_loss": (neg / head.loss_norm["neg_cls_weight"]).detach(),
            "dir_loss_reduced": dir_loss.detach() if dir_loss is not None else torch.tensor(0),
            "cls_loss_reduced": cls_red.detach().mean(), "loc_loss_reduced": loc_red.detach().mean(),
            "loc_loss_elem": [(loc[:, :, i].sum() / batch).detach() for i in range(loc.shape[-1])],
            "num_pos": (labels > 0)[0].sum(), "num_neg": (labels == 0)[0].sum()}


FUSE_HEAD_CONV = os.environ.get("D3D_FUSE_HEAD_CONV", "1") != "0"


def _head_conv_ok(x, convs):
    """the configuration d3d_head_conv_* implements: a CUDA channels-last f32 / bf16 map of 128 channels, plain 1x1
    convolutions with f32 parameters, at most 32 output channels in total"""
    if not (FUSE_HEAD_CONV and torch.is_tensor(x) and x.is_cuda and x.dim() == 4):
        return False
    if x.dtype == torch.float32:
        if torch.is_autocast_enabled("cuda"):
            return False
    elif x.dtype != torch.bfloat16:
        return False
    if x.shape[1] == 1 or not x.is_contiguous(memory_format=torch.channels_last) or x.data_ptr() % 16:
        return False
    for c in convs:
        if type(c) is not nn.Conv2d or c.kernel_size != (1, 1) or c.stride != (1, 1) or c.padding != (0, 0) or \
                c.groups != 1 or c.weight.dtype != torch.float32 or (c.bias is not None and c.bias.dtype != torch.float32):
            return False
        if c.weight.data_ptr() % 16:  # the kernels stage weights and rows with 16-byte loads
            return False
    from . import _lib

    return bool(_lib.lib().d3d_head_conv_supported(int(x.shape[1]), sum(c.out_channels for c in convs)))


def _ptr_array(tensors):
    import ctypes

    return (ctypes.c_void_p * len(tensors))(*[t.data_ptr() if t is not None else None for t in tensors])


class _HeadConvFn(torch.autograd.Function):
    """conv_box / conv_cls / conv_dir of a task group on a channels-last map: outputs [B, H, W, n_k] (what
    `.permute(0, 2, 3, 1).contiguous()` of the reference's NCHW conv outputs is, mg_head.py:223-231)"""

    @staticmethod
    @_lib.ctx_forward
    def forward(ctx, x, count, *params):
        from . import _lib

        L = _lib.lib()
        b, cin, h, w = x.shape
        m = b * h * w
        weights = [params[2 * k].detach() for k in range(count)]
        biases = [params[2 * k + 1] for k in range(count)]
        n = [int(wt.shape[0]) for wt in weights]
        outs = [torch.empty((b, h, w, nk), dtype=x.dtype, device=x.device) for nk in n]
        dt = _lib.D3D_BF16 if x.dtype == torch.bfloat16 else _lib.D3D_F32
        bias_arr = _ptr_array([bb.detach() if bb is not None else None for bb in biases])
        rc = L.d3d_head_conv_fwd(x.data_ptr(), m, cin, dt, _ptr_array(weights), bias_arr, _ptr_array(outs),
                                 _lib.i32_array(n), count, _lib.stream_ptr(x.device))
        _lib.check(rc, "d3d_head_conv_fwd")
        ctx.save_for_backward(x, *weights)
        ctx.meta = (count, n, [bb is not None for bb in biases])
        return tuple(outs)

    @staticmethod
    @_lib.ctx_backward
    def backward(ctx, *gouts):
        from . import _lib

        L = _lib.lib()
        count, n, has_bias = ctx.meta
        x, weights = ctx.saved_tensors[0], list(ctx.saved_tensors[1:])
        b, cin, h, w = x.shape
        m = b * h * w
        dev = x.device
        st = _lib.stream_ptr(dev)
        dt = _lib.D3D_BF16 if x.dtype == torch.bfloat16 else _lib.D3D_F32
        g = [(go if go is not None else torch.zeros((b, h, w, nk), dtype=x.dtype, device=dev)).to(x.dtype).contiguous()
             for go, nk in zip(gouts, n)]
        garr, narr = _ptr_array(g), _lib.i32_array(n)
        dx = None
        if ctx.needs_input_grad[0]:
            dxr = torch.empty((b, h, w, cin), dtype=x.dtype, device=dev)
            rc = L.d3d_head_conv_dgrad(garr, narr, count, m, cin, dt, _ptr_array(weights), dxr.data_ptr(), st)
            _lib.check(rc, "d3d_head_conv_dgrad")
            dx = dxr.permute(0, 3, 1, 2)
        nws = L.d3d_head_conv_wgrad_ws_floats(m)
        ws = torch.empty((nws,), dtype=torch.float32, device=dev)
        rc = L.d3d_head_conv_wgrad(x.data_ptr(), garr, narr, count, m, cin, dt, ws.data_ptr(), nws, st)
        _lib.check(rc, "d3d_head_conv_wgrad")
        # the result in a tensor of its own: views into `ws` would keep the whole slab workspace (~8 MB) alive as
        # p.grad and alias every head parameter's gradient to one buffer
        res = ws[: 33 * cin + 32].clone() if nws > 33 * cin + 32 else ws
        dw, db = res[: 32 * cin].view(32, cin), res[32 * cin: 32 * cin + 32]
        grads, first = [], 0
        for k in range(count):
            grads.append(dw[first: first + n[k]].reshape(n[k], cin, 1, 1))
            grads.append(db[first: first + n[k]] if has_bias[k] else None)
            first += n[k]
        return (dx, None) + tuple(grads)


@HEADS.register_module
class Head(nn.Module):
    def __init__(self, num_input, num_pred, num_cls, use_dir=False, num_dir=0, header=True, name="",
                 focal_loss_init=False, **kwargs):
        super().__init__(**kwargs)
        self.use_dir = use_dir
        self.conv_box = nn.Conv2d(num_input, num_pred, 1)
        self.conv_cls = nn.Conv2d(num_input, num_cls, 1)
        if self.use_dir:
            self.conv_dir = nn.Conv2d(num_input, num_dir, 1)

    def forward(self, x):
        convs = [self.conv_box, self.conv_cls] + ([self.conv_dir] if self.use_dir else [])
        if _head_conv_ok(x, convs):
            # one launch for the three 1x1 convolutions of the task group (csrc/d3d_head.hip): the map is read once
            outs = _HeadConvFn.apply(x, len(convs), *[t for c in convs for t in (c.weight, c.bias)])
        else:
            outs = [c(x).permute(0, 2, 3, 1).contiguous() for c in convs]
        ret = {"box_preds": outs[0], "cls_preds": outs[1]}
        if self.use_dir:
            ret["dir_cls_preds"] = outs[2]
        return ret


@HEADS.register_module
class MultiGroupHead(nn.Module):
    def __init__(self, mode="3d", in_channels=[128], norm_cfg=None, tasks=[], weights=[],
                 num_classes=[1], box_coder=None, with_cls=True, with_reg=True,
                 reg_class_agnostic=False, encode_background_as_zeros=True,
                 loss_norm=dict(type="NormByNumPositives", pos_class_weight=1.0, neg_class_weight=1.0),
                 loss_cls=dict(type="CrossEntropyLoss", use_sigmoid=False, loss_weight=1.0),
                 use_sigmoid_score=True, loss_bbox=dict(type="SmoothL1Loss", beta=1.0, loss_weight=1.0),
                 encode_rad_error_by_sin=True, loss_aux=None, direction_offset=0.0, name="rpn",
                 logger=None):
        super().__init__()
        assert with_cls or with_reg
        num_classes = [len(t["class_names"]) for t in tasks]
        self.class_names = [t["class_names"] for t in tasks]
        self.num_anchor_per_locs = [2 * n for n in num_classes]
        self.box_coder = box_coder
        box_code_sizes = [box_coder.code_size] * len(num_classes)
        self.with_cls = with_cls
        self.with_reg = with_reg
        self.in_channels = in_channels
        self.num_classes = num_classes
        self.reg_class_agnostic = reg_class_agnostic
        self.encode_rad_error_by_sin = encode_rad_error_by_sin
        self.encode_background_as_zeros = encode_background_as_zeros
        self.use_sigmoid_score = use_sigmoid_score
        self.box_n_dim = self.box_coder.code_size
        self.anchor_dim = self.box_coder.n_dim
        self.loss_cls = build_loss(loss_cls)
        self.loss_reg = build_loss(loss_bbox)
        if loss_aux is not None:
            self.loss_aux = build_loss(loss_aux)
        self.loss_norm = loss_norm
        self.logger = logger or logging.getLogger("MultiGroupHead")
        self.use_direction_classifier = loss_aux is not None
        if loss_aux:
            self.direction_offset = direction_offset
        self.bev_only = mode == "bev"

        self.tasks = nn.ModuleList()
        for num_c, num_a, box_cs in zip(num_classes, self.num_anchor_per_locs, box_code_sizes):
            num_cls = num_a * num_c if self.encode_background_as_zeros else num_a * (num_c + 1)
            num_pred = num_a * (box_cs - 2) if self.bev_only else num_a * box_cs
            num_dir = num_a * 2 if self.use_direction_classifier else None
            self.tasks.append(Head(in_channels, num_pred, num_cls, use_dir=self.use_direction_classifier,
                                   num_dir=num_dir, header=False))
        self.logger.info("Finish MultiGroupHead Initialization")

    def init_weights(self, pretrained=None):
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity="relu")
                if m.bias is not None:
                    nn.init.constant_(m.bias, 0)

    def forward(self, x):
        convs = _fused_heads_convs(self, x)
        if convs is not None:  # every task's 1x1 convolutions as one stacked convolution on the library's kernels
            return _run_fused_heads(self, x, convs)
        return [task(x) for task in self.tasks]

    # ------------------------------------------------------------------- loss
    def prepare_loss_weights(self, labels, loss_norm=dict(type="NormByNumPositives", pos_cls_weight=1.0,
                                                         neg_cls_weight=1.0), dtype=torch.float32):
        """per-anchor weights of the classification and the box terms, and the mask of the anchors that count
        (label >= 0).  The four normalisations of mg_head.py:505-560, from their definitions: the box term is always
        averaged over the frame's positives; the classification term is divided by the positives
        (NormByNumPositives), by the counted anchors (NormByNumExamples), by the size of the anchor's own class --
        positives or negatives -- (NormByNumPosNeg) or not at all (DontNorm)."""
        pos, neg = (labels > 0).to(dtype), (labels == 0).to(dtype)
        n_pos = pos.sum(1, keepdim=True)
        cls_w = pos * loss_norm["pos_cls_weight"] + neg * loss_norm["neg_cls_weight"]
        reg_w = pos / n_pos.clamp(min=1.0)
        kind = loss_norm["type"]
        if kind == "NormByNumPositives":
            cls_w = cls_w / n_pos.clamp(min=1.0)
        elif kind == "NormByNumExamples":
            cls_w = cls_w / (labels >= 0).to(dtype).sum(1, keepdim=True).clamp(min=1.0)
        elif kind == "NormByNumPosNeg":
            cls_w = cls_w / (pos * n_pos + neg * neg.sum(1, keepdim=True)).clamp(min=1.0)
        elif kind != "DontNorm":
            raise ValueError("unknown loss norm type %s" % kind)
        return cls_w, reg_w, labels >= 0

    def _fused_loss_ok(self, preds_dict, labels):
        """the configuration d3d_head_loss implements (every shipped config), on device tensors"""
        from . import losses as L

        box = preds_dict["box_preds"]
        if not (FUSE_LOSS and box.is_cuda and box.dtype in (torch.float32, torch.bfloat16)):
            return False
        if labels.dtype != torch.int32 or self.bev_only or not self.encode_background_as_zeros:
            return False
        if self.loss_norm["type"] != "NormByNumPositives":
            return False
        if type(self.loss_reg) is not L.WeightedSmoothL1Loss or not self.loss_reg._codewise:
            return False
        if type(self.loss_cls) is not L.SigmoidFocalLoss or self.loss_cls._alpha is None:
            return False
        if self.use_direction_classifier and (type(self.loss_aux) is not L.WeightedSoftmaxClassificationLoss
                                              or self.loss_aux._logit_scale != 1.0):
            return False
        return self.box_n_dim <= 10

    def loss_vector(self, example, preds_dict, task_id, batch_size, want=0):
        """the [D3D_LOSS_SCALARS] result of d3d_head_loss for one task group (element 0 = the loss); tensor in,
        tensor out, so that graph_tail.py can capture it.  want=slice(None): (vector, loss as a 0-dim tensor of its own)"""
        num_class = self.num_classes[task_id]
        labels = example["labels"][task_id].contiguous()
        reg_targets = example["reg_targets"][task_id].float().contiguous()
        box_preds = preds_dict["box_preds"].view(batch_size, -1, self.box_n_dim)
        cls_preds = preds_dict["cls_preds"].view(batch_size, -1, num_class)
        dir_preds = anchors = None
        if self.use_direction_classifier:
            dir_preds = preds_dict["dir_cls_preds"].view(batch_size, -1, 2)
            anchors = example["anchors"][task_id].float().contiguous()
            anchors = anchors.view(batch_size, -1, anchors.shape[-1])
        cfg = dict(sin_diff=self.encode_rad_error_by_sin, dir_offset=getattr(self, "direction_offset", 0.0) or 0.0,
                   pos_w=self.loss_norm["pos_cls_weight"], neg_w=self.loss_norm["neg_cls_weight"],
                   alpha=self.loss_cls._alpha, gamma=self.loss_cls._gamma or 0.0, sigma=self.loss_reg._sigma,
                   w_loc=self.loss_reg._loss_weight, w_cls=self.loss_cls._loss_weight,
                   w_dir=self.loss_aux._loss_weight if self.use_direction_classifier else 0.0)
        return _HeadLossFn.apply(box_preds, cls_preds, dir_preds, labels, reg_targets, anchors, cfg)[want]

    def _loss_dict(self, s, loss=None):
        d = s.detach()
        counts = d[6:8].long()  # (one conversion launch for the two anchor counts)
        return {"loss": s[0] if loss is None else loss, "cls_pos_loss": d[4], "cls_neg_loss": d[5],
                "dir_loss_reduced": d[3] if self.use_direction_classifier else torch.tensor(0),
                "cls_loss_reduced": d[2], "loc_loss_reduced": d[1],
                "loc_loss_elem": list(d[8: 8 + self.box_n_dim].unbind(0)),
                "num_pos": counts[0], "num_neg": counts[1]}

    def _loss_fused(self, example, preds_dict, task_id, batch_size):
        return self._loss_dict(*self.loss_vector(example, preds_dict, task_id, batch_size, want=slice(None)))

    def loss_vector_ok(self, feature_map, labels):
        """every task group of this head goes through the fused loss for inputs like these"""
        probe = {"box_preds": feature_map}
        return all(self._fused_loss_ok(probe, lab) for lab in labels)

    def loss_from_vectors(self, vecs):
        """the dict `loss()` returns, from the per-task vectors of `loss_vector`"""
        merged = defaultdict(list)
        for s in vecs:
            for k, v in self._loss_dict(s).items():
                merged[k].append(v)
        return merged

    def loss(self, example, preds_dicts, **kwargs):
        batch_anchors = example["anchors"]
        batch_size_device = batch_anchors[0].shape[0]
        rets = []
        for task_id, preds_dict in enumerate(preds_dicts):
            if self._fused_loss_ok(preds_dict, example["labels"][task_id]):
                rets.append(self._loss_fused(example, preds_dict, task_id, batch_size_device))
                continue
            rets.append(_eager_task_loss(self, preds_dict, example["labels"][task_id], example["reg_targets"][task_id],
                                         batch_anchors[task_id], self.num_classes[task_id], batch_size_device))
        merged = defaultdict(list)
        for ret in rets:
            for k, v in ret.items():
                merged[k].append(v)
        return merged

    # ---------------------------------------------------------------- predict
    def _fused_predict_ok(self, example, preds_dicts, test_cfg):
        nms_cfg = test_cfg["nms"] if isinstance(test_cfg, dict) else test_cfg.nms
        nget = (lambda k: nms_cfg[k]) if isinstance(nms_cfg, dict) else (lambda k: getattr(nms_cfg, k))
        box = preds_dicts[0]["box_preds"]
        return (FUSE_PREDICT and box.is_cuda and nget("use_rotate_nms") and not nget("use_multi_class_nms")
                and self.encode_background_as_zeros and self.use_sigmoid_score and not self.bev_only
                and "anchors_mask" not in example and 1 <= nget("nms_pre_max_size") <= 1024
                and self.anchor_dim in (7, 9) and self.box_n_dim <= 10 and not self.box_coder.linear_dim)

    def _predict_fused(self, example, preds_dicts, test_cfg):
        """predict() on the device: per task group decode + score filter + top-k (2 launches) and one batched
        rotated NMS (3 launches) for all frames, then a handful of [B, post] tensor ops for the direction flip
        (mg_head.py:1042-1051), the centre-range mask (:1055-1063) and the label offsets (:787-801); the per-frame
        result counts are read with ONE host sync per batch."""
        nms_cfg = test_cfg["nms"] if isinstance(test_cfg, dict) else test_cfg.nms
        tget = (lambda k: test_cfg[k]) if isinstance(test_cfg, dict) else (lambda k: getattr(test_cfg, k))
        nget = (lambda k: nms_cfg[k]) if isinstance(nms_cfg, dict) else (lambda k: getattr(nms_cfg, k))
        post, pre = int(nget("nms_post_max_size")), int(nget("nms_pre_max_size"))
        rng = tget("post_center_limit_range")
        batch = example["anchors"][0].shape[0]
        dev = preds_dicts[0]["box_preds"].device
        rng_t = torch.tensor(rng, dtype=torch.float32, device=dev) if len(rng) > 0 else None
        ar = torch.arange(post, device=dev)[None]
        all_boxes, all_scores, all_labels, all_valid = [], [], [], []
        flag = 0
        for task_id, preds in enumerate(preds_dicts):
            nc = self.num_classes[task_id]
            box = preds["box_preds"].reshape(batch, -1, self.box_n_dim)
            cls = preds["cls_preds"].reshape(batch, -1, nc)
            dirp = preds["dir_cls_preds"].reshape(batch, -1, 2) if self.use_direction_classifier else None
            anchors = example["anchors"][task_id].reshape(batch, -1, self.anchor_dim)
            cb, cs, cl, cd, cnt = box_ops.decode_filter_topk(box, cls, dirp, anchors, self.box_coder.vec_encode,
                                                             tget("score_threshold"), pre)
            keep, nkeep = box_ops.rotate_nms_batched(cb, cnt, nget("nms_iou_threshold"), post)
            idx = keep.clamp(min=0).long()
            sb = torch.gather(cb, 1, idx[..., None].expand(-1, -1, self.anchor_dim))
            ss = torch.gather(cs, 1, idx)
            sl = torch.gather(cl, 1, idx).long() + flag
            valid = ar < nkeep[:, None]
            if self.use_direction_classifier:
                sd = torch.gather(cd, 1, idx).bool()
                opp = ((sb[..., -1] - self.direction_offset) > 0) ^ sd
                sb[..., -1] += torch.where(opp, torch.tensor(np.pi, dtype=sb.dtype, device=dev),
                                           torch.tensor(0.0, dtype=sb.dtype, device=dev))
            if rng_t is not None:
                valid &= (sb[..., :3] >= rng_t[:3]).all(-1) & (sb[..., :3] <= rng_t[3:]).all(-1)
            all_boxes.append(sb); all_scores.append(ss); all_labels.append(sl); all_valid.append(valid)
            flag += nc
        boxes, scores = torch.cat(all_boxes, 1), torch.cat(all_scores, 1)
        labels, valid = torch.cat(all_labels, 1), torch.cat(all_valid, 1)
        # valid entries first, original order kept (tasks in order, NMS order within a task)
        order = torch.sort((~valid).to(torch.int8), dim=1, stable=True)[1]
        boxes = torch.gather(boxes, 1, order[..., None].expand(-1, -1, self.anchor_dim))
        scores, labels = torch.gather(scores, 1, order), torch.gather(labels, 1, order)
        n_valid = valid.sum(1).cpu().tolist()  # the one host sync
        meta = example["metadata"] if "metadata" in example and len(example["metadata"]) else [None] * batch
        return [{"box3d_lidar": boxes[b, : n_valid[b]], "scores": scores[b, : n_valid[b]],
                 "label_preds": labels[b, : n_valid[b]], "metadata": meta[b]} for b in range(batch)]

    def predict(self, example, preds_dicts, test_cfg, **kwargs):
        """detections per frame: [{"box3d_lidar" [n, box_dim], "scores" [n], "label_preds" [n], "metadata"}] -- the
        contract of MultiGroupHead.predict (mg_head.py:697-803).  On the GPU with the shipped settings the whole batch
        goes through `_predict_fused`; everything else (CPU tensors, anchor masks, axis-aligned NMS, softmax scores)
        is decided frame by frame in `_frame_detections`."""
        if self._fused_predict_ok(example, preds_dicts, test_cfg):
            return self._predict_fused(example, preds_dicts, test_cfg)
        batch = example["anchors"][0].shape[0]
        meta = example["metadata"] if len(example.get("metadata", [])) else [None] * batch
        frames = [dict(box3d_lidar=[], scores=[], label_preds=[], metadata=meta[b]) for b in range(batch)]
        first_label = 0  # labels of task t are offset by the class counts of tasks 0 .. t-1 (mg_head.py:787-801)
        for task_id, preds in enumerate(preds_dicts):
            n_cls = self.num_classes[task_id]
            n_logit = n_cls if self.encode_background_as_zeros else n_cls + 1
            code = self.box_n_dim - 2 if self.bev_only else self.box_n_dim
            anchors = example["anchors"][task_id].view(batch, -1, self.anchor_dim)
            raw = preds["box_preds"].float().view(batch, -1, code)
            boxes = self.box_coder.decode_torch(raw[:, :, : self.box_coder.code_size], anchors)
            logits = preds["cls_preds"].float().view(batch, -1, n_logit)
            dirs = preds["dir_cls_preds"].float().view(batch, -1, 2) if self.use_direction_classifier else None
            masks = example["anchors_mask"][task_id].view(batch, -1) if "anchors_mask" in example else None
            for b in range(batch):
                sel = slice(None) if masks is None else masks[b]
                det = self._frame_detections(test_cfg, boxes[b][sel], logits[b][sel],
                                             None if dirs is None else dirs[b][sel])
                frames[b]["box3d_lidar"].append(det[0])
                frames[b]["scores"].append(det[1])
                frames[b]["label_preds"].append(det[2] + first_label)
            first_label += n_cls
        for f in frames:
            for k in ("box3d_lidar", "scores", "label_preds"):
                f[k] = torch.cat(f[k])
        return frames

    def _class_scores(self, logits):
        """[A, classes] foreground scores from the classification logits"""
        if self.encode_background_as_zeros:
            if not self.use_sigmoid_score:
                raise AssertionError("encode_background_as_zeros needs use_sigmoid_score")
            return torch.sigmoid(logits)
        probs = torch.sigmoid(logits) if self.use_sigmoid_score else torch.softmax(logits, dim=-1)
        return probs[..., 1:]  # column 0 is the background

    def _frame_detections(self, test_cfg, boxes, logits, dir_logits):
        """(boxes [n, box_dim], scores [n], labels [n] int64) of one frame and one task group:
        score threshold -> NMS in the BEV plane -> direction bit -> centre range  (mg_head.py:805-1085)"""
        nms_cfg = test_cfg["nms"] if isinstance(test_cfg, dict) else test_cfg.nms
        tget = (lambda k: test_cfg[k]) if isinstance(test_cfg, dict) else (lambda k: getattr(test_cfg, k))
        nget = (lambda k: nms_cfg[k]) if isinstance(nms_cfg, dict) else (lambda k: getattr(nms_cfg, k))
        if nget("use_multi_class_nms"):
            raise NotImplementedError("use_multi_class_nms: unreachable in the reference too "
                                      "(mg_head.py:911-913 reads undefined attributes)")
        scores, labels = self._class_scores(logits).max(dim=-1)
        heading = dir_logits.argmax(dim=-1).bool() if dir_logits is not None else None
        thr = float(tget("score_threshold"))
        if thr > 0.0:
            hit = scores >= thr
            boxes, scores, labels = boxes[hit], scores[hit], labels[hit]
            heading = heading[hit] if heading is not None else None
        if scores.numel():
            bev = boxes[:, [0, 1, 3, 4, -1]]  # x, y, w, l, yaw
            if nget("use_rotate_nms"):
                suppress = box_ops.rotate_nms
            else:
                suppress = box_ops.nms
                bev = box_ops.corner_to_standup_nd(box_ops.center_to_corner_box2d(bev[:, :2], bev[:, 2:4], bev[:, 4]))
            keep = suppress(bev, scores, pre_max_size=nget("nms_pre_max_size"), post_max_size=nget("nms_post_max_size"),
                            iou_threshold=nget("nms_iou_threshold"))
            boxes, scores, labels = boxes[keep], scores[keep], labels[keep]
            heading = heading[keep] if heading is not None else None
        if heading is not None and boxes.shape[0]:
            # the regressed yaw is only known modulo pi; the direction classifier supplies the missing bit
            flip = ((boxes[:, -1] - self.direction_offset) > 0) ^ heading
            boxes = boxes.clone()
            boxes[:, -1] += flip.to(boxes.dtype) * np.pi
        limits = tget("post_center_limit_range")
        if len(limits) and boxes.shape[0]:
            lim = torch.as_tensor(limits, dtype=boxes.dtype, device=boxes.device)
            inside = ((boxes[:, :3] >= lim[:3]) & (boxes[:, :3] <= lim[3:])).all(dim=1)
            boxes, scores, labels = boxes[inside], scores[inside], labels[inside]
        return boxes.reshape(-1, self.anchor_dim), scores, labels


# ------------------------------------------------------------------------------------------------ fused task heads
# The multi-task configs (nuScenes: 6 task groups, Lyft: 5; det3d/models/bbox_heads/mg_head.py:193-232, :453-470) run
# 12-18 1x1 convolutions with 2-40 output channels each on the SAME 384 / 512-channel map.  As separate MIOpen calls
# (exact-f32 igemm at the f32 MFMA rate, a zero-fill launch in front of every weight / data gradient) they were 12 %
# of the nuScenes-PointPillars step and 5 % of the CBGS step.  Here they are ONE 1x1 convolution whose output channels
# are the task heads' rows stacked (zero rows up to a multiple of 128) on the library's NHWC implicit-GEMM kernels in
# split arithmetic: forward, data gradient (the transposed stack) and weight gradient are one launch each, the
# per-head tensors are slices of the result.
FUSE_HEADS = os.environ.get("D3D_FUSE_HEADS", "1") != "0"


def _fused_heads_convs(head, x):
    """[(conv, ...)] of every task in forward order if `head`'s tasks can run as one stacked 1x1 convolution on x"""
    from . import dense_conv as dc

    if not (FUSE_HEADS and torch.is_tensor(x) and x.is_cuda and x.dim() == 4 and x.dtype == torch.float32):
        return None
    if torch.is_autocast_enabled("cuda") or x.shape[1] not in (256, 384, 512):
        return None
    if not x.is_contiguous(memory_format=torch.channels_last) or x.numel() * 2 >= (1 << 31) - 16:
        return None
    convs = []
    for task in head.tasks:
        if type(task) is not Head or task._forward_hooks or task._forward_pre_hooks:
            return None  # (hooks registered on a Head must keep seeing its forward)
        cs = [task.conv_box, task.conv_cls] + ([task.conv_dir] if task.use_dir else [])
        for c in cs:
            if type(c) is not nn.Conv2d or c.kernel_size != (1, 1) or c.stride != (1, 1) or c.padding != (0, 0) or \
                    c.groups != 1 or c.weight.dtype != torch.float32 or c.bias is None or c.bias.dtype != torch.float32:
                return None
        convs.append(cs)
    total = sum(c.out_channels for cs in convs for c in cs)
    if len(convs) < 2 or total > 256:
        return None
    npad = 128 if total <= 128 else 256
    cin = int(x.shape[1])
    # the three launches of the node at THIS shape: forward cin -> npad, data gradient npad -> cin (in 128 / 256 column
    # tiles of the same kernel), weight gradient
    if not (dc.supported(cin, npad, 1, 1) and dc.supported(npad, cin, 1, 1)):
        return None
    return convs


_STACKED_PACKS = {}  # ids of the heads' conv weights -> (versions, weakrefs, packed planes)


class _FusedHeadsFn(torch.autograd.Function):
    @staticmethod
    @_lib.ctx_forward
    def forward(ctx, x, sizes, *params):
        from . import _lib
        from . import dense_conv as dc

        L = _lib.lib()
        dev = x.device
        b, cin, h, w = x.shape
        weights, biases = params[0::2], params[1::2]
        total = sum(sizes)
        npad = 128 if total <= 128 else 256
        rows = x.permute(0, 2, 3, 1)
        if not rows.is_contiguous():
            rows = rows.contiguous()
        x_hi, x_lo = dc.split_planes(rows)
        # the stacked [npad, cin] operand in both packed forms, cached per (parameters, versions): evaluation and repeated
        # forwards between optimizer steps pack nothing; a step moves the versions (fused_adam bumps them)
        key = tuple(id(wt) for wt in weights)
        stamp = tuple((wt._version, wt.data_ptr()) for wt in weights) + (npad, cin, str(dev))
        hit = _STACKED_PACKS.get(key)
        if hit is not None and hit[0] == stamp and all(r() is wt for r, wt in zip(hit[1], weights)):
            f_hi, f_lo, d_hi, d_lo = hit[2]
        else:
            w_all = torch.zeros((npad, cin), dtype=torch.float32, device=dev)
            torch.cat([wt.detach().reshape(wt.shape[0], cin) for wt in weights], 0, out=w_all[:total])
            f_hi = torch.empty((1, npad, cin), dtype=torch.bfloat16, device=dev)
            f_lo = torch.empty_like(f_hi)
            d_hi = torch.empty((1, cin, npad), dtype=torch.bfloat16, device=dev)
            d_lo = torch.empty_like(d_hi)
            rc = L.d3d_conv2d_pack_weights(w_all.data_ptr(), npad, cin, 1, f_hi.data_ptr(), f_lo.data_ptr(), d_hi.data_ptr(),
                                           d_lo.data_ptr(), _lib.stream_ptr(dev))
            _lib.check(rc, "d3d_conv2d_pack_weights")
            if len(_STACKED_PACKS) > 64:
                _STACKED_PACKS.clear()
            _STACKED_PACKS[key] = (stamp, [weakref.ref(wt) for wt in weights], (f_hi, f_lo, d_hi, d_lo))
        y_all, _ = dc.conv2d_raw(x_hi, x_lo, f_hi, f_lo, 1, 1, 0, "split")  # [B, H, W, npad] f32
        outs, off = [], 0
        for n, bias in zip(sizes, biases):
            outs.append(y_all[..., off: off + n] + bias.detach())  # contiguous [B, H, W, n]
            off += n
        ctx.save_for_backward(x_hi, x_lo, d_hi, d_lo)
        ctx.meta = (tuple(sizes), npad, (b, cin, h, w))
        return tuple(outs)

    @staticmethod
    @_lib.ctx_backward
    def backward(ctx, *gouts):
        from . import dense_conv as dc

        x_hi, x_lo, d_hi, d_lo = ctx.saved_tensors
        sizes, npad, (b, cin, h, w) = ctx.meta
        dev = x_hi.device
        total = sum(sizes)
        parts = [(g if g is not None else torch.zeros((b, h, w, n), dtype=torch.float32, device=dev)).float()
                 for g, n in zip(gouts, sizes)]
        if npad > total:
            parts.append(torch.zeros((b, h, w, npad - total), dtype=torch.float32, device=dev))
        g_all = torch.cat(parts, -1)                         # [B, H, W, npad]
        db_all = g_all.sum((0, 1, 2))                        # bias gradients of every head at once
        g_hi, g_lo = dc.split_planes(g_all)
        dx = None
        if ctx.needs_input_grad[0]:
            dxr, _ = dc.conv2d_raw(g_hi, g_lo, d_hi, d_lo, 1, 1, 0, "split")  # [B, H, W, cin]
            dx = dxr.permute(0, 3, 1, 2)
        dw_all = dc.wgrad_raw(g_hi, g_lo, x_hi, x_lo, 1, 1, 0, "split")       # [npad, cin, 1, 1]
        grads, off = [], 0
        for n in sizes:
            grads.append(dw_all[off: off + n].clone())
            grads.append(db_all[off: off + n].clone())
            off += n
        return (dx, None) + tuple(grads)


def _run_fused_heads(head, x, convs):
    flat = [c for cs in convs for c in cs]
    outs = _FusedHeadsFn.apply(x, [c.out_channels for c in flat], *[t for c in flat for t in (c.weight, c.bias)])
    rets, i = [], 0
    for task, cs in zip(head.tasks, convs):
        ret = {"box_preds": outs[i], "cls_preds": outs[i + 1]}
        if task.use_dir:
            ret["dir_cls_preds"] = outs[i + 2]
        rets.append(ret)
        i += len(cs)
    return rets
