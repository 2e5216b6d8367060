"""TEST INFRASTRUCTURE -- the reference's CPU path for the SECOND detector, restated with
stock torch.nn layers (fp32, PyTorch-CPU) + this package's CPU voxelizer / sparse conv / NMS
oracles.  Used (a) as the end-to-end checker for the HIP path (box regressions within 1e-3)
and (b) as bench.py's `cpu_baseline` ("port": reference CPU voxelizer + PyTorch-CPU backbone
on the host cores).

Follows, with the same module nesting so a det3d_amd state_dict loads unchanged:
  reader    det3d/models/readers/voxel_encoder.py:206-211      mean VFE
  backbone  det3d/models/backbones/scn.py:92-197               SpMiddleFHD (layer list :106-157)
            det3d/models/backbones/scn.py:46-89,308-370        SparseBasicBlock, SpMiddleResNetFHD
            det3d/models/readers/pillar_encoder.py:17-211      PFNLayer, PillarFeatureNet, PointPillarsScatter
  neck      det3d/models/necks/rpn.py:70-159                   RPN blocks / deblocks / concat
  head      det3d/models/bbox_heads/mg_head.py:215-230,528-533 1x1 convs, NHWC permute
  loss      mg_head.py:535-695 + losses/losses.py:146-336,407-446
  predict   mg_head.py:697-1085 + core/bbox/box_torch_ops.py:80-148,528-549
The RPN/head/loss restatements are pinned by tests/golden/{rpn,head}.pt (outputs of the
reference's own modules); the sparse part inherits spconv_ref's "parity unpinned" status.
Whole detectors (KITTI SECOND, nuScenes CBGS, Lyft CBGS, nuScenes PointPillars) are pinned by
tests/golden/detectors.pt: the reference's own VoxelNet / PointPillars classes -- scn.py executed over
oracle/spconv_shim.py -- run in the build container on cropped ranges (tests/golden/make_golden.py).
"""
import math

import numpy as np
import torch
import torch.nn.functional as F
from torch import nn

from . import clib
from .spconv_ref import SparseConvRef, dense

BN = dict(eps=1e-3, momentum=0.01)


class SpMiddleFHDRef(nn.Module):
    def __init__(self, cin=4):
        super().__init__()
        L = []

        def subm(i, o, key):
            L.extend([SparseConvRef(i, o, 3, subm=True, indice_key=key), nn.BatchNorm1d(o, **BN), nn.ReLU()])

        def down(i, o, k, s, p):
            L.extend([SparseConvRef(i, o, k, s, p), nn.BatchNorm1d(o, **BN), nn.ReLU()])

        subm(cin, 16, "subm0"); subm(16, 16, "subm0")
        down(16, 32, 3, 2, 1)
        subm(32, 32, "subm1"); subm(32, 32, "subm1")
        down(32, 64, 3, 2, 1)
        for _ in range(3):
            subm(64, 64, "subm2")
        down(64, 64, 3, 2, [0, 1, 1])
        for _ in range(3):
            subm(64, 64, "subm3")
        down(64, 64, (3, 1, 1), (2, 1, 1), 0)
        self.middle_conv = nn.Sequential(*L)

    def forward(self, feats, coors, batch_size, input_shape):
        shape = (np.array(input_shape[::-1]) + [1, 0, 0]).tolist()
        st = (feats, np.asarray(coors).astype(np.int32), shape, batch_size, {})
        for m in self.middle_conv:
            if isinstance(m, SparseConvRef):
                st = m(st)
            else:
                st = (m(st[0]),) + st[1:]
        feats, idx, shape, batch, cache = st
        self.last = st
        d = dense(feats, idx, batch, shape)
        n, c, dd, h, w = d.shape
        return d.view(n, c * dd, h, w)


class SparseBasicBlockRef(nn.Module):
    """scn.py:46-89 -- conv1 -> bn1 -> relu -> conv2 -> bn2 -> (+ identity) -> relu; both convs carry a bias
    (`bias = norm_cfg is not None` after the default was filled in, :60-63)."""

    def __init__(self, c, key):
        super().__init__()
        self.conv1 = SparseConvRef(c, c, 3, subm=True, bias=True, indice_key=key)
        self.bn1 = nn.BatchNorm1d(c, **BN)
        self.conv2 = SparseConvRef(c, c, 3, subm=True, bias=True, indice_key=key)
        self.bn2 = nn.BatchNorm1d(c, **BN)

    def forward(self, st):
        identity = st[0]
        out = self.conv1(st)
        out = (F.relu(self.bn1(out[0])),) + out[1:]
        out = self.conv2(out)
        return (F.relu(self.bn2(out[0]) + identity),) + out[1:]


def _run_sparse_stack(seq, feats, coors, batch_size, input_shape):
    shape = (np.array(input_shape[::-1]) + [1, 0, 0]).tolist()
    st = (feats, np.asarray(coors).astype(np.int32), shape, batch_size, {})
    for m in seq:
        if isinstance(m, (SparseConvRef, SparseBasicBlockRef)):
            st = m(st)
        else:
            st = (m(st[0]),) + st[1:]
    feats, idx, shape, batch, cache = st
    d = dense(feats, idx, batch, shape)
    n, c, dd, h, w = d.shape
    return st, d.view(n, c * dd, h, w)


class SpMiddleResNetFHDRef(nn.Module):
    """scn.py:308-370; child indices of `middle_conv` equal the reference's SparseSequential."""

    def __init__(self, cin=5):
        super().__init__()
        L = [SparseConvRef(cin, 16, 3, subm=True, indice_key="res0"), nn.BatchNorm1d(16, **BN), nn.ReLU(),
             SparseBasicBlockRef(16, "res0"), SparseBasicBlockRef(16, "res0")]
        for i, (ci, co, pad) in enumerate([(16, 32, 1), (32, 64, 1), (64, 128, [0, 1, 1])]):
            L += [SparseConvRef(ci, co, 3, 2, pad), nn.BatchNorm1d(co, **BN), nn.ReLU(),
                  SparseBasicBlockRef(co, "res%d" % (i + 1)), SparseBasicBlockRef(co, "res%d" % (i + 1))]
        L += [SparseConvRef(128, 128, (3, 1, 1), (2, 1, 1), 0), nn.BatchNorm1d(128, **BN), nn.ReLU()]
        self.middle_conv = nn.Sequential(*L)

    def forward(self, feats, coors, batch_size, input_shape):
        self.last, out = _run_sparse_stack(self.middle_conv, feats, coors, batch_size, input_shape)
        return out


class _PFNLayerRef(nn.Module):
    def __init__(self, cin, cout):
        super().__init__()
        self.linear = nn.Linear(cin, cout, bias=False)
        self.norm = nn.BatchNorm1d(cout, **BN)


class PillarFeatureNetRef(nn.Module):
    """pillar_encoder.py:60-155 with a single (last) PFNLayer, as every shipped config has."""

    def __init__(self, num_input_features, num_filters, voxel_size, pc_range):
        super().__init__()
        assert len(num_filters) == 1
        self.pfn_layers = nn.ModuleList([_PFNLayerRef(num_input_features + 5, num_filters[0])])
        self.vx, self.vy = voxel_size[0], voxel_size[1]
        self.x_offset = self.vx / 2 + pc_range[0]
        self.y_offset = self.vy / 2 + pc_range[1]

    def forward(self, features, num_voxels, coors):
        mean = features[:, :, :3].sum(1, keepdim=True) / num_voxels.type_as(features).view(-1, 1, 1)
        f_cluster = features[:, :, :3] - mean
        f_center = torch.zeros_like(features[:, :, :2])
        f_center[:, :, 0] = features[:, :, 0] - (coors[:, 3].to(features.dtype).unsqueeze(1) * self.vx + self.x_offset)
        f_center[:, :, 1] = features[:, :, 1] - (coors[:, 2].to(features.dtype).unsqueeze(1) * self.vy + self.y_offset)
        x = torch.cat([features, f_cluster, f_center], -1)
        T = x.shape[1]
        mask = (torch.arange(T)[None, :] < num_voxels.view(-1, 1)).type_as(x).unsqueeze(-1)
        x = x * mask
        layer = self.pfn_layers[0]
        x = layer.linear(x)
        x = layer.norm(x.permute(0, 2, 1).contiguous()).permute(0, 2, 1).contiguous()
        return F.relu(x).max(1)[0]


class PointPillarsScatterRef(nn.Module):
    """pillar_encoder.py:158-211"""

    def __init__(self, c):
        super().__init__()
        self.c = c

    def forward(self, feats, coors, batch_size, input_shape):
        nx, ny = int(input_shape[0]), int(input_shape[1])
        coors = torch.as_tensor(np.asarray(coors)).long()
        canvas = feats.new_zeros((batch_size, self.c, ny * nx))
        canvas[coors[:, 0], :, coors[:, 2] * nx + coors[:, 3]] = feats
        return canvas.view(batch_size, self.c, ny, nx)


def _rpn_block(cin, cout, n, stride):
    L = [nn.ZeroPad2d(1), nn.Conv2d(cin, cout, 3, stride=stride, bias=False), nn.BatchNorm2d(cout, **BN), nn.ReLU()]
    for _ in range(n):
        L += [nn.Conv2d(cout, cout, 3, padding=1, bias=False), nn.BatchNorm2d(cout, **BN), nn.ReLU()]
    return nn.Sequential(*L)


class RPNRef(nn.Module):
    def __init__(self, layer_nums, ds_layer_strides, ds_num_filters, us_layer_strides, us_num_filters,
                 num_input_features):
        super().__init__()
        ins = [num_input_features] + list(ds_num_filters[:-1])
        self.start = len(layer_nums) - len(us_layer_strides)
        self.blocks = nn.ModuleList()
        self.deblocks = nn.ModuleList()
        for i, n in enumerate(layer_nums):
            self.blocks.append(_rpn_block(ins[i], ds_num_filters[i], n, ds_layer_strides[i]))
            j = i - self.start
            if j >= 0:
                s, oc = us_layer_strides[j], us_num_filters[j]
                if s > 1:
                    up = nn.ConvTranspose2d(ds_num_filters[i], oc, s, stride=s, bias=False)
                else:
                    k = int(round(1 / s))
                    up = nn.Conv2d(ds_num_filters[i], oc, k, stride=k, bias=False)
                self.deblocks.append(nn.Sequential(up, nn.BatchNorm2d(oc, **BN), nn.ReLU()))

    def forward(self, x):
        ups = []
        for i, b in enumerate(self.blocks):
            x = b(x)
            if i - self.start >= 0:
                ups.append(self.deblocks[i - self.start](x))
        return torch.cat(ups, 1) if ups else x


class _TaskHead(nn.Module):
    def __init__(self, cin, n_box, n_cls, n_dir):
        super().__init__()
        self.conv_box = nn.Conv2d(cin, n_box, 1)
        self.conv_cls = nn.Conv2d(cin, n_cls, 1)
        if n_dir:
            self.conv_dir = nn.Conv2d(cin, n_dir, 1)
        self.has_dir = bool(n_dir)

    def forward(self, x):
        nhwc = lambda t: t.permute(0, 2, 3, 1).contiguous()
        out = dict(box_preds=nhwc(self.conv_box(x)), cls_preds=nhwc(self.conv_cls(x)))
        if self.has_dir:
            out["dir_cls_preds"] = nhwc(self.conv_dir(x))
        return out


class _HeadRef(nn.Module):
    def __init__(self, cin, num_classes, code_size, use_dir):
        super().__init__()
        self.tasks = nn.ModuleList(_TaskHead(cin, 2 * nc * code_size, 2 * nc * nc, 4 * nc if use_dir else 0)
                                   for nc in num_classes)


def _focal(logits, onehot, weights, alpha=0.25, gamma=2.0):
    ce = torch.clamp(logits, min=0) - logits * onehot + torch.log1p(torch.exp(-logits.abs()))
    p = torch.sigmoid(logits)
    pt = onehot * p + (1 - onehot) * (1 - p)
    aw = onehot * alpha + (1 - onehot) * (1 - alpha)
    return (1 - pt) ** gamma * aw * ce * weights.unsqueeze(-1)


def _smooth_l1(pred, tgt, weights, sigma=3.0):
    d = (pred - tgt).abs()
    lt = (d <= 1 / sigma ** 2).float()
    loss = lt * 0.5 * (d * sigma) ** 2 + (d - 0.5 / sigma ** 2) * (1 - lt)
    return loss * weights.unsqueeze(-1)


def _decode(enc, anchors, vec=False):
    """second_box_decode, core/bbox/box_torch_ops.py:80-148 (7- and 9-dim anchors, optional angle vector)"""
    if anchors.shape[-1] == 9:
        xa, ya, za, wa, la, ha, vxa, vya, ra = anchors.unbind(-1)
    else:
        xa, ya, za, wa, la, ha, ra = anchors.unbind(-1)
    t = enc.unbind(-1)
    xt, yt, zt, wt, lt, ht = t[:6]
    diag = torch.sqrt(la ** 2 + wa ** 2)
    out = [xt * diag + xa, yt * diag + ya, zt * ha + za, torch.exp(wt) * wa, torch.exp(lt) * la, torch.exp(ht) * ha]
    rest = t[6:]
    if anchors.shape[-1] == 9:
        out += [rest[0] + vxa, rest[1] + vya]
        rest = rest[2:]
    if vec:
        out.append(torch.atan2(rest[1] + torch.sin(ra), rest[0] + torch.cos(ra)))
    else:
        out.append(rest[0] + ra)
    return torch.stack(out, -1)


class DetectorRef(nn.Module):
    """VoxelNet (detectors/voxelnet.py:22-59) / PointPillars (detectors/point_pillars.py:22-59) on the CPU,
    any of the shipped model configs.  `example` uses the collate_kitti keys with CPU tensors."""

    def __init__(self, cfg):
        super().__init__()
        g = (lambda o, k: o[k]) if isinstance(cfg, dict) else (lambda o, k: getattr(o, k))
        m = g(cfg, "model")
        neck = m["neck"]
        rd, bb = m["reader"], m["backbone"]
        self.reader_type = rd["type"]
        self.n_in = rd["num_input_features"]
        if rd["type"] == "PillarFeatureNet":
            self.reader = PillarFeatureNetRef(rd["num_input_features"], list(rd["num_filters"]), rd["voxel_size"],
                                              rd["pc_range"])
        if bb["type"] == "SpMiddleFHD":
            self.backbone = SpMiddleFHDRef(bb["num_input_features"])
        elif bb["type"] == "SpMiddleResNetFHD":
            self.backbone = SpMiddleResNetFHDRef(bb["num_input_features"])
        elif bb["type"] == "PointPillarsScatter":
            self.backbone = PointPillarsScatterRef(bb.get("num_input_features", 64))
        else:
            raise ValueError(bb["type"])
        self.neck = RPNRef(neck["layer_nums"], neck["ds_layer_strides"], neck["ds_num_filters"],
                           neck["us_layer_strides"], neck["us_num_filters"], neck["num_input_features"])
        h = m["bbox_head"]
        coder = h["box_coder"]
        self.code_size, self.n_dim, self.vec = coder.code_size, coder.n_dim, bool(coder.vec_encode)
        self.num_classes = [len(t["class_names"]) for t in h["tasks"]]
        self.use_dir = h.get("loss_aux") is not None
        self.sin_diff = bool(h.get("encode_rad_error_by_sin", True))
        self.dir_offset = h.get("direction_offset", 0.0)
        in_ch = h["in_channels"]
        self.bbox_head = _HeadRef(in_ch[0] if isinstance(in_ch, (list, tuple)) else in_ch, self.num_classes,
                                  self.code_size, self.use_dir)
        self.w_loc, self.w_cls = h["loss_bbox"]["loss_weight"], h["loss_cls"]["loss_weight"]
        self.w_dir = h["loss_aux"]["loss_weight"] if self.use_dir else 0.0
        self.pos_w = h["loss_norm"].get("pos_cls_weight", 1.0)
        self.neg_w = h["loss_norm"].get("neg_cls_weight", 1.0)
        self.test_cfg = g(cfg, "test_cfg")

    def features(self, example):
        v, n = example["voxels"], example["num_points"]
        coors = example["coordinates"]
        if self.reader_type == "PillarFeatureNet":
            f = self.reader(v, n, coors)
        else:  # VoxelFeatureExtractorV3, voxel_encoder.py:206-211
            f = v[:, :, : self.n_in].sum(1) / n.type_as(v).view(-1, 1)
        x = self.backbone(f, coors.numpy(), len(example["num_voxels"]), [int(s) for s in example["shape"][0]])
        return self.neck(x)

    def forward(self, example, return_loss=True):
        x = self.features(example)
        preds = [t(x) for t in self.bbox_head.tasks]
        return self.loss(example, preds) if return_loss else self.predict(example, preds)

    def loss(self, example, preds):
        out = []
        cs = self.code_size
        for ti, p in enumerate(preds):
            nc = self.num_classes[ti]
            labels = example["labels"][ti]
            B = labels.shape[0]
            box = p["box_preds"].view(B, -1, cs)
            cls = p["cls_preds"].view(B, -1, nc)
            tgt = example["reg_targets"][ti]
            pos, neg = labels > 0, labels == 0
            norm = torch.clamp(pos.sum(1, keepdim=True).float(), min=1.0)
            cls_w = (neg.float() * self.neg_w + pos.float() * self.pos_w) / norm
            reg_w = pos.float() / norm
            ct = (labels * (labels >= 0).type_as(labels)).long()
            onehot = F.one_hot(ct, nc + 1)[..., 1:].float()
            if self.sin_diff:
                b1 = torch.cat([box[..., :-1], torch.sin(box[..., -1:]) * torch.cos(tgt[..., -1:])], -1)
                b2 = torch.cat([tgt[..., :-1], torch.cos(box[..., -1:]) * torch.sin(tgt[..., -1:])], -1)
            else:
                b1, b2 = box, tgt
            # WeightedSmoothL1Loss never applies its code_weights (losses.py:173: the attribute is read
            # before it is set), so all code dimensions weigh 1
            loss = _smooth_l1(b1, b2, reg_w).sum() / B * self.w_loc + _focal(cls, onehot, cls_w).sum() / B * self.w_cls
            if self.use_dir:
                an = example["anchors"][ti].view(B, -1, self.n_dim)
                rot = tgt[..., -1] + an[..., -1] - self.dir_offset
                rot = rot - torch.floor(rot / (2 * math.pi) + 0.5) * 2 * math.pi
                dt = (rot > 0).long()
                logits = p["dir_cls_preds"].view(B, -1, 2)
                w = pos.float() / torch.clamp(pos.float().sum(-1, keepdim=True), min=1.0)
                ce = F.cross_entropy(logits.reshape(-1, 2), dt.reshape(-1), reduction="none").view(B, -1)
                loss = loss + (ce * w).sum() / B * self.w_dir
            out.append(loss)
        return dict(loss=out)

    @torch.no_grad()
    def predict(self, example, preds):
        tc = self.test_cfg
        nms = tc["nms"]
        rng = torch.tensor(tc["post_center_limit_range"], dtype=torch.float32)
        B = example["anchors"][0].shape[0]
        dets = []
        for b in range(B):
            boxes_all, scores_all, labels_all = [], [], []
            off = 0
            for ti, p in enumerate(preds):
                nc = self.num_classes[ti]
                an = example["anchors"][ti][b].view(-1, self.n_dim)
                box = _decode(p["box_preds"][b].reshape(-1, self.code_size).float(), an, self.vec)
                sc_all = torch.sigmoid(p["cls_preds"][b].reshape(-1, nc).float())
                sc, lab = (sc_all.squeeze(-1), torch.zeros(len(sc_all), dtype=torch.long)) if nc == 1 else sc_all.max(-1)
                keep = sc >= tc["score_threshold"]
                box, sc, lab = box[keep], sc[keep], lab[keep]
                dl = p["dir_cls_preds"][b].reshape(-1, 2).max(-1)[1][keep] if self.use_dir else None
                if len(sc):
                    k = min(len(sc), nms["nms_pre_max_size"])
                    sc_k, idx = torch.topk(sc, k)
                    d6 = torch.cat([box[idx][:, [0, 1, 3, 4, -1]], sc_k[:, None]], 1).numpy()
                    sel = clib.rotate_nms_cc(d6, nms["nms_iou_threshold"])[: nms["nms_post_max_size"]]
                    sel = idx[torch.as_tensor(sel, dtype=torch.long)]
                else:
                    sel = torch.zeros(0, dtype=torch.long)
                box, sc, lab = box[sel], sc[sel], lab[sel]
                if self.use_dir and len(sel):
                    opp = ((box[:, -1] - self.dir_offset) > 0) ^ dl[sel].bool()
                    box[:, -1] += torch.where(opp, torch.tensor(math.pi), torch.tensor(0.0))
                m = ((box[:, :3] >= rng[:3]).all(1) & (box[:, :3] <= rng[3:]).all(1)) if len(sel) else torch.zeros(0, dtype=torch.bool)
                boxes_all.append(box[m]); scores_all.append(sc[m]); labels_all.append(lab[m] + off)
                off += nc
            dets.append(dict(box3d_lidar=torch.cat(boxes_all), scores=torch.cat(scores_all),
                             label_preds=torch.cat(labels_all)))
        return dets


SecondRef = DetectorRef  # the KITTI-car SECOND instance (bench.py's cpu_baseline, tests/test_e2e_gpu.py)


def fill_deterministic(module, seed=0):
    """Give `module` weights that depend only on (state-dict key, shape, seed) -- the golden script fills the
    REFERENCE's model and the tests fill the product / oracle models with the same values without a 30 MB
    weight fixture.  Conv / linear weights ~ N(0, 2 / fan_in); BatchNorm gamma ~ U(0.5, 1.5); biases and running
    means ~ N(0, 0.1); running variances ~ U(0.5, 1.5)."""
    import zlib

    sd = module.state_dict()
    with torch.no_grad():
        for name in sorted(sd):
            t = sd[name]
            if not t.is_floating_point():
                continue
            g = torch.Generator().manual_seed((zlib.crc32(name.encode()) + 7919 * seed) % (2 ** 31))
            shape = tuple(t.shape)
            if name.endswith("running_var"):
                v = torch.rand(shape, generator=g) + 0.5
            elif name.endswith("running_mean") or name.endswith("bias"):
                v = torch.randn(shape, generator=g) * 0.1
            elif t.dim() == 1:  # BatchNorm weight
                v = torch.rand(shape, generator=g) + 0.5
            else:
                fan_in = (t.numel() // shape[-1]) if t.dim() == 5 else (t.numel() // shape[0])
                v = torch.randn(shape, generator=g) * math.sqrt(2.0 / max(fan_in, 1))
            t.copy_(v.to(t.dtype))
    return module


def tame_head(module, box_scale=0.03, cls_scale=0.5, cls_bias=-1.0):
    """After fill_deterministic: give the prediction convs of every task head the magnitude of a TRAINED head -- box
    regressions of O(0.1) (fill_deterministic's O(10) size regressions decode to boxes of 1e4 m whose NMS decisions are
    arbitrary on every implementation), class logits spread around `cls_bias` so that some anchors pass the score
    threshold and the rotated NMS has real overlaps to resolve.  tests/golden/detectors_sane.pt is made with it."""
    with torch.no_grad():
        for name, t in module.state_dict().items():
            if ".conv_box." in name:
                t.mul_(box_scale)
            elif name.endswith("conv_cls.weight"):
                t.mul_(cls_scale)
            elif name.endswith("conv_cls.bias"):
                t.mul_(cls_scale).add_(cls_bias)
    return module


def cpu_example(host_batch, cfg_vox):
    """reference CPU data path: per-frame oracle voxelizer + collate_kitti layout."""
    vox, coors, nums, nv = [], [], [], []
    off = host_batch["frame_offsets"]
    for b in range(len(off) - 1):
        v, c, n = clib.points_to_voxel(host_batch["points"][off[b]: off[b + 1]], cfg_vox["voxel_size"],
                                       cfg_vox["pc_range"], cfg_vox["max_points"], True, cfg_vox["max_voxels"])
        vox.append(v); nums.append(n); nv.append(len(c))
        coors.append(np.pad(c, ((0, 0), (1, 0)), mode="constant", constant_values=b))
    grid = clib.grid_size(cfg_vox["voxel_size"], cfg_vox["pc_range"]).astype(np.int64)
    t = torch.from_numpy
    ex = dict(voxels=t(np.concatenate(vox)), coordinates=t(np.concatenate(coors)), num_points=t(np.concatenate(nums)),
              num_voxels=np.array(nv, dtype=np.int64), shape=np.stack([grid] * (len(off) - 1)),
              anchors=[t(host_batch["anchors"])])
    for k in ("labels", "reg_targets"):
        if k in host_batch:
            ex[k] = [t(host_batch[k])]
    return ex
