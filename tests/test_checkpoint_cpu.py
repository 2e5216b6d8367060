"""Checkpoint import (det3d/torchie/trainer/checkpoint.py:42-215 semantics) and BatchNorm folding."""
import os
from collections import OrderedDict

import numpy as np
import pytest
import torch
from torch import nn

from det3d_amd import Config, build_detector, checkpoint, spconv
from det3d_amd.necks import RPN

HERE = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _kitti_model():
    cfg = Config.fromfile(os.path.join(HERE, "det3d_amd", "configs", "kitti_car_second.py"))
    return build_detector(cfg.model, cfg.train_cfg, cfg.test_cfg)


def test_save_load_roundtrip_with_ddp_prefix_and_tolerant_matching(tmp_path, caplog):
    torch.manual_seed(0)
    a, b = _kitti_model(), _kitti_model()
    for p in a.parameters():
        nn.init.normal_(p, std=0.1)
    f = str(tmp_path / "ckpt" / "epoch_1.pth")
    checkpoint.save_checkpoint(a, f, meta=dict(epoch=1))
    ck = torch.load(f, weights_only=False)
    assert set(ck) == {"meta", "state_dict"} and ck["meta"]["epoch"] == 1
    # a DDP-wrapped save: every key prefixed with "module."
    ck2 = dict(state_dict=OrderedDict(("module." + k, v) for k, v in ck["state_dict"].items()))
    f2 = str(tmp_path / "ddp.pth")
    torch.save(ck2, f2)
    checkpoint.load_checkpoint(b, f2, map_location="cpu", strict=True)
    for (n1, p1), (n2, p2) in zip(a.state_dict().items(), b.state_dict().items()):
        assert n1 == n2 and torch.equal(p1, p2), n1
    # sparse weights keep spconv's [kD, kH, kW, Cin, Cout] layout
    assert tuple(b.state_dict()["backbone.middle_conv.0.weight"].shape) == (3, 3, 3, 4, 16)
    # tolerant mode: unexpected / missing / mismatched keys are reported, not fatal; strict raises
    sd = OrderedDict(ck["state_dict"])
    sd["bogus.weight"] = torch.zeros(1)
    sd["neck.blocks.0.1.weight"] = torch.zeros(3, 3)
    del sd["bbox_head.tasks.0.conv_cls.bias"]
    del sd["neck.blocks.0.2.num_batches_tracked"]  # ignored like the reference does
    rep = checkpoint.load_state_dict(b, sd, strict=False)
    assert rep["unexpected"] == ["bogus.weight"] and rep["missing"] == ["bbox_head.tasks.0.conv_cls.bias"]
    assert rep["mismatched"][0][0] == "neck.blocks.0.1.weight"
    with pytest.raises(RuntimeError):
        checkpoint.load_state_dict(b, sd, strict=True)
    with pytest.raises(IOError):
        checkpoint.load_checkpoint(b, str(tmp_path / "missing.pth"))
    with pytest.raises(IOError):
        checkpoint.load_checkpoint(b, "https://example.com/x.pth")


def _randomise_bn(model, seed=0):
    g = torch.Generator().manual_seed(seed)
    for m in model.modules():
        if isinstance(m, nn.modules.batchnorm._BatchNorm):
            m.running_mean.copy_(torch.randn(m.running_mean.shape, generator=g) * 0.3)
            m.running_var.copy_(torch.rand(m.running_var.shape, generator=g) + 0.5)
            m.weight.data.copy_(torch.rand(m.weight.shape, generator=g) + 0.5)
            m.bias.data.copy_(torch.randn(m.bias.shape, generator=g) * 0.2)


@pytest.mark.parametrize("strides", [([1], [1]), ([2, 2, 2], [0.5, 1, 2])])
def test_fold_bn_rpn_matches_unfolded_eval_output(strides):
    torch.manual_seed(1)
    ds, us = strides
    n = len(ds)
    rpn = RPN(layer_nums=[2] * n, ds_layer_strides=ds, ds_num_filters=[16 * (i + 1) for i in range(n)],
              us_layer_strides=us, us_num_filters=[24] * n, num_input_features=8).eval()
    _randomise_bn(rpn)
    x = torch.randn(2, 8, 32, 24)
    with torch.no_grad():
        want = rpn(x)
    n_bn = sum(isinstance(m, nn.BatchNorm2d) for m in rpn.modules())
    assert checkpoint.fuse_bn_recursively(rpn) == n_bn > 0
    assert not any(isinstance(m, nn.BatchNorm2d) for m in rpn.modules())
    with torch.no_grad():
        got = rpn(x)
    assert torch.allclose(got, want, atol=1e-5, rtol=1e-5)


def test_fold_bn_sparse_layers_matches_oracle_conv_plus_bn():
    """sparse conv + BatchNorm1d folding checked with the CPU oracle's sparse convolution
    (oracle/spconv_ref.py): BN(conv(x; W)) == conv(x; W', b')."""
    from det3d_amd.backbones import SpMiddleResNetFHD
    from oracle import spconv_ref

    torch.manual_seed(2)
    net = SpMiddleResNetFHD(num_input_features=5).eval()
    _randomise_bn(net, 3)
    convs = [m for m in net.modules() if isinstance(m, spconv.SparseConvolution)]
    before = {id(c): (c.weight.detach().clone(), None if c.bias is None else c.bias.detach().clone()) for c in convs}
    # (conv, bn) pairs as the forward applies them: neighbours in SparseSequential, conv1/bn1, conv2/bn2
    pairs = []
    mods = list(net.middle_conv._modules.values())
    for m0, m1 in zip(mods[:-1], mods[1:]):
        if isinstance(m0, spconv.SparseConvolution) and isinstance(m1, nn.BatchNorm1d):
            pairs.append((m0, m1))
    for blk in net.modules():
        if type(blk).__name__ == "SparseBasicBlock":
            pairs += [(blk.conv1, blk.bn1), (blk.conv2, blk.bn2)]
    n_bn = sum(isinstance(m, nn.BatchNorm1d) for m in net.modules())
    assert len(pairs) == n_bn
    bns = {id(c): bn for c, bn in pairs}
    assert checkpoint.fuse_bn_recursively(net) == n_bn
    assert not any(isinstance(m, nn.BatchNorm1d) for m in net.modules())
    rng = np.random.default_rng(0)
    for conv in (convs[0], convs[3], convs[-1]):
        w0, b0 = before[id(conv)]
        bn = bns[id(conv)]
        k = list(w0.shape[:3])
        cin = w0.shape[3]
        shape = [6, 9, 8]
        cells = rng.choice(np.prod(shape), 120, replace=False)
        idx = np.stack([np.zeros_like(cells)] + list(np.unravel_index(cells, shape)), 1).astype(np.int32)
        feats = torch.randn(len(idx), cin)
        subm = conv.subm
        stride, pad = ([1, 1, 1], [x // 2 for x in k]) if subm else (list(conv.stride), list(conv.padding))
        out_idx, _, pr = spconv_ref.get_indice_pairs(idx, 1, shape, k, stride, pad, subm)
        n_out = len(out_idx)
        y = spconv_ref.indice_conv(feats, w0, pr, n_out, bias=b0)
        want = nn.functional.batch_norm(y, bn.running_mean, bn.running_var, bn.weight, bn.bias, False, 0.0, bn.eps)
        got = spconv_ref.indice_conv(feats, conv.weight.detach(), pr, n_out, bias=conv.bias.detach())
        assert torch.allclose(got, want, atol=2e-5, rtol=1e-4)


class _ToyDetector(nn.Module):
    """losses-dict interface of the detectors (forward(example, return_loss=True) -> {"loss": [...]})"""

    def __init__(self):
        super().__init__()
        self.net = nn.Sequential(nn.Linear(6, 16), nn.Tanh(), nn.Linear(16, 3))

    def forward(self, example, return_loss=True):
        return {"loss": [(self.net(example["x"]) - example["y"]).pow(2).mean()]}


def test_train_step_resume_continues_the_same_trajectory(tmp_path):
    """save after 3 steps (model + optimizer + schedule position), resume in fresh objects, 2 more steps ==
    5 uninterrupted steps: one-cycle lr / momentum, true weight decay and Adam moments all carry over."""
    from det3d_amd.trainer import TrainStep

    torch.manual_seed(0)
    data = [dict(x=torch.randn(8, 6), y=torch.randn(8, 3)) for _ in range(5)]

    def fresh():
        torch.manual_seed(1)
        m = _ToyDetector()
        return m, TrainStep(m, total_steps=5, lr_max=1e-2, wd=0.01, grad_clip=35.0)

    m_ref, t_ref = fresh()
    ref_losses = [float(t_ref(d)) for d in data]
    m_a, t_a = fresh()
    for d in data[:3]:
        t_a(d)
    f = str(tmp_path / "resume.pth")
    checkpoint.save_checkpoint(m_a, f, meta=dict(train_step=t_a.state_dict()))
    m_b, t_b = fresh()
    with torch.no_grad():
        for p in m_b.parameters():
            p.add_(1.0)  # make sure the loaded state is what counts
    ck = checkpoint.load_checkpoint(m_b, f, map_location="cpu", strict=True)
    t_b.load_state_dict(ck["meta"]["train_step"])
    assert t_b.global_step == 3
    resumed = [float(t_b(d)) for d in data[3:]]
    assert resumed == pytest.approx(ref_losses[3:], rel=1e-6)
    for p, q in zip(m_ref.parameters(), m_b.parameters()):
        assert torch.allclose(p, q, atol=1e-7)
