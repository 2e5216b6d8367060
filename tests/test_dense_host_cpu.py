"""Host logic of the dense conv path that needs no GPU: the re-arrangements and group sizes of the kernel == stride layers
(dense_conv.ks_conv, reference det3d/models/necks/rpn.py:83-100), which layers the containers hand to the library's kernels
(dense_stack._conv_ok, nn_utils.Sequential), and that CPU maps never take those paths."""
import torch
from torch import nn

from det3d_amd import dense_conv as dc
from det3d_amd import dense_stack, nn_utils


def test_space_to_depth_and_back_are_the_k_equals_s_convolutions():
    """_s2d + a [cout, s*s*cin] matrix product IS Conv2d(k = s); a [s*s*cout, cin] product + _d2s IS ConvTranspose2d(k = s)
    -- with the weight re-arrangements ks_conv uses, in float64 on the CPU"""
    torch.manual_seed(0)
    for s in (2, 4):
        b, h, w, cin, cout = 2, 8, 12, 5, 3
        x = torch.randn(b, cin, h, w, dtype=torch.float64)
        rows = x.permute(0, 2, 3, 1).contiguous()
        xs = dc._s2d(rows, s)
        assert xs.shape == (b, h // s, w // s, s * s * cin)
        assert torch.equal(dc._d2s(xs, s), rows)
        conv = nn.Conv2d(cin, cout, s, stride=s, bias=False).double()
        w2 = conv.weight.detach().permute(0, 2, 3, 1).reshape(cout, s * s * cin)
        y = xs @ w2.t()
        assert torch.allclose(y.permute(0, 3, 1, 2), conv(x), atol=1e-12)
        dec = nn.ConvTranspose2d(cin, cout, s, stride=s, bias=False).double()
        wt = dec.weight.detach()  # [cin, cout, a, b]
        w3 = wt.permute(2, 3, 1, 0).reshape(s * s * cout, cin)
        o = rows @ w3.t()  # [B, H, W, (a, b, co)]
        assert torch.allclose(dc._d2s(o, s).permute(0, 3, 1, 2), dec(x), atol=1e-12)


def test_row_phase_groups_stay_within_the_widest_kernel():
    # ConvTranspose2d(k = s): a launch covers `ga` of the s row phases, ga * s * cout <= 512 channels, ga | s
    assert dc._ks_groups(2, 128) == 2 and dc._ks_groups(2, 256) == 1 and dc._ks_groups(4, 128) == 1
    assert dc._ks_groups(2, 64) == 2 and dc._ks_groups(4, 32) == 4 and dc._ks_groups(4, 64) == 2
    for s in (2, 3, 4):
        for cout in (32, 64, 128, 256):
            ga = dc._ks_groups(s, cout)
            assert s % ga == 0 and (ga == 1 or ga * s * cout <= 512)


def test_cpu_maps_and_odd_layers_keep_the_module_path():
    m = nn.ConvTranspose2d(256, 128, 2, stride=2, bias=False)
    x = torch.randn(1, 256, 4, 4).contiguous(memory_format=torch.channels_last)
    assert not dc.ks_eligible(m, x)  # not a CUDA map
    assert not dc.ks_eligible(nn.Conv2d(64, 128, 3, stride=2, bias=False), x)  # kernel != stride
    seq = nn_utils.Sequential(m, nn.BatchNorm2d(128), nn.ReLU())
    y = seq(x)  # the container must fall through to the modules on the CPU
    assert y.shape == (1, 128, 8, 8)
    layers, nxt = dense_stack.collect(list(seq._modules.values()), 0, x)
    assert layers == [] and nxt == 0


def test_which_convolutions_the_dense_stack_takes():
    """_conv_ok: 'same' stride-1 layers of 128 / 256 channels (64 in split mode), 1x1 layers, and the stride-2 3x3 first layer
    of a block with 128 / 256 channels on both sides; everything else -> None (module path)"""
    ok = dense_stack._conv_ok
    c = lambda *a, **k: nn.Conv2d(*a, bias=False, **k)
    assert ok(c(128, 128, 3, padding=1), 0) == 1 and ok(c(256, 128, 3), 1) == 1  # ZeroPad2d(1) folded in
    assert ok(c(128, 256, 1), 0) == 0
    assert ok(c(64, 64, 3, padding=1), 0) is None and ok(c(64, 64, 3, padding=1), 0, split=True) == 1
    assert ok(c(128, 128, 3), 0) is None  # not a 'same' convolution
    assert ok(c(128, 256, 3, stride=2), 1) == 1 and ok(c(256, 256, 3, stride=2, padding=1), 0) == 1
    assert ok(c(64, 128, 3, stride=2), 1, split=True) is None and ok(c(64, 64, 3, stride=2), 1, split=True) is None
    assert ok(c(128, 256, 2, stride=2), 0) is None  # kernel == stride layers go through dense_conv.ks_conv
    assert ok(nn.Conv2d(128, 128, 3, padding=1, bias=True), 0) is None
    assert ok(c(128, 128, 3, padding=1, dilation=2), 0) is None
    assert dc.dgrad_s2_supported(128, 256, 3) and not dc.dgrad_s2_supported(64, 64, 3) and not dc.dgrad_s2_supported(128, 256, 1)


def test_default_switches_of_the_dense_backward():
    """Round 6: the BatchNorm-backward sums of the split 3x3 layers ride in the data gradient's epilogue by default
    (dense_stack.FUSE_BN_BACKWARD; the reference has no such choice -- cuDNN's batch-norm backward, det3d/models/necks/rpn.py
    through autograd); D3D_DENSE_FUSE_BN_BWD=0 is the switch back to the stand-alone d3d_bn_bwd_reduce sweep."""
    import importlib
    import os

    saved = os.environ.pop("D3D_DENSE_FUSE_BN_BWD", None)
    try:
        importlib.reload(dense_stack)
        assert dense_stack.FUSE_BN_BACKWARD is True
        os.environ["D3D_DENSE_FUSE_BN_BWD"] = "0"
        importlib.reload(dense_stack)
        assert dense_stack.FUSE_BN_BACKWARD is False
    finally:
        if saved is None:
            os.environ.pop("D3D_DENSE_FUSE_BN_BWD", None)
        else:
            os.environ["D3D_DENSE_FUSE_BN_BWD"] = saved
        importlib.reload(dense_stack)
