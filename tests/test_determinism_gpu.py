"""Round 3: every reduction of the training step is summed in a fixed order (csrc/d3d_fold.h) -- BatchNorm
statistics, BatchNorm-backward sums (stand-alone sweeps and conv epilogues), weight gradients.  Two identical
launches / steps must therefore be BIT-equal, which is what lets the other GPU tests compare paths tightly instead
of against the step's own run-to-run noise."""
import copy
import os

import pytest
import torch

from conftest import ROOT

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("n,c", [(777, 16), (43153, 32), (126683, 64), (140800, 128), (70400, 256)])
def test_bn_reductions_are_bit_reproducible_and_right(dtype, n, c):
    from det3d_amd import _lib

    L = _lib.lib()
    dt = _lib.D3D_BF16 if dtype == torch.bfloat16 else _lib.D3D_F32
    g = torch.Generator(device="cuda").manual_seed(n + c)
    x = (torch.randn((n, c), device="cuda", generator=g) * 1.7 + 0.4).to(dtype)
    dy = torch.randn((n, c), device="cuda", generator=g).to(dtype)
    gamma = torch.rand(c, device="cuda", generator=g) + 0.5
    beta = torch.randn(c, device="cuda", generator=g) * 0.2
    st = _lib.stream_ptr(x.device)
    nws = L.d3d_bn_reduce_ws_floats(n, c, dt)
    assert nws >= 2 * c
    runs = []
    for rep in range(4):
        # a fresh, garbage-filled workspace every time: nothing may depend on its previous content
        ws = torch.full((nws,), float("nan"), device="cuda")
        _lib.check(L.d3d_bn_stats(x.data_ptr(), n, c, dt, ws.data_ptr(), st), "stats")
        runs.append(ws[:2 * c].clone())
        if rep == 1:  # load on the other CUs while the last blocks fold: the order of arrival changes
            torch.mm(torch.randn(2048, 2048, device="cuda"), torch.randn(2048, 2048, device="cuda"))
    for r in runs[1:]:
        assert torch.equal(r, runs[0])
    xf = x.double()
    ref = torch.cat([xf.sum(0), (xf * xf).sum(0)])
    assert torch.allclose(runs[0].double(), ref, rtol=2e-5, atol=1e-3)
    mean = (ref[:c] / n).float()
    var = (ref[c:] / n - (ref[:c] / n) ** 2).clamp_min(0).float()
    invstd = torch.rsqrt(var + 1e-3)
    red = []
    for rep in range(3):
        ws = torch.full((nws,), float("nan"), device="cuda")
        _lib.check(L.d3d_bn_bwd_reduce(x.data_ptr(), dy.data_ptr(), n, c, dt, gamma.data_ptr(), beta.data_ptr(),
                                       mean.data_ptr(), invstd.data_ptr(), 1, ws.data_ptr(), st), "bwd_reduce")
        red.append(ws[:2 * c].clone())
    assert torch.equal(red[0], red[1]) and torch.equal(red[0], red[2])
    # the reference evaluates the ReLU mask with the kernel's own arithmetic, z = fmaf(x, g * invstd, fmaf(-mean * invstd,
    # g, b)): an f32 product is exact in f64 and the rounding of the sum keeps its sign, so the f64 expression has the
    # sign of the kernel's fmaf
    mi = -mean * invstd
    zs = gamma * invstd
    zb = (mi.double() * gamma.double() + beta.double()).float()
    mask = (x.float().double() * zs.double() + zb.double()) > 0
    xh = x.float().double() * invstd.double() + mi.double()
    dz = dy.float().double() * mask
    ref = torch.cat([(dz * xh).sum(0), dz.sum(0)])
    assert float((red[0].double() - ref).abs().max()) <= 2e-5 * float(ref.abs().max()) + 1e-3  # f32 accumulation


def _model(name="kitti_car_second.py"):
    from det3d_amd import Config, build_detector
    from oracle import model_ref as M

    cfg = Config.fromfile(os.path.join(ROOT, "det3d_amd", "configs", name))
    model = build_detector(cfg.model, cfg.train_cfg, cfg.test_cfg)
    M.fill_deterministic(model)
    return cfg, model.cuda().train()


def _step(model, ex, dtype):
    model.zero_grad(set_to_none=True)
    ctx = torch.autocast("cuda", dtype=dtype) if dtype is not None else torch.autocast("cuda", enabled=False)
    with ctx:
        losses = model(ex, return_loss=True)
    loss = sum(losses["loss"])
    loss.backward()
    return float(loss.detach()), {k: p.grad.clone() for k, p in model.named_parameters() if p.grad is not None}


@pytest.mark.parametrize("dtype", [None, torch.bfloat16])
def test_two_identical_training_steps_are_bit_equal(dtype):
    """SECOND (14 sparse conv + BN + ReLU layers, RPN, head, loss): same weights, same batch, twice -- on two copies of
    the model and twice on the same one.  Everything this repo launches is deterministic; MIOpen's convolution
    kernels are asked for their deterministic solvers."""
    from det3d_amd import synthetic

    cfg, a = _model()
    b = copy.deepcopy(a)
    ds = synthetic.KittiCarSynthetic(n_points=9000, n_seeds=4)
    dev = torch.device("cuda", 0)
    ex = synthetic.make_example(ds.device_batch(0, 2, dev), synthetic.KITTI_CAR)
    with torch.backends.cudnn.flags(enabled=True, benchmark=False, deterministic=True):
        la, ga = _step(a, ex, dtype)
        lb, gb = _step(b, ex, dtype)
        la2, ga2 = _step(a, ex, dtype)  # running statistics moved, the batch statistics did not
    assert ga.keys() == gb.keys()
    bad = [k for k in ga if not torch.equal(ga[k], gb[k])]
    assert not bad, "not bit-equal across two copies: %s" % bad[:6]
    bad = [k for k in ga if not torch.equal(ga[k], ga2[k])]
    assert not bad, "not bit-equal across two steps of one model: %s" % bad[:6]


@pytest.mark.parametrize("config", ["kitti_car_second.py", "nusc_cbgs.py"])
def test_weight_gradients_on_the_second_stream_change_no_bit(config):
    """sparse_stack.WGRAD_STREAM (round 5): the weight-gradient kernels of the fused sparse stack, the fused residual
    blocks and the dense RPN stack run on a second stream next to the data-gradient chain.  Same kernels, same arguments:
    loss and every gradient of a step must equal the one-stream step bit for bit (and a racing read of an unfinished
    buffer would show up here as a difference)."""
    from det3d_amd import sparse_stack, synthetic

    cfg, a = _model(config)
    b = copy.deepcopy(a)
    dev = torch.device("cuda", 0)
    if config.startswith("kitti"):
        ds = synthetic.KittiCarSynthetic(n_points=20000, n_seeds=4)
        ex = synthetic.make_example(ds.device_batch(0, 4, dev), synthetic.KITTI_CAR)
    else:
        ds = synthetic.TaskGroupSynthetic(cfg, n_points=30000, n_seeds=2)
        ex = ds.make_example(ds.device_batch(0, 2, dev))
    prev = sparse_stack.WGRAD_STREAM
    try:
        with torch.backends.cudnn.flags(enabled=True, benchmark=False, deterministic=True):
            sparse_stack.WGRAD_STREAM = True
            la, ga = _step(a, ex, None)
            la2, ga2 = _step(a, ex, None)
            sparse_stack.WGRAD_STREAM = False
            lb, gb = _step(b, ex, None)
    finally:
        sparse_stack.WGRAD_STREAM = prev
    assert la == lb == la2 and ga.keys() == gb.keys()
    bad = [k for k in ga if not (torch.equal(ga[k], gb[k]) and torch.equal(ga[k], ga2[k]))]
    assert not bad, "second stream changed: %s" % bad[:6]


@pytest.mark.parametrize("config", ["nusc_cbgs.py", "lyft_cbgs.py", "nusc_point_pillars.py"])
def test_two_identical_steps_of_the_multi_task_configs_are_bit_equal(config):
    """The same for the CBGS (SpMiddleResNetFHD: fused residual blocks, 128-channel sparse layers; rpn2 neck; 5-6 task
    groups as one stacked head convolution) and the PointPillars configs (PillarFeatureNet, 64-channel dense layers):
    two copies of the model on one batch, f32 / split arithmetic.  The strided and transposed convolutions of the
    necks are still MIOpen's -- asked for their deterministic solvers."""
    from det3d_amd import synthetic

    cfg, a = _model(config)
    b = copy.deepcopy(a)
    dev = torch.device("cuda", 0)
    ds = synthetic.TaskGroupSynthetic(cfg, n_points=30000, n_seeds=2)
    ex = ds.make_example(ds.device_batch(0, 2, dev))
    with torch.backends.cudnn.flags(enabled=True, benchmark=False, deterministic=True):
        la, ga = _step(a, ex, None)
        lb, gb = _step(b, ex, None)
    assert la == lb and ga.keys() == gb.keys()
    bad = [k for k in ga if not torch.equal(ga[k], gb[k])]
    assert not bad, "not bit-equal across two copies: %s" % bad[:6]


def test_conv_epilogue_reduction_refuses_a_channel_pair_without_mfma_tiles():
    """(128, 16), (48, 48) ... pass the old `% 16` test but have no MFMA tiling; the fused reduction
    must be refused loudly (and Python must not ask for it) instead of leaving all-zero statistics behind."""
    import ctypes

    from det3d_amd import _lib, spconv

    L = _lib.lib()
    assert not spconv.has_epilogue(128, 16, torch.bfloat16) and not spconv.has_epilogue(48, 48, torch.float32)
    assert not spconv.has_epilogue(64, 16, torch.float32)
    assert spconv.has_epilogue(64, 64, torch.bfloat16) and spconv.has_epilogue(16, 32, torch.float32)
    n, cin, cout = 500, 48, 48
    feats = torch.randn(n, cin, device="cuda")
    w = torch.randn(1, cout, cin, device="cuda")
    nbr = torch.arange(n, device="cuda", dtype=torch.int32)
    out = torch.empty(n, cout, device="cuda")
    ws = torch.empty(L.d3d_conv_epilogue_ws_floats(n, cout), device="cuda")
    ep = _lib.ConvEpilogue(stats=ws.data_ptr(), ws_floats=ws.numel())
    rc = L.d3d_sparse_conv_fwd_ex(feats.data_ptr(), w.data_ptr(), None, nbr.data_ptr(), n, n, 1, cin, cout, _lib.D3D_F32,
                                  out.data_ptr(), ctypes.byref(ep), _lib.stream_ptr(feats.device))
    assert rc == -1 and b"d3d_sparse_conv_has_epilogue" in L.d3d_last_error()
    # ... and the module path falls back to the stand-alone statistics sweep with correct results
    torch.manual_seed(0)
    conv = spconv.SubMConv3d(64, 16, 3, bias=False, indice_key="t").cuda()  # SpMiddleFHD(num_input_features=64)
    bn = torch.nn.BatchNorm1d(16, eps=1e-3, momentum=0.01).cuda().train()
    idx = torch.stack([torch.zeros(300, dtype=torch.int32), torch.randint(0, 8, (300,), dtype=torch.int32),
                       torch.randint(0, 16, (300,), dtype=torch.int32), torch.randint(0, 16, (300,), dtype=torch.int32)], 1)
    idx = torch.unique(idx, dim=0).cuda()
    x = spconv.SparseConvTensor(torch.randn(idx.shape[0], 64, device="cuda"), idx, [8, 16, 16], 1)
    y, _ = spconv.conv_bn_relu(conv, bn, True, x)
    ref = torch.relu(torch.nn.functional.batch_norm(conv(x).features, None, None, bn.weight, bn.bias, True, 0.0, 1e-3))
    assert torch.allclose(y.features, ref, atol=1e-4, rtol=1e-4)
