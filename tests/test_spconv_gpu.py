"""HIP rulebooks + sparse conv (fwd / dgrad / wgrad) vs oracle/spconv_ref.py.
Rulebooks: bit-exact as canonical sorted sets.  Features: f32
path within 1e-4 (fp32 reassociation), bf16 path within the bf16 budget stated inline."""
import numpy as np
import pytest
import torch

from oracle import spconv_ref as R

pytestmark = pytest.mark.gpu


def _random_sparse(rng, batch, shape, n_per, c, clustered=True):
    idx = []
    for b in range(batch):
        if clustered:
            ctr = rng.integers(0, [shape[0], shape[1], shape[2]], size=(max(n_per // 12, 1), 3))
            pts = ctr[rng.integers(0, len(ctr), n_per * 2)] + rng.integers(-2, 3, size=(n_per * 2, 3))
            pts = np.clip(pts, 0, np.array(shape) - 1)
            pts = np.unique(pts, axis=0)
            rng.shuffle(pts)
            pts = pts[:n_per]
        else:
            lin = rng.choice(shape[0] * shape[1] * shape[2], size=n_per, replace=False)
            pts = np.stack([lin // (shape[1] * shape[2]), (lin // shape[2]) % shape[1], lin % shape[2]], 1)
        idx.append(np.concatenate([np.full((len(pts), 1), b), pts], 1))
    idx = np.concatenate(idx).astype(np.int32)
    feats = rng.standard_normal((len(idx), c)).astype(np.float32)
    return idx, feats


GEOMS = [
    ([3, 3, 3], [1, 1, 1], [0, 0, 0], True),
    ([3, 3, 3], [2, 2, 2], [1, 1, 1], False),
    ([3, 3, 3], [2, 2, 2], [0, 1, 1], False),
    ([3, 1, 1], [2, 1, 1], [0, 0, 0], False),
    ([1, 1, 1], [1, 1, 1], [0, 0, 0], True),
]


def _build(idx, batch, shape, ksize, stride, padding, subm, count_pairs=True):
    from det3d_amd import spconv

    ind = torch.from_numpy(idx).cuda()
    n = len(idx)
    n_dev = torch.tensor([n], dtype=torch.int32, device="cuda")
    if subm:
        return spconv.build_subm_rulebook(ind, n_dev, n, max(n, 1), batch, shape, ksize, count_pairs=count_pairs)
    rb, n_out = spconv.build_conv_rulebook(ind, n_dev, n, max(n, 1), batch, shape, ksize, stride, padding,
                                           count_pairs=count_pairs)
    return spconv.finalize_conv_rulebook(rb, n, int(n_out.item()))


@pytest.mark.parametrize("ksize,stride,padding,subm", GEOMS)
def test_rulebook_bit_exact(ksize, stride, padding, subm):
    rng = np.random.default_rng(1)
    shape = [11, 40, 36]
    idx, _ = _random_sparse(rng, 3, shape, 900, 1)
    rb = _build(idx, 3, shape, ksize, stride, padding, subm)
    out_idx, out_shape, pairs = R.get_indice_pairs(idx, 3, shape, ksize, stride, padding, subm)
    assert rb.n_out == len(out_idx)
    assert np.array_equal(rb.out_indices.cpu().numpy(), out_idx)  # sorted order for strided
    assert list(rb.out_shape) == list(out_shape)
    ref = R.canonical_pairs(pairs)
    assert np.array_equal(rb.canonical_pairs(), ref)
    assert rb.pairs() == len(ref)
    pc = rb.pair_count.cpu().numpy()
    assert pc.tolist() == [len(p[0]) for p in pairs]           # in-kernel counters
    lazy = _build(idx, 3, shape, ksize, stride, padding, subm, count_pairs=False)
    assert lazy.pair_count is None
    assert lazy.pair_counts().cpu().tolist() == pc.tolist()     # counted on demand from the table
    if not subm:  # input-stationary twin table describes the same pair set
        t = rb.nbr_in.view(rb.kvol, rb.cap_in)[:, : rb.n_in].cpu().numpy()
        k, i = np.nonzero(t >= 0)
        trip = np.stack([k, i, t[k, i]], 1)
        trip = trip[np.lexsort((trip[:, 2], trip[:, 1], trip[:, 0]))]
        assert np.array_equal(trip, ref)


def test_rulebook_kitti_scale_and_determinism():
    """full KITTI grid [41,1600,1408], batch 4: exercises the 46 MB bitmap path."""
    rng = np.random.default_rng(3)
    shape = [41, 1600, 1408]
    idx, _ = _random_sparse(rng, 4, shape, 12000, 1)
    a = _build(idx, 4, shape, [3, 3, 3], [2, 2, 2], [1, 1, 1], False)
    out_idx, _, pairs = R.get_indice_pairs(idx, 4, shape, [3, 3, 3], [2, 2, 2], [1, 1, 1], False)
    assert np.array_equal(a.out_indices.cpu().numpy(), out_idx)
    ref = R.canonical_pairs(pairs)
    assert np.array_equal(a.canonical_pairs(), ref)
    b = _build(idx, 4, shape, [3, 3, 3], [2, 2, 2], [1, 1, 1], False)
    assert torch.equal(a.nbr_out[: a.kvol * a.cap_out], b.nbr_out[: b.kvol * b.cap_out]) or \
        np.array_equal(a.canonical_pairs(), b.canonical_pairs())
    s = _build(idx, 4, shape, [3, 3, 3], [1, 1, 1], [0, 0, 0], True)
    _, _, sp = R.get_indice_pairs(idx, 4, shape, [3, 3, 3], [1, 1, 1], [0, 0, 0], True)
    assert np.array_equal(s.canonical_pairs(), R.canonical_pairs(sp))


CHANNELS = [(4, 16), (5, 16), (16, 16), (16, 32), (32, 32), (32, 64), (64, 64), (64, 128), (128, 128), (3, 7)]


@pytest.mark.parametrize("cin,cout", CHANNELS)
@pytest.mark.parametrize("gi", [0, 1, 3])
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_conv_fwd_bwd_vs_oracle(cin, cout, gi, dtype):
    from det3d_amd import spconv

    ksize, stride, padding, subm = GEOMS[gi]
    rng = np.random.default_rng(cin * 131 + cout + gi)
    shape = [9, 30, 28]
    idx, feats = _random_sparse(rng, 2, shape, 700, cin)
    w = (rng.standard_normal((*ksize, cin, cout)) / np.sqrt(cin * 9)).astype(np.float32)
    bias = rng.standard_normal(cout).astype(np.float32) if gi == 0 else None
    out_idx, out_shape, pairs = R.get_indice_pairs(idx, 2, shape, ksize, stride, padding, subm)
    g = rng.standard_normal((len(out_idx), cout)).astype(np.float32)

    # quantise inputs the same way for both sides so the comparison isolates the kernel
    q = (lambda a: torch.from_numpy(a).to(dtype).float()) if dtype == torch.bfloat16 else torch.from_numpy
    f_ref = q(feats).clone().requires_grad_(True)
    w_ref = q(w).clone().requires_grad_(True)
    b_ref = torch.from_numpy(bias).requires_grad_(True) if bias is not None else None
    out_ref = R.indice_conv(f_ref, w_ref, pairs, len(out_idx), b_ref)
    (out_ref * q(g)).sum().backward()

    cls = spconv.SubMConv3d if subm else spconv.SparseConv3d
    m = cls(cin, cout, ksize, stride=stride, padding=padding, bias=bias is not None).cuda()
    with torch.no_grad():
        m.weight.copy_(q(w))
        if bias is not None:
            m.bias.copy_(torch.from_numpy(bias))
    f = q(feats).cuda().to(dtype).requires_grad_(True)
    x = spconv.SparseConvTensor(f, torch.from_numpy(idx).cuda(), shape, 2)
    y = m(x)
    assert np.array_equal(y.indices.cpu().numpy(), out_idx)
    (y.features.float() * q(g).cuda()).sum().backward()

    if dtype == torch.float32:
        tol = dict(atol=2e-4, rtol=2e-4)
        gtol = dict(atol=5e-4, rtol=5e-4)
    else:
        # bf16 output rounding: 2^-8 relative on O(1) values; wgrad sums ~1e3 products of
        # bf16-rounded grads
        tol = dict(atol=3e-2, rtol=2e-2)
        gtol = dict(atol=6e-2, rtol=3e-2)
    assert torch.allclose(y.features.float().cpu(), out_ref.detach(), **tol)
    assert torch.allclose(f.grad.float().cpu(), f_ref.grad, **gtol)
    assert torch.allclose(m.weight.grad.cpu(), w_ref.grad, **gtol)
    if bias is not None:
        assert torch.allclose(m.bias.grad.cpu(), b_ref.grad, **gtol)


def test_conv_vs_dense_conv3d_cropped_grid():
    """independent of the restatement: HIP f32 path vs F.conv3d on the densified crop."""
    from det3d_amd import spconv

    rng = np.random.default_rng(8)
    shape = [8, 24, 20]
    idx, feats = _random_sparse(rng, 2, shape, 500, 16)
    for ksize, stride, padding, subm in GEOMS[:3]:
        w = (rng.standard_normal((*ksize, 16, 32)) / 12).astype(np.float32)
        ref, ref_idx = R.dense_conv_reference(torch.from_numpy(feats), idx, 2, shape, torch.from_numpy(w),
                                              ksize, stride, padding, subm)
        cls = spconv.SubMConv3d if subm else spconv.SparseConv3d
        m = cls(16, 32, ksize, stride=stride, padding=padding, bias=False).cuda()
        with torch.no_grad():
            m.weight.copy_(torch.from_numpy(w))
        y = m(spconv.SparseConvTensor(torch.from_numpy(feats).cuda(), torch.from_numpy(idx).cuda(), shape, 2))
        assert np.array_equal(y.indices.cpu().numpy(), ref_idx)
        assert torch.allclose(y.features.cpu(), ref, atol=2e-4, rtol=2e-4)


def test_dense_and_its_backward():
    from det3d_amd import spconv

    rng = np.random.default_rng(2)
    shape = [2, 20, 18]
    idx, feats = _random_sparse(rng, 3, shape, 200, 64, clustered=False)
    for dtype in (torch.float32, torch.bfloat16):
        f = torch.from_numpy(feats).cuda().to(dtype).requires_grad_(True)
        x = spconv.SparseConvTensor(f, torch.from_numpy(idx).cuda(), shape, 3)
        d = x.dense()
        ref = R.dense(torch.from_numpy(feats).to(dtype).float(), idx, 3, shape)
        assert d.shape == (3, 64, 2, 20, 18)
        assert torch.equal(d.float().cpu(), ref)
        g = torch.randn_like(d)
        d.backward(g)
        ii = torch.from_numpy(idx).long()
        assert torch.equal(f.grad.cpu(), g.cpu()[ii[:, 0], :, ii[:, 1], ii[:, 2], ii[:, 3]])


def test_dense_2d_channels_last_equals_nchw():
    from det3d_amd import spconv

    rng = np.random.default_rng(12)
    shape = [2, 24, 20]
    idx, feats = _random_sparse(rng, 3, shape, 300, 64, clustered=False)
    for dtype in (torch.float32, torch.bfloat16):
        outs, grads = [], []
        for cl in (False, True):
            f = torch.from_numpy(feats).cuda().to(dtype).requires_grad_(True)
            x = spconv.SparseConvTensor(f, torch.from_numpy(idx).cuda(), shape, 3)
            d = x.dense_2d(channels_last=cl)
            assert d.shape == (3, 128, 24, 20)
            assert d.is_contiguous(memory_format=torch.channels_last) == cl or not cl
            g = torch.arange(d.numel(), device="cuda", dtype=torch.float32).reshape(d.shape).to(dtype) / d.numel()
            d.backward(g)
            outs.append(d.detach().float().cpu().contiguous())
            grads.append(f.grad.float().cpu())
        assert torch.equal(outs[0], outs[1]) and torch.equal(grads[0], grads[1])


def test_plan_single_sync_matches_lazy_path():
    """SparseSequential.plan() (one host sync) builds the same rulebooks as the per-layer path."""
    from det3d_amd import spconv

    rng = np.random.default_rng(4)
    shape = [21, 80, 72]
    idx, feats = _random_sparse(rng, 2, shape, 3000, 16)
    torch.manual_seed(0)
    net = spconv.SparseSequential(
        spconv.SubMConv3d(16, 16, 3, bias=False, indice_key="s0"), torch.nn.ReLU(),
        spconv.SparseConv3d(16, 32, 3, 2, padding=1, bias=False),
        spconv.SubMConv3d(32, 32, 3, bias=False, indice_key="s1"),
        spconv.SubMConv3d(32, 32, 3, bias=False, indice_key="s1"),
        spconv.SparseConv3d(32, 32, 3, 2, padding=[0, 1, 1], bias=False),
        spconv.SparseConv3d(32, 32, (3, 1, 1), (2, 1, 1), bias=False)).cuda()
    mk = lambda: spconv.SparseConvTensor(torch.from_numpy(feats).cuda(), torch.from_numpy(idx).cuda(), shape, 2)
    a = net(mk())
    xb = mk()
    net.plan(xb)
    b = net(xb)
    assert torch.equal(a.indices, b.indices) and torch.equal(a.features, b.features)
    assert a.spatial_shape == b.spatial_shape == [2, 20, 18]


@pytest.mark.parametrize("c", [16, 32, 64, 128])
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("relu", [True, False])
def test_fused_bn_relu_vs_torch(c, dtype, relu):
    """d3d_bn_relu_fwd/bwd vs nn.BatchNorm1d(+ReLU) in f32 on the same (dtype-rounded) input:
    outputs, running statistics, input / weight / bias gradients."""
    from det3d_amd import spconv

    torch.manual_seed(c + relu)
    n = 5003
    x0 = (torch.randn(n, c) * torch.linspace(0.5, 2.0, c) + torch.linspace(-1, 1, c)).to(dtype)
    g = torch.randn(n, c).to(dtype)
    ref = torch.nn.BatchNorm1d(c, eps=1e-3, momentum=0.01)
    with torch.no_grad():
        ref.weight.uniform_(0.5, 1.5)
        ref.bias.uniform_(-0.5, 0.5)
    mine = torch.nn.BatchNorm1d(c, eps=1e-3, momentum=0.01).cuda()
    mine.load_state_dict(ref.state_dict())
    xr = x0.float().clone().requires_grad_(True)
    yr = ref(xr)
    yr = torch.relu(yr) if relu else yr
    (yr * g.float()).sum().backward()
    xm = x0.cuda().clone().requires_grad_(True)
    assert spconv._fusable_bn(mine, xm)
    ym = spconv.bn_relu(mine, xm, relu)
    assert ym.dtype == dtype
    (ym.float() * g.cuda().float()).sum().backward()
    tol = dict(atol=1e-4, rtol=1e-4) if dtype == torch.float32 else dict(atol=3e-2, rtol=2e-2)
    assert torch.allclose(ym.float().cpu(), yr.detach(), **tol)
    assert torch.allclose(mine.running_mean.cpu(), ref.running_mean, atol=1e-5, rtol=1e-4)
    assert torch.allclose(mine.running_var.cpu(), ref.running_var, atol=1e-5, rtol=1e-4)
    gtol = dict(atol=2e-3, rtol=2e-3) if dtype == torch.float32 else dict(atol=6e-2, rtol=5e-2)
    assert torch.allclose(xm.grad.float().cpu(), xr.grad, **gtol)
    wscale = float(ref.weight.grad.abs().max())
    assert torch.allclose(mine.weight.grad.cpu(), ref.weight.grad, atol=(2e-3 if dtype == torch.float32 else 2e-2) * wscale)
    assert torch.allclose(mine.bias.grad.cpu(), ref.bias.grad, atol=(2e-3 if dtype == torch.float32 else 2e-2) * float(ref.bias.grad.abs().max()))
    # eval mode uses the running statistics
    mine.eval(); ref.eval()
    with torch.no_grad():
        ye = spconv.bn_relu(mine, x0.cuda(), relu)
        yre = ref(x0.float())
        yre = torch.relu(yre) if relu else yre
    assert torch.allclose(ye.float().cpu(), yre, **tol)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("shape", [(2, 128, 50, 44), (1, 64, 31, 17), (4, 256, 25, 22)])
def test_fused_bn2d_relu_channels_last_vs_torch(shape, dtype):
    """The RPN's Conv2d -> BatchNorm2d -> ReLU triples on channels-last maps (necks.py through
    nn_utils.Sequential) use d3d_bn_relu_* on the [B*H*W, C] view: same outputs, running statistics and
    gradients as the stock modules; non-channels-last input keeps the stock path."""
    from det3d_amd import nn_utils, spconv

    torch.manual_seed(shape[1])
    b, c, h, w = shape
    x0 = (torch.randn(shape) * torch.linspace(0.5, 2.0, c).view(1, c, 1, 1)
          + torch.linspace(-1, 1, c).view(1, c, 1, 1)).to(dtype)
    g = torch.randn(shape).to(dtype)
    ref = torch.nn.Sequential(torch.nn.BatchNorm2d(c, eps=1e-3, momentum=0.01), torch.nn.ReLU())
    with torch.no_grad():
        ref[0].weight.uniform_(0.5, 1.5)
        ref[0].bias.uniform_(-0.5, 0.5)
    mine = nn_utils.Sequential(torch.nn.BatchNorm2d(c, eps=1e-3, momentum=0.01), torch.nn.ReLU()).cuda()
    mine.load_state_dict(ref.state_dict())
    xr = x0.float().clone().requires_grad_(True)
    yr = ref(xr)
    (yr * g.float()).sum().backward()
    xm = x0.cuda().contiguous(memory_format=torch.channels_last).requires_grad_(True)
    assert spconv.fusable_bn2d(mine[0], xm) and not spconv.fusable_bn2d(mine[0], x0.cuda())
    nn_utils.FUSE_BN2D_RELU = True  # opt-in path (default off: MIOpen's NHWC batch norm is faster on big maps)
    try:
        ym = mine(xm)
    finally:
        nn_utils.FUSE_BN2D_RELU = False
    assert ym.dtype == dtype and ym.shape == shape and ym.is_contiguous(memory_format=torch.channels_last)
    (ym.float() * g.cuda().float()).sum().backward()
    tol = dict(atol=1e-4, rtol=1e-4) if dtype == torch.float32 else dict(atol=3e-2, rtol=2e-2)
    assert torch.allclose(ym.float().cpu(), yr.detach(), **tol)
    assert torch.allclose(mine[0].running_mean.cpu(), ref[0].running_mean, atol=1e-5, rtol=1e-4)
    assert torch.allclose(mine[0].running_var.cpu(), ref[0].running_var, atol=1e-5, rtol=1e-4)
    gtol = dict(atol=2e-3, rtol=2e-3) if dtype == torch.float32 else dict(atol=6e-2, rtol=5e-2)
    assert torch.allclose(xm.grad.float().cpu(), xr.grad, **gtol)
    rel = 2e-3 if dtype == torch.float32 else 2e-2
    assert torch.allclose(mine[0].weight.grad.cpu(), ref[0].weight.grad, atol=rel * float(ref[0].weight.grad.abs().max()))
    assert torch.allclose(mine[0].bias.grad.cpu(), ref[0].bias.grad, atol=rel * float(ref[0].bias.grad.abs().max()))
    # default: the stock modules
    with torch.no_grad():
        y2 = mine(xm.detach())
    assert torch.allclose(y2.float(), ym.detach().float(), **tol)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_fused_bn_large_row_count_matches_torch(dtype):
    """140 800 x 128 rows (the KITTI RPN map): the large-input reduction sweeps (1024-thread blocks, one per CU,
    65 k instead of 524 k global atomics), the 4x unrolled sweeps and the replicated accumulators."""
    from det3d_amd import spconv

    torch.manual_seed(5)
    n, c = 140800, 128
    x = (torch.randn(n, c, device="cuda") * 1.5 + 0.25).to(dtype).float()
    g = torch.randn(n, c, device="cuda").to(dtype).float()
    ref = torch.nn.BatchNorm1d(c, eps=1e-3, momentum=0.01).cuda()
    mine = torch.nn.BatchNorm1d(c, eps=1e-3, momentum=0.01).cuda()
    xr = x.clone().requires_grad_(True)
    xm = x.clone().requires_grad_(True)
    yr = torch.relu(ref(xr))
    (yr * g).sum().backward()
    if dtype == torch.bfloat16:
        xm = x.clone().to(dtype).requires_grad_(True)
        ym = spconv.bn_relu(mine, xm, True)
        (ym.float() * g).sum().backward()
        assert torch.allclose(ym.float(), yr.detach(), atol=3e-2, rtol=2e-2)
        assert torch.allclose(xm.grad.float(), xr.grad, atol=3e-2, rtol=3e-2)
        assert torch.allclose(mine.weight.grad, ref.weight.grad, rtol=2e-2, atol=2e-2 * float(ref.weight.grad.abs().max()))
        assert torch.allclose(mine.running_var, ref.running_var, rtol=1e-3)
        return
    ym = spconv.bn_relu(mine, xm, True)
    (ym * g).sum().backward()
    assert torch.allclose(ym, yr.detach(), atol=1e-4, rtol=1e-4)
    assert torch.allclose(xm.grad, xr.grad, atol=2e-3, rtol=2e-3)
    assert torch.allclose(mine.weight.grad, ref.weight.grad, rtol=1e-3, atol=1e-3 * float(ref.weight.grad.abs().max()))
    assert torch.allclose(mine.running_var, ref.running_var, rtol=1e-4)


def test_sequential_fuses_bn_relu_transparently():
    from det3d_amd import spconv

    rng = np.random.default_rng(6)
    shape = [9, 30, 28]
    idx, feats = _random_sparse(rng, 2, shape, 600, 16)
    torch.manual_seed(1)
    net = spconv.SparseSequential(spconv.SubMConv3d(16, 32, 3, bias=False, indice_key="a"),
                                  torch.nn.BatchNorm1d(32, eps=1e-3, momentum=0.01), torch.nn.ReLU(),
                                  spconv.SubMConv3d(32, 32, 3, bias=False, indice_key="a"),
                                  torch.nn.BatchNorm1d(32, eps=1e-3, momentum=0.01), torch.nn.ReLU()).cuda()
    mk = lambda: spconv.SparseConvTensor(torch.from_numpy(feats).cuda(), torch.from_numpy(idx).cuda(), shape, 2)
    import copy
    net2 = copy.deepcopy(net)
    spconv.FUSE_BN_RELU = True
    a = net(mk()).features
    spconv.FUSE_BN_RELU = False
    try:
        b = net2(mk()).features
    finally:
        spconv.FUSE_BN_RELU = True
    assert torch.allclose(a, b, atol=1e-4, rtol=1e-4)
    assert torch.allclose(net[1].running_var, net2[1].running_var, atol=1e-6, rtol=1e-4)


@pytest.mark.parametrize("cin,cout,gi,n_per", [(64, 64, 0, 30000), (32, 64, 1, 24000), (128, 128, 0, 9000),
                                               (16, 16, 0, 30000), (64, 64, 3, 20000)])
def test_wgrad_dgrad_at_kitti_scale_vs_oracle_and_bit_reproducible(cin, cout, gi, n_per):
    """weight / data gradients pinned at toy sizes only.  ~60 k active sites (a KITTI batch level),
    bf16: the two-pass tile kernel (many site splits, all offset groups, deep prefetch pipeline) against the
    oracle's gather-mm-scatter, and the weight gradient must be bit-identical between two launches (no atomics)."""
    from det3d_amd import spconv

    ksize, stride, padding, subm = GEOMS[gi]
    rng = np.random.default_rng(cin + cout + gi)
    shape = [21, 200, 176]
    idx, feats = _random_sparse(rng, 2, shape, n_per, cin)
    w = (rng.standard_normal((*ksize, cin, cout)) / np.sqrt(cin * 9)).astype(np.float32)
    out_idx, out_shape, pairs = R.get_indice_pairs(idx, 2, shape, ksize, stride, padding, subm)
    g = rng.standard_normal((len(out_idx), cout)).astype(np.float32)
    q = lambda a: torch.from_numpy(a).to(torch.bfloat16).float()
    f_ref = q(feats).clone().requires_grad_(True)
    w_ref = q(w).clone().requires_grad_(True)
    out_ref = R.indice_conv(f_ref, w_ref, pairs, len(out_idx))
    (out_ref * q(g)).sum().backward()

    cls = spconv.SubMConv3d if subm else spconv.SparseConv3d
    m = cls(cin, cout, ksize, stride=stride, padding=padding, bias=False).cuda()
    with torch.no_grad():
        m.weight.copy_(q(w))
    grads = []
    for _ in range(2):
        m.weight.grad = None
        f = q(feats).cuda().to(torch.bfloat16).requires_grad_(True)
        y = m(spconv.SparseConvTensor(f, torch.from_numpy(idx).cuda(), shape, 2))
        (y.features.float() * q(g).cuda()).sum().backward()
        grads.append(m.weight.grad.clone())
    assert torch.equal(grads[0], grads[1]), "weight gradient differs between two identical launches"
    gw, gw_ref = grads[0].cpu(), w_ref.grad
    # sums of up to ~6e4 products of bf16-rounded factors accumulated in f32: compare against the tensor scale
    assert float((gw - gw_ref).abs().max()) < 2e-3 * float(gw_ref.abs().max()) + 1e-3
    assert torch.allclose(y.features.float().cpu(), out_ref.detach(), atol=3e-2, rtol=2e-2)
    assert torch.allclose(f.grad.float().cpu(), f_ref.grad, atol=6e-2, rtol=3e-2)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("cin0", [4, 16])
def test_conv_epilogue_bn_fusions_match_the_separate_sweeps(dtype, cin0):
    """d3d_sparse_conv_fwd_ex: BN batch statistics from the forward epilogue and BN-backward sums from the dgrad
    epilogue (spconv.FUSE_BN_STATS) vs the same stack with separate bn_stats / bn_bwd_reduce sweeps: activations,
    running statistics, input gradient and every parameter gradient.  cin0 = 4 also exercises the zero-padded
    stem layer."""
    import copy

    from det3d_amd import spconv

    rng = np.random.default_rng(9)
    shape = [11, 60, 56]
    idx, feats = _random_sparse(rng, 2, shape, 5000, cin0)
    torch.manual_seed(2)
    bn = lambda c: torch.nn.BatchNorm1d(c, eps=1e-3, momentum=0.01)
    net = spconv.SparseSequential(
        spconv.SubMConv3d(cin0, 16, 3, bias=False, indice_key="a"), bn(16), torch.nn.ReLU(),
        spconv.SubMConv3d(16, 32, 3, bias=False, indice_key="a"), bn(32), torch.nn.ReLU(),
        spconv.SparseConv3d(32, 64, 3, 2, padding=1, bias=False), bn(64), torch.nn.ReLU(),
        spconv.SubMConv3d(64, 64, 3, bias=True, indice_key="b"), bn(64),
        spconv.SubMConv3d(64, 64, 3, bias=False, indice_key="b"), bn(64), torch.nn.ReLU()).cuda()
    for m in net.modules():
        if isinstance(m, torch.nn.BatchNorm1d):
            torch.nn.init.uniform_(m.weight, 0.5, 1.5)
            torch.nn.init.normal_(m.bias, 0, 0.2)
    net2 = copy.deepcopy(net)
    gout = None
    res = []
    for fused, model in ((True, net), (False, net2)):
        spconv.FUSE_BN_STATS = fused
        try:
            spconv.ARENA.reset(torch.device("cuda", 0))
            f = torch.from_numpy(feats).cuda().to(dtype).requires_grad_(True)
            y = model(spconv.SparseConvTensor(f, torch.from_numpy(idx).cuda(), shape, 2)).features
            if gout is None:
                gout = torch.randn(y.shape, generator=torch.Generator().manual_seed(3)).cuda().to(dtype)
            y.backward(gout)
            res.append((y.detach().float(), f.grad.float(), [p.grad.float() for p in model.parameters()],
                        [b.clone() for b in model.buffers()]))
        finally:
            spconv.FUSE_BN_STATS = True
    (ya, ga, pa, ba), (yb, gb, pb, bb) = res
    tol = 1e-4 if dtype == torch.float32 else 2e-2
    rel = lambda a, b: float((a - b).abs().max()) / (float(b.abs().max()) + 1e-12)
    # gradients in relative L2: the two routes sum the statistics in different orders, the activations differ in
    # their last bits, and a pre-activation that close to zero takes the other side of a ReLU in one of them -- the
    # backward mask is discontinuous there (single sites differ by O(1)), the tensors agree
    l2 = lambda a, b: float((a - b).norm()) / (float(b.norm()) + 1e-30)
    assert rel(ya, yb) < tol
    gtol = 2e-3 if dtype == torch.float32 else 5 * tol
    assert l2(ga, gb) < gtol
    names = [n for n, _ in net.named_parameters()]
    for name, a, b in zip(names, pa, pb):
        if name == "9.bias":  # conv bias in front of a train-mode BN: zero true gradient, rounding noise only
            continue
        assert l2(a, b) < gtol, name
    for a, b in zip(ba, bb):
        if a.is_floating_point():
            assert torch.allclose(a, b, rtol=1e-3, atol=1e-5)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_split_bn_entry_points_and_channel_affine_sweeps(dtype):
    """d3d_bn_stats + d3d_bn_apply == d3d_bn_relu_fwd; d3d_channel_affine(_bwd) against their formulas."""
    from det3d_amd import _lib

    L = _lib.lib()
    dt = _lib.D3D_BF16 if dtype == torch.bfloat16 else _lib.D3D_F32
    n, c = 12345, 64
    torch.manual_seed(4)
    x = (torch.randn(n, c, device="cuda") * 2 + 0.5).to(dtype)
    dy = torch.randn(n, c, device="cuda").to(dtype)
    gamma, beta = torch.rand(c, device="cuda") + 0.5, torch.randn(c, device="cuda") * 0.1
    st = _lib.stream_ptr(x.device)
    rm, rv = torch.zeros(c, device="cuda"), torch.ones(c, device="cuda")
    rm2, rv2 = rm.clone(), rv.clone()
    y1, y2 = torch.empty_like(x), torch.empty_like(x)
    save1, save2 = torch.empty(2, c, device="cuda"), torch.empty(2, c, device="cuda")
    ws1 = torch.empty(L.d3d_bn_reduce_ws_floats(n, c, dt), device="cuda")
    ws2 = torch.empty_like(ws1)
    _lib.check(L.d3d_bn_relu_fwd(x.data_ptr(), n, c, dt, gamma.data_ptr(), beta.data_ptr(), rm.data_ptr(), rv.data_ptr(),
                                 0.01, 1e-3, 1, 1, y1.data_ptr(), save1[0].data_ptr(), save1[1].data_ptr(),
                                 ws1.data_ptr(), st), "fwd")
    _lib.check(L.d3d_bn_stats(x.data_ptr(), n, c, dt, ws2.data_ptr(), st), "stats")
    _lib.check(L.d3d_bn_apply(x.data_ptr(), n, c, dt, gamma.data_ptr(), beta.data_ptr(), rm2.data_ptr(), rv2.data_ptr(),
                              0.01, 1e-3, 1, 1, ws2.data_ptr(), 1, y2.data_ptr(), save2[0].data_ptr(),
                              save2[1].data_ptr(), st), "apply")
    # deterministic reductions: the two routes are the same kernels on the same data -> bit-equal
    assert torch.equal(ws1[:2 * c], ws2[:2 * c])
    assert torch.equal(y1, y2) and torch.equal(save1, save2) and torch.equal(rv, rv2) and torch.equal(rm, rm2)
    # channel affine forward / backward
    scale, shift = gamma * save1[1], beta - save1[0] * gamma * save1[1]
    y3 = torch.empty_like(x)
    _lib.check(L.d3d_channel_affine(x.data_ptr(), n, c, dt, scale.data_ptr(), shift.data_ptr(), 1, y3.data_ptr(), st), "aff")
    ref = torch.relu(x.float() * scale + shift)
    assert torch.allclose(y3.float(), ref, atol=2e-2 if dtype == torch.bfloat16 else 1e-5, rtol=1e-2 if dtype == torch.bfloat16 else 1e-5)
    ca, cb, cc = torch.rand(c, device="cuda"), torch.randn(c, device="cuda") * 0.01, torch.randn(c, device="cuda") * 0.01
    dx = torch.empty_like(x)
    _lib.check(L.d3d_channel_affine_bwd(x.data_ptr(), dy.data_ptr(), n, c, dt, scale.data_ptr(), shift.data_ptr(), 1,
                                        ca.data_ptr(), cb.data_ptr(), cc.data_ptr(), dx.data_ptr(), st), "affb")
    dz = dy.float() * ((x.float() * scale + shift) > 0)
    ref = ca * dz + cb * x.float() + cc
    assert torch.allclose(dx.float(), ref, atol=2e-2 if dtype == torch.bfloat16 else 1e-5, rtol=1e-2 if dtype == torch.bfloat16 else 1e-5)


# ---------------------------------------------------------------- strided data gradient in parity-class row order
def _plan_reference(idx, ksize, stride, padding):
    """numpy restatement of det3d_amd/csrc/d3d_rowplan.h: classes, stable order, per-class offset lists"""
    r = (idx[:, 1:4] + np.array(padding)) % np.array(stride)
    cls = (r[:, 0] * stride[1] + r[:, 1]) * stride[2] + r[:, 2]
    ncls = stride[0] * stride[1] * stride[2]
    perm = np.argsort(cls, kind="stable")
    start = np.concatenate([[0], np.cumsum(np.bincount(cls, minlength=ncls))])
    lists = []
    for c in range(ncls + 1):
        rc = [c // (stride[1] * stride[2]), (c // stride[2]) % stride[1], c % stride[2]]
        ks = [k for k in range(ksize[0] * ksize[1] * ksize[2])
              if c == ncls or all(kk % s == q for kk, s, q in zip(
                  [k // (ksize[1] * ksize[2]), (k // ksize[2]) % ksize[1], k % ksize[2]], stride, rc))]
        lists.append(ks or [0])
    return ncls, perm, start, lists


@pytest.mark.parametrize("gi", [1, 2, 3])
def test_dgrad_plan_is_the_stable_class_sort(gi):
    ksize, stride, padding, _ = GEOMS[gi]
    rng = np.random.default_rng(40 + gi)
    shape = [11, 60, 52]
    idx, _ = _random_sparse(rng, 3, shape, 2500, 1)   # > 1024 rows per class: several sort blocks
    rb = _build(idx, 3, shape, ksize, stride, padding, False)
    assert rb.dgrad_plan is not None
    plan = rb.dgrad_plan.cpu().numpy()
    ncls, perm, start, lists = _plan_reference(idx, ksize, stride, padding)
    n = len(idx)
    assert plan[0] == ncls
    assert np.array_equal(plan[1:2 + ncls], start)
    assert np.array_equal(plan[304:304 + n], perm)
    for c, ks in enumerate(lists):
        L = plan[16 + 32 * c: 16 + 32 * (c + 1)]
        assert L[0] == len(ks) and list(L[1:1 + len(ks)]) == ks
        packed = [int(L[28:30].view(np.uint64)[0]), int(L[30:32].view(np.uint64)[0])]
        for j, k in enumerate(ks[:24]):
            assert (packed[j // 12] >> (5 * (j % 12))) & 31 == k
    # every pair of the input-stationary table sits at an offset of its row's class list
    t = rb.nbr_in.view(rb.kvol, rb.cap_in)[:, :n].cpu().numpy()
    r = (idx[:, 1:4] + np.array(padding)) % np.array(stride)
    cls = (r[:, 0] * stride[1] + r[:, 1]) * stride[2] + r[:, 2]
    k, i = np.nonzero(t >= 0)
    assert all(kk in lists[cls[ii]] for kk, ii in zip(k[:5000], i[:5000]))


@pytest.mark.parametrize("cin,cout", [(16, 32), (32, 64), (64, 64), (64, 128), (128, 128)])
@pytest.mark.parametrize("gi", [1, 2, 3])
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_planned_strided_dgrad_is_bit_identical(cin, cout, gi, dtype):
    """the planned launch adds the same products in the same order per row (skipped offsets contribute exact
    zeros): torch.equal with the 27-offset launch, with and without the BN-backward epilogue's row sums"""
    from det3d_amd import _lib, spconv

    ksize, stride, padding, _ = GEOMS[gi]
    rng = np.random.default_rng(cin + cout + gi)
    shape = [11, 70, 64]
    idx, feats = _random_sparse(rng, 2, shape, 4000, cin)
    rb = _build(idx, 2, shape, ksize, stride, padding, False)
    n = len(idx)
    kvol = rb.kvol
    w = torch.from_numpy((rng.standard_normal((kvol, cin, cout)) / np.sqrt(cin * 9)).astype(np.float32)).cuda()
    gout = torch.from_numpy(rng.standard_normal((rb.n_out, cout)).astype(np.float32)).cuda().to(dtype)
    _, wd = spconv._packed_weights(w, kvol, cin, cout, dtype, False, True, cin)
    stream = _lib.stream_ptr(gout.device)
    x = torch.from_numpy(feats).cuda().to(dtype)
    mean, invstd = x.float().mean(0), 1.0 / (x.float().var(0, unbiased=False) + 1e-3).sqrt()
    gamma, beta = torch.rand(cin, device="cuda") + 0.5, torch.randn(cin, device="cuda") * 0.1

    def run(planned, with_epilogue):
        gin = torch.full((n, cin), float("nan"), dtype=dtype, device="cuda")
        ep = part = None
        if with_epilogue:
            part = torch.full((int(_lib.lib().d3d_conv_epilogue_ws_floats(n, cin)),), float("nan"), device="cuda")
            ep = _lib.ConvEpilogue(bn_x=x.data_ptr(), bn_mean=mean.data_ptr(), bn_invstd=invstd.data_ptr(),
                                   bn_gamma=gamma.data_ptr(), bn_beta=beta.data_ptr(), bn_relu=1,
                                   bn_partial=part.data_ptr(), ws_floats=part.numel())
        saved, rb.dgrad_plan = rb.dgrad_plan, (rb.dgrad_plan if planned else None)
        try:
            spconv.launch_dgrad(rb, gout, wd, cout, cin, gin, ep, stream)
        finally:
            rb.dgrad_plan = saved
        torch.cuda.synchronize()
        return gin, (part[:2 * cin].clone() if part is not None else None)

    assert _lib.lib().d3d_sparse_conv_dgrad_plan_supported(cout, cin, spconv._DT[dtype])
    a, _ = run(False, False)
    b, _ = run(True, False)
    assert not torch.isnan(a.float()).any()
    assert torch.equal(a, b)
    a2, pa = run(False, True)
    b2, pb = run(True, True)
    assert torch.equal(a2, a) and torch.equal(b2, a)
    # the epilogue's sums run over the rows in plan order: same terms, other order
    assert torch.allclose(pa, pb, rtol=2e-4, atol=2e-3 * float(pa.abs().max()))
    b3, pb3 = run(True, True)
    assert torch.equal(pb3, pb)   # ... and reproducibly so


RESIDENT_CASES = [(16, 16), (16, 32), (32, 16), (32, 32), (32, 64), (64, 32), (64, 64)]


@pytest.mark.parametrize("cin,cout", RESIDENT_CASES)
@pytest.mark.parametrize("gi", [0, 1, 3])
@pytest.mark.parametrize("cell_order", [False, True])
def test_resident_weights_kernel_is_the_per_offset_kernel_bit_for_bit(cin, cout, gi, cell_order):
    """spconv_fwd_resident_kernel (d3d_set_tuning knob 11, tiling sets 1-3 and the default policy -1: KC weight matrices resident in LDS, waves
    free-running between the chunk fences) against spconv_fwd_mfma_kernel (knob 11 = 0) through the C ABI, f32 features:
    the produced rows bit for bit -- plain, with bias + addend, with the BatchNorm-statistics epilogue, with the
    BatchNorm-backward epilogue -- on a submanifold table, a strided one and the (3, 1, 1) kernel (kvol 3 < a chunk);
    the statistics within f32 reassociation (the block shape, hence the fold tree, may differ).
    `cell_order`: input rows in cell order (as behind a strided layer) or shuffled."""
    import ctypes

    from det3d_amd import _lib

    ksize, stride, padding, subm = GEOMS[gi]
    rng = np.random.default_rng(cin * 7 + cout + gi)
    shape = [9, 40, 36]
    idx, _ = _random_sparse(rng, 3, shape, 1500, 1)
    if cell_order:
        idx = idx[np.lexsort((idx[:, 3], idx[:, 2], idx[:, 1], idx[:, 0]))]
    rb = _build(idx, 3, shape, ksize, stride, padding, subm)
    n, kvol = rb.n_out, rb.kvol
    n_in = len(idx)
    L = _lib.lib()
    dev = torch.device("cuda", 0)
    st = _lib.stream_ptr(dev)
    torch.manual_seed(cin + cout)
    f = torch.randn(n_in, cin, device=dev)
    w32 = torch.randn(kvol, cin, cout, device=dev) / np.sqrt(cin * 9.0)
    wpack = torch.empty((kvol, cin, cout), device=dev)
    _lib.check(L.d3d_pack_conv_weights(w32.data_ptr(), kvol, cin, cout, _lib.D3D_F32, 1, wpack.data_ptr(), None, st), "pack")
    bias = torch.randn(cout, device=dev)
    addend = torch.randn(n, cout, device=dev)
    bn_x = torch.randn(n, cout, device=dev)
    mean, invstd = torch.randn(cout, device=dev) * 0.1, torch.rand(cout, device=dev) + 0.5
    gamma, beta = torch.rand(cout, device=dev) + 0.5, torch.randn(cout, device=dev) * 0.1
    nws = L.d3d_conv_epilogue_ws_floats(n, cout)

    def run(v, mode):
        o = torch.full((n, cout), float("nan"), device=dev)
        ws = torch.zeros(nws, device=dev)
        ep, b = None, None
        if mode == "bias_addend":
            ep, b = _lib.ConvEpilogue(addend=addend.data_ptr()), bias.data_ptr()
        elif mode == "stats":
            ep = _lib.ConvEpilogue(stats=ws.data_ptr(), ws_floats=nws)
        elif mode == "bn_backward":
            ep = _lib.ConvEpilogue(bn_partial=ws.data_ptr(), ws_floats=nws, bn_x=bn_x.data_ptr(), bn_mean=mean.data_ptr(),
                                   bn_invstd=invstd.data_ptr(), bn_gamma=gamma.data_ptr(), bn_beta=beta.data_ptr(), bn_relu=1)
        _lib.check(L.d3d_set_tuning(11, v), "knob")
        try:
            _lib.check(L.d3d_sparse_conv_fwd_ex(f.data_ptr(), wpack.data_ptr(), b, rb.nbr_out.data_ptr(), rb.cap_out, n, kvol,
                                                cin, cout, _lib.D3D_F32, o.data_ptr(), ctypes.byref(ep) if ep else None, st),
                       "fwd")
        finally:
            L.d3d_set_tuning(11, -1)
        torch.cuda.synchronize()
        return o, ws[:2 * cout].clone()

    for mode in ("plain", "bias_addend", "stats", "bn_backward"):
        ref, ref_red = run(0, mode)
        assert torch.isfinite(ref).all()
        for v in (-1, 1, 2, 3):
            o, red = run(v, mode)
            assert torch.equal(o, ref), (mode, v, (o - ref).abs().max().item())
            if mode in ("stats", "bn_backward"):
                scale = ref_red.abs().max().item() + 1.0
                assert (red - ref_red).abs().max().item() <= 2e-5 * scale, (mode, v)
