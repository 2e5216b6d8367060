"""Fused HIP PillarFeatureNet (d3d_pfn_fwd / d3d_pfn_bwd) vs the PyTorch module path that is
itself pinned to the reference's PillarFeatureNet by tests/golden/vfe_pfn.pt."""
import copy
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN

pytestmark = pytest.mark.gpu

PP = dict(voxel_size=[0.16, 0.16, 4], pc_range=[0, -39.68, -3, 69.12, 39.68, 1])


def test_fused_pfn_matches_reference_golden():
    from det3d_amd import readers

    g = torch.load(os.path.join(GOLDEN, "vfe_pfn.pt"), weights_only=False)["pfn"]
    pfn = readers.PillarFeatureNet(num_input_features=4, num_filters=(16,), with_distance=False, **PP)
    pfn.load_state_dict(g["state"])
    pfn = pfn.cuda()
    f, n, c = g["features"].cuda(), g["num_voxels"].cuda(), g["coors"].cuda()
    assert pfn._fusable(f)
    pfn.eval()
    with torch.no_grad():
        assert torch.allclose(pfn(f, n, c).cpu(), g["out_eval"], atol=1e-5, rtol=1e-5)
    pfn.train()
    assert torch.allclose(pfn(f, n, c).detach().cpu(), g["out_train"], atol=2e-5, rtol=1e-4)


@pytest.mark.parametrize("c,nf,t", [(4, 64, 100), (5, 64, 20), (4, 32, 7)])
def test_fused_pfn_fwd_bwd_vs_torch_path(c, nf, t):
    from det3d_amd import readers

    torch.manual_seed(c * 100 + t)
    rng = np.random.default_rng(t)
    m = 3000
    num = torch.from_numpy(np.minimum(rng.geometric(0.35, m), t).astype(np.int32))
    num[:5] = t  # full pillars: no padded row competes in the max
    feats = torch.rand(m, t, c) * torch.tensor([69.0, 79.0, 4.0, 1.0, 0.5][:c]) + torch.tensor([0, -39.6, -3.0, 0, 0][:c])
    feats = feats * (torch.arange(t)[None, :] < num[:, None]).float()[:, :, None]
    coors = torch.stack([torch.randint(0, 2, (m,)), torch.zeros(m, dtype=torch.long),
                         torch.randint(0, 496, (m,)), torch.randint(0, 432, (m,))], 1).int()
    a = readers.PillarFeatureNet(num_input_features=c, num_filters=(nf,), with_distance=False, **PP).cuda()
    with torch.no_grad():
        a.pfn_layers[0].norm.weight.uniform_(0.5, 1.5)
        a.pfn_layers[0].norm.bias.uniform_(-0.3, 0.3)
    b = copy.deepcopy(a)
    f, n, co = feats.cuda(), num.cuda(), coors.cuda()
    g = torch.randn(m, nf).cuda()
    ya = a(f, n, co)
    ya.backward(g)
    readers.FUSE_PFN = False
    try:
        yb = b(f, n, co)
        yb.backward(g)
    finally:
        readers.FUSE_PFN = True
    assert torch.allclose(ya, yb, atol=1e-4, rtol=1e-4)
    la, lb = a.pfn_layers[0], b.pfn_layers[0]
    assert torch.allclose(la.norm.running_mean, lb.norm.running_mean, atol=1e-6, rtol=1e-4)
    assert torch.allclose(la.norm.running_var, lb.norm.running_var, atol=1e-6, rtol=1e-4)
    for pa, pb, name in ((la.linear.weight, lb.linear.weight, "linear"), (la.norm.weight, lb.norm.weight, "gamma"),
                         (la.norm.bias, lb.norm.bias, "beta")):
        scale = float(pb.grad.abs().max())
        assert torch.allclose(pa.grad, pb.grad, atol=2e-3 * scale, rtol=2e-3), name


def test_fused_pfn_sums_are_reproducible():
    """round 3: statistics, dgamma / dbeta and dW are summed slab by slab in a fixed order (no float atomics): two
    runs with other work and garbage scratch in between are bit-equal, also at more pillars than waves (> 2048)"""
    from det3d_amd import readers

    torch.manual_seed(11)
    m, t, c, nf = 9000, 20, 5, 64
    num = torch.randint(1, t + 1, (m,), dtype=torch.int32)
    feats = torch.rand(m, t, c) * 50.0
    feats = feats * (torch.arange(t)[None, :] < num[:, None]).float()[:, :, None]
    coors = torch.stack([torch.zeros(m, dtype=torch.long), torch.zeros(m, dtype=torch.long),
                         torch.randint(0, 496, (m,)), torch.randint(0, 432, (m,))], 1).int()
    f, n, co = feats.cuda(), num.cuda(), coors.cuda()
    g = torch.randn(m, nf).cuda()
    net = readers.PillarFeatureNet(num_input_features=c, num_filters=(nf,), with_distance=False, **PP).cuda()
    outs = []
    for rep in range(2):
        a = copy.deepcopy(net)
        torch.empty(1 << 24, device="cuda").fill_(float("nan"))   # whatever the allocator hands out next is garbage
        y = a(f, n, co)
        y.backward(g)
        outs.append((y.detach().clone(), a.pfn_layers[0].linear.weight.grad.clone(),
                     a.pfn_layers[0].norm.weight.grad.clone(), a.pfn_layers[0].norm.bias.grad.clone(),
                     a.pfn_layers[0].norm.running_var.clone()))
        if rep == 0:
            torch.randn(4096, 4096, device="cuda") @ torch.randn(4096, 4096, device="cuda")   # other CUs busy
    for x, y in zip(*outs):
        assert torch.isfinite(x).all() and torch.equal(x, y)
