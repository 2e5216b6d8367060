"""d3d_adam_step (csrc/d3d_optim.hip) through det3d_amd.fused_adam.FusedAdam: clip + true weight decay + Adam in two
launches against (1) the update written out in float64 numpy and (2) the PyTorch sequence it replaces --
clip_grad_norm_, `p.mul_(1 - wd * lr)`, torch.optim.Adam with betas[0] rewritten each step
(det3d/solver/fastai_optim.py:155-171, hooks/optimizer.py:10-13).  A floating-point kernel: tolerance 2e-6 relative
to the parameter scale per step (f32 rounding of a different but equivalent operation order)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

SHAPES = [(1,), (7,), (64,), (3, 3, 3, 16, 16), (4097,), (128, 128, 3, 3), (100003,), (2, 5)]


def _setup(seed, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    params = [torch.randn(s, generator=g).cuda().requires_grad_(True) for s in SHAPES]
    grads = [[(torch.randn(s, generator=g) * scale).cuda() for s in SHAPES] for _ in range(4)]
    return params, grads


def _reference(params, grads, lrs, moms, wd, max_norm, grad_scale, beta2=0.99, eps=1e-8):
    p = [x.detach().double().cpu().numpy().copy() for x in params]
    m = [np.zeros_like(x) for x in p]
    v = [np.zeros_like(x) for x in p]
    norms = []
    for t, (gs, lr, b1) in enumerate(zip(grads, lrs, moms), 1):
        g64 = [g.double().cpu().numpy() * grad_scale for g in gs]
        total = np.sqrt(sum((g * g).sum() for g in g64))
        norms.append(total)
        coef = min(1.0, max_norm / (total + 1e-6)) if max_norm else 1.0
        for i in range(len(p)):
            g = g64[i] * coef
            m[i] = m[i] + (g - m[i]) * (1 - b1)
            v[i] = beta2 * v[i] + (1 - beta2) * g * g
            p[i] = p[i] * (1 - wd * lr) - lr / (1 - b1 ** t) * m[i] / (np.sqrt(v[i]) / np.sqrt(1 - beta2 ** t) + eps)
    return p, m, v, norms


@pytest.mark.parametrize("max_norm,scale", [(35.0, 0.01), (35.0, 1.0), (None, 1.0)])
@pytest.mark.parametrize("grad_scale", [1.0, 0.5])
def test_fused_step_matches_the_update_in_float64(max_norm, scale, grad_scale):
    from det3d_amd.fused_adam import FusedAdam

    params, grads = _setup(1, scale)
    lrs, moms, wd = [3e-4, 1e-3, 3e-3, 2e-3], [0.95, 0.92, 0.88, 0.85], 0.01
    ref_p, ref_m, ref_v, ref_norm = _reference(params, grads, lrs, moms, wd, max_norm, grad_scale)
    opt = FusedAdam(params)
    for gs, lr, b1, rn in zip(grads, lrs, moms, ref_norm):
        for p, g in zip(params, gs):
            p.grad = g.clone()
        norm = opt.step(lr, b1, decay=1.0 - wd * lr, max_norm=max_norm, grad_scale=grad_scale)
        assert abs(float(norm) - rn) <= 2e-6 * rn
    clipped = max_norm is not None and ref_norm[0] > max_norm
    assert clipped == (scale >= 1.0 and max_norm is not None)   # the case list covers both sides of the threshold
    for i, p in enumerate(params):
        tol = 4e-6 * max(1.0, float(np.abs(ref_p[i]).max()))
        assert np.abs(p.detach().cpu().numpy() - ref_p[i]).max() <= tol, (i, SHAPES[i])
        assert np.allclose(opt._m[i].cpu().numpy().reshape(ref_m[i].shape), ref_m[i], rtol=1e-5, atol=1e-7 * scale)
        assert np.allclose(opt._v[i].cpu().numpy().reshape(ref_v[i].shape), ref_v[i], rtol=1e-5, atol=1e-9 * scale * scale)


def test_fused_step_against_the_torch_sequence_and_its_checkpoint_layout():
    from det3d_amd.fused_adam import FusedAdam

    params, grads = _setup(2, 3.0)
    twins = [p.detach().clone().requires_grad_(True) for p in params]
    lrs, moms, wd, max_norm = [3e-4, 1e-3, 3e-3, 2e-3], [0.95, 0.92, 0.88, 0.85], 0.01, 35.0
    opt = FusedAdam(params)
    ref = torch.optim.Adam(twins, lr=lrs[0], betas=(moms[0], 0.99), weight_decay=0.0, foreach=True)
    for step, (gs, lr, b1) in enumerate(zip(grads, lrs, moms)):
        for p, q, g in zip(params, twins, gs):
            p.grad, q.grad = g.clone(), g.clone()
        opt.step(lr, b1, decay=1.0 - wd * lr, max_norm=max_norm)
        for grp in ref.param_groups:
            grp["lr"], grp["betas"] = lr, (b1, 0.99)
        torch.nn.utils.clip_grad_norm_(twins, max_norm)
        with torch.no_grad():
            torch._foreach_mul_(twins, 1.0 - wd * lr)
        ref.step()
        if step == 1:   # resume both from the OTHER one's checkpoint half way
            sd_f, sd_t = opt.state_dict(), ref.state_dict()
            opt2 = FusedAdam(params)
            opt2.load_state_dict(sd_t)
            ref.load_state_dict(dict(state=sd_f["state"], param_groups=sd_t["param_groups"]))
            opt = opt2
    for p, q in zip(params, twins):
        assert torch.allclose(p, q, rtol=0, atol=4e-6 * max(1.0, float(q.detach().abs().max())))


def test_fused_step_is_reproducible_and_loud():
    from det3d_amd import _lib
    from det3d_amd.fused_adam import FusedAdam

    outs = []
    for _ in range(2):
        params, grads = _setup(3, 10.0)
        opt = FusedAdam(params)
        torch.empty(1 << 22, device="cuda").normal_()   # other allocations / timing between the runs
        for gs in grads:
            for p, g in zip(params, gs):
                p.grad = g.clone()
            opt.step(1e-3, 0.9, decay=0.999, max_norm=35.0)
        outs.append([p.detach().clone() for p in params])
    assert all(torch.equal(a, b) for a, b in zip(*outs))
    params, _ = _setup(4)
    opt = FusedAdam(params)
    with pytest.raises(_lib.D3dError):
        opt.step(1e-3, 0.9)          # no gradients
    with pytest.raises(_lib.D3dError):
        FusedAdam([torch.zeros(4, dtype=torch.float64, device="cuda", requires_grad=True)])
    # a channels-last gradient (library convolutions produce them) is taken through a contiguous copy
    w = torch.randn(8, 4, 3, 3, device="cuda", requires_grad=True)
    o2 = FusedAdam([w])
    g = torch.randn(8, 4, 3, 3, device="cuda").contiguous(memory_format=torch.channels_last)
    w0 = w.detach().clone()
    w.grad = g
    o2.step(1e-2, 0.9)
    expect = w0 - 1e-2 * torch.sign(g)           # first Adam step: lr * g / |g|
    assert torch.allclose(w.detach(), expect, atol=1e-5)


def test_training_steps_with_the_fused_optimizer_follow_the_torch_sequence(monkeypatch):
    """the whole KITTI SECOND step, three updates, fused optimizer against PyTorch's launch sequence on the same
    initial weights: same losses and gradient norms from step to step.  (Catches an update the layers do not see:
    the packed-weight caches are keyed on the parameters' version counters, which a raw kernel does not move.)"""
    import os

    from det3d_amd import Config, build_detector, synthetic
    from det3d_amd.trainer import TrainStep

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cfg = Config.fromfile(os.path.join(root, "det3d_amd", "configs", "kitti_car_second.py"))
    dev = torch.device("cuda", 0)
    ds = synthetic.KittiCarSynthetic(n_points=6000, n_seeds=2)
    ex = synthetic.make_example(ds.device_batch(0, 1, dev), synthetic.KITTI_CAR)
    runs = {}
    for fused in ("1", "0"):
        monkeypatch.setenv("D3D_FUSED_ADAM", fused)
        torch.manual_seed(0)
        model = build_detector(cfg.model, cfg.train_cfg, cfg.test_cfg).to(dev).train()
        step = TrainStep(model, total_steps=10, autocast_dtype=None)
        assert (step.fused is not None) == (fused == "1")
        p0 = [p.detach().clone() for p in model.parameters()]
        losses, norms, first = [], [], None
        for i in range(3):
            losses.append(float(step(ex)))
            norms.append(float(step.last["grad_norm"]))
            if i == 0:
                first = [p.detach() - q for p, q in zip(model.parameters(), p0)]
        runs[fused] = (losses, norms, first)
    (la, na, pa), (lb, nb, pb) = runs["1"], runs["0"]
    assert la[0] == lb[0] and lb[2] < 0.9 * lb[0]           # same start, and the steps do move the loss
    assert np.allclose(la, lb, rtol=2e-3), (la, lb)
    # gradient norms: equal at the first step (same weights, same kernels: only the norm's own summation differs), then the
    # two optimizers drift apart the way ANY reordering of a sum does -- an element whose gradient is rounding noise takes
    # Adam's +-lr with the sign of that noise, and the next steps amplify it (tools/optim_follow.py: the fold order of the
    # weight-gradient slabs alone moves the second step's norm by 1 %, in both optimizers alike).  An update the layers do
    # not see leaves the loss where it was and the norms a factor apart, not a few per cent.
    assert np.allclose(na[:1], nb[:1], rtol=1e-6) and np.allclose(na[1:2], nb[1:2], rtol=2e-3), (na, nb)
    assert np.allclose(na, nb, rtol=3e-2), (na, nb)
    # the first update, tensor by tensor (same gradients on both sides): f32 rounding of two operation orders.  Later
    # updates are not compared element-wise: an element whose gradient is ~0 takes Adam's +-0.75 lr on rounding noise,
    # in either implementation
    for x, y in zip(pa, pb):
        assert float((x - y).abs().max()) <= 1e-7 + 1e-5 * float(y.abs().max())
    assert max(float(y.abs().max()) for y in pb) > 1e-4


def test_pack_all_rewrites_every_cached_operand_like_the_per_layer_packs():
    """d3d_pack_all / repack.Repacker: after the parameters changed, ONE launch must leave every cached packed operand
    (sparse [kvol][cout][cin] / mirrored dgrad form in hi / lo planes or bf16, dense [tap][cout][cin] planes) bit-equal
    to what the per-layer pack launches produce from the same weights, and stamp the caches with the new versions."""
    from det3d_amd import dense_conv, spconv
    from det3d_amd.repack import Repacker

    dev = torch.device("cuda", 0)
    torch.manual_seed(0)
    sparse = [torch.nn.Parameter(torch.randn(3, 3, 3, ci, co, device=dev) * 0.1) for ci, co in ((16, 16), (32, 64), (64, 64), (128, 128))]
    sparse.append(torch.nn.Parameter(torch.randn(3, 1, 1, 64, 64, device=dev) * 0.1))
    dense = [torch.nn.Parameter(torch.randn(co, ci, k, k, device=dev) * 0.1) for co, ci, k in ((128, 128, 3), (256, 128, 3), (128, 256, 1), (64, 64, 3))]
    subm = [True, False, True, True, False]
    dts = [torch.float32, torch.float32, torch.bfloat16, torch.float32, torch.float32]

    def pack_now():
        out = []
        for p, sm, dt in zip(sparse, subm, dts):
            kvol = p.numel() // (p.shape[-2] * p.shape[-1])
            out.append(spconv._packed_weights(p, kvol, p.shape[-2], p.shape[-1], dt, sm, True, p.shape[-2]))
        for p in dense:
            out.append((dense_conv._packed(p, "fwd", "split"), dense_conv._packed(p, "dgrad", "split")))
        out.append((dense_conv._packed(dense[0], "fwd", "bf16"), dense_conv._packed(dense[0], "dgrad", "bf16")))
        return out

    first = pack_now()
    rp = Repacker(sparse + dense)
    with torch.no_grad():
        for p in sparse + dense:
            p.add_(torch.randn_like(p) * 0.05)  # bumps ._version, like an optimizer step
    n = rp.run()
    assert n == len(sparse) + len(dense) + 1
    flat = lambda o: [t for t in (o if isinstance(o, (list, tuple)) else [o]) for t in (flat(t) if isinstance(t, (list, tuple)) else [t])]
    got = [t.clone() if t is not None else None for t in flat(first)]          # the SAME buffers, rewritten in place
    again = pack_now()                                                          # cache hits: no new pack launches
    for a, b in zip(flat(first), flat(again)):
        assert (a is None and b is None) or a.data_ptr() == b.data_ptr()
    spconv._PACK_CACHE.clear()
    dense_conv._PACK.clear()
    ref = pack_now()                                                            # fresh per-layer packs of the new weights
    for a, b in zip(got, flat(ref)):
        bits = lambda t: t.view(torch.int16 if t.dtype == torch.bfloat16 else torch.int32)  # (f32-typed buffers hold planes)
        assert (a is None and b is None) or torch.equal(bits(a), bits(b))
