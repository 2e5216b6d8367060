"""a steady-state training step allocates nothing from the device.

The data-dependent tensors of the sparse stack come from the step arena and the geometry ring (det3d_amd/arena.py,
prefetch.py); everything else has the same size every step and is served from the caching allocator's pools.  Checked
the way bench.py reports it: `num_device_alloc` and `reserved_bytes` of torch.cuda.memory_stats over 50 steps of the
threaded-prefetch loop, after a warm-up that has seen every batch once."""
import os

import pytest
import torch

from conftest import ROOT

pytestmark = pytest.mark.gpu


def _setup(step_arena=True):
    from det3d_amd import Config, build_detector, synthetic
    from det3d_amd.trainer import TrainStep

    cfg = Config.fromfile(os.path.join(ROOT, "det3d_amd", "configs", "kitti_car_second.py"))
    torch.manual_seed(0)
    model = build_detector(cfg.model, cfg.train_cfg, cfg.test_cfg).cuda().train()
    model.backbone.dense_channels_last = True
    ds = synthetic.KittiCarSynthetic(n_points=20000, n_seeds=16)
    dev = torch.device("cuda", 0)
    batches = [ds.device_batch(i, 2, dev) for i in range(8)]
    step = TrainStep(model, total_steps=200, step_arena=step_arena)
    make = lambda i: synthetic.make_example(batches[i % len(batches)], synthetic.KITTI_CAR, fused_mean=True, device_targets=True)
    return model, step, make, dev


def _loop(pf, step, n, start=0, state=None):
    state = state or {"ex": pf.prepare(start), "next": pf.submit(start + 1)}
    losses = []
    for i in range(start, start + n):
        losses.append(step(state["ex"]))
        state["ex"] = state["next"].result()
        state["next"] = pf.submit(i + 2)
    return state, losses


def test_steady_state_step_allocates_nothing_from_the_device():
    from det3d_amd import arena
    from det3d_amd.prefetch import GeometryPrefetcher

    model, step, make, dev = _setup()
    pf = GeometryPrefetcher(make, model, dev)
    state, _ = _loop(pf, step, 64)  # every batch eight times: pools, arenas and the ring have seen their largest sizes
    torch.cuda.synchronize()
    m0 = torch.cuda.memory_stats(dev)
    state, losses = _loop(pf, step, 50, start=64, state=state)
    torch.cuda.synchronize()
    m1 = torch.cuda.memory_stats(dev)
    state["next"].result()
    pf.close()
    d = {k: m1[k] - m0[k] for k in ("num_device_alloc", "num_device_free", "reserved_bytes.all.current",
                                    "reserved_bytes.large_pool.current", "reserved_bytes.small_pool.current")}
    sa = arena.step_arena(dev).stats()
    print("50 steady-state steps: %s; step arena %s; geometry ring %s" % (d, sa, pf.arena_stats()))
    # The claim: nothing data-dependent reaches hipMalloc -- the LARGE pool (every tensor >= 1 MB) does not move at all.
    # The small pool (2 MB segments of sub-megabyte tensors PyTorch itself allocates: scalars, count tensors, H2D staging)
    # may still take a few segments on a box where the host runs further ahead than during the warm-up -- their release
    # waits for another stream's event -- : 0 on three boxes, 1 and 5 segments on two others.  Bounded, not asserted to zero.
    assert d["reserved_bytes.large_pool.current"] == 0, d
    assert d["reserved_bytes.small_pool.current"] <= (32 << 20) and d["num_device_alloc"] <= 16, d
    assert sa["spills"] == 0 or sa["grows"] >= 1
    assert all(torch.isfinite(l) for l in losses)


def test_arena_step_is_bit_identical_to_the_allocator_step():
    """same kernels, same arguments, other addresses: losses of 6 steps from identical starts are bit-equal"""
    from det3d_amd.prefetch import GeometryPrefetcher

    out = []
    for use in (True, False):
        model, step, make, dev = _setup(step_arena=use)
        pf = GeometryPrefetcher(make, model, dev, ring=3 if use else 0)
        _, losses = _loop(pf, step, 6)
        torch.cuda.synchronize()
        pf.close()
        out.append(torch.stack(losses).cpu())
    assert torch.equal(out[0], out[1]), (out[0], out[1])


def test_backward_outside_the_step_arena_generation_is_refused():
    from det3d_amd import arena, spconv

    dev = torch.device("cuda", 0)
    torch.manual_seed(0)
    n = 5000
    idx = torch.unique(torch.cat([torch.zeros(n, 1, dtype=torch.int64), torch.randint(0, 40, (n, 3))], 1), dim=0).int().to(dev)
    feats = torch.randn(idx.shape[0], 16, device=dev, requires_grad=True)
    conv = spconv.SubMConv3d(16, 16, 3, bias=False, indice_key="k").to(dev)
    a = arena.BumpArena(dev, "step", initial_bytes=64 << 20).begin()
    with arena.bound(a):
        y = conv(spconv.SparseConvTensor(feats, idx, [40, 40, 40], 1)).features
    y.sum().backward()  # same generation: fine, from the autograd thread, out of the captured arena
    assert torch.isfinite(feats.grad).all()
    with arena.bound(a):
        y = conv(spconv.SparseConvTensor(feats, idx, [40, 40, 40], 1)).features
    a.begin()
    with pytest.raises(arena.ArenaError, match="rewound"):
        y.sum().backward()


def test_backward_runs_under_the_forwards_library_context():
    """knob 10 of a bound context changes the weight pack format; the backward of the fused nodes
    runs on the autograd thread, which re-binds the forward's context -- gradients equal those of the same knob set on
    the default context."""
    import ctypes

    from det3d_amd import _lib, spconv

    L = _lib.lib()
    dev = torch.device("cuda", 0)
    torch.manual_seed(1)
    n = 6000
    idx = torch.unique(torch.cat([torch.zeros(n, 1, dtype=torch.int64), torch.randint(0, 30, (n, 3))], 1), dim=0).int().to(dev)
    f0 = torch.randn(idx.shape[0], 32, device=dev)
    seq = spconv.SparseSequential(spconv.SubMConv3d(32, 32, 3, bias=False, indice_key="a"), torch.nn.BatchNorm1d(32), torch.nn.ReLU(),
                                  spconv.SubMConv3d(32, 32, 3, bias=False, indice_key="a"), torch.nn.BatchNorm1d(32), torch.nn.ReLU()).to(dev).train()

    def run():
        spconv._PACK_CACHE.clear()
        for p in seq.parameters():
            p.grad = None
        f = f0.clone().requires_grad_(True)
        x = spconv.SparseConvTensor(f, idx, [30, 30, 30], 1)
        seq.plan(x)
        spconv.ARENA.reset(dev)
        seq(x).features.square().sum().backward()
        torch.cuda.synchronize()
        return [f.grad.clone()] + [p.grad.clone() for p in seq.parameters()]

    # (a) exact-f32 mode on the default context: forward and backward both see knob 10
    assert L.d3d_set_tuning(10, 1) == 0
    try:
        want = run()
    finally:
        L.d3d_set_tuning(10, 0)
    # (b) the same knob on a context bound to THIS thread only
    with _lib.Context() as c:
        assert L.d3d_set_tuning(10, 1) == 0
        v = ctypes.c_int(-1)
        assert L.d3d_ctx_get_tuning(10, ctypes.byref(v)) == 0 and v.value == 1
        assert _lib.current_ctx() == c.handle.value
        got = run()
        torch.cuda.synchronize()
    assert _lib.current_ctx() is None
    for a, b in zip(want, got):
        assert torch.equal(a, b)


def test_an_example_that_outlives_its_ring_slot_is_refused():
    """the geometry ring's contract (prefetch.py): an example is dead by the time `ring - 1` further builds were requested.
    The backbone checks the slot's generation and raises instead of convolving over recycled neighbour tables."""
    from det3d_amd import _lib
    from det3d_amd.prefetch import GeometryPrefetcher

    model, step, make, dev = _setup()
    pf = GeometryPrefetcher(make, model, dev, ring=3)
    old = pf.prepare(0)
    keep = [pf.prepare(i) for i in (1, 2)]
    step(old)                       # two further builds: still its slot's tenant
    keep.append(pf.prepare(3))      # the third rewinds slot 0
    torch.cuda.synchronize()
    with pytest.raises(_lib.D3dError, match="rewound"):
        step(old)
    pf.close()
