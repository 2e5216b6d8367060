"""DistributedSyncBN (det3d_amd/syncbn.py) against the reference's semantics (det3d/ops/syncbn/syncbn.py:15-118),
world_size 2 over gloo on CPU.  The expected values come from a single-process differentiable restatement of
the reference formulas evaluated on both ranks' data at once (autograd supplies the gradients), so the test
checks the packed-collective implementation, the unbiased-variance normalisation, equal-weight rank means with
different N per rank, and the summed affine gradients."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _data(rank, shape_kind):
    g = torch.Generator().manual_seed(100 + rank)
    if shape_kind == "sparse":  # [N, C] with a different N per rank
        return torch.randn((37 if rank == 0 else 52, 16), generator=g) * 2 + 0.5
    return torch.randn((2, 8, 5, 6), generator=g) * (1 + rank) - 0.3


def _head(y, rank):
    w = torch.linspace(0.5, 1.5, y.shape[1]).view([1, -1] + [1] * (y.dim() - 2))
    return ((y * w) ** 2).sum() * (1.0 + 0.5 * rank)


def _worker(rank, world, port, shape_kind, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from det3d_amd.syncbn import DistributedSyncBN

        torch.manual_seed(0)
        x = _data(rank, shape_kind).requires_grad_(True)
        bn = DistributedSyncBN(x.shape[1], eps=1e-3, momentum=0.01)
        with torch.no_grad():
            bn.weight.copy_(torch.linspace(0.5, 2.0, x.shape[1]))
            bn.bias.copy_(torch.linspace(-1.0, 1.0, x.shape[1]))
        y = bn(x)
        _head(y, rank).backward()
        q.put((rank, y.detach().numpy(), x.grad.numpy(), bn.weight.grad.numpy(), bn.bias.grad.numpy(),
               bn.running_mean.numpy().copy(), bn.running_var.numpy().copy()))
    except Exception:  # report instead of leaving the parent waiting on the queue
        import traceback

        q.put((rank, "error", traceback.format_exc()))
    finally:
        dist.destroy_process_group()


def _expected(shape_kind, world=2, eps=1e-3, momentum=0.01):
    xs = [_data(r, shape_kind).double().requires_grad_(True) for r in range(world)]
    c = xs[0].shape[1]
    gamma = torch.linspace(0.5, 2.0, c).double().requires_grad_(True)
    beta = torch.linspace(-1.0, 1.0, c).double().requires_grad_(True)
    dims = [0] + list(range(2, xs[0].dim()))
    ex = sum(x.mean(dims) for x in xs) / world          # equal-weight mean of the rank means
    exs = sum((x * x).mean(dims) for x in xs) / world
    outs = []
    for r, x in enumerate(xs):
        n = (x.numel() // c) * world                   # the reference uses the LOCAL count times world
        cf = n / (n - 1)
        var = (exs - ex * ex) * cf
        shp = [1, -1] + [1] * (x.dim() - 2)
        outs.append((gamma.view(shp) * (x - ex.view(shp)) / (var + eps).sqrt().view(shp) + beta.view(shp), var))
    loss = sum(_head(y, r) for r, (y, _) in enumerate(outs))
    loss.backward()
    return xs, outs, gamma, beta, ex


@pytest.mark.parametrize("shape_kind", ["sparse", "dense"])
def test_syncbn_matches_reference_formulas_world2(shape_kind):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, shape_kind, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = {}
    for _ in range(2):
        item = q.get(timeout=180)
        assert not (isinstance(item[1], str) and item[1] == "error"), item[2]
        got[item[0]] = item[1:]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    xs, outs, gamma, beta, ex = _expected(shape_kind)
    for r in range(2):
        y, dx, dg, db, rm, rv = got[r]
        if shape_kind == "dense":  # same N on both ranks -> the single-process function IS the reference's
            assert np.allclose(y, outs[r][0].detach().numpy(), atol=2e-5)
            assert np.allclose(dx, xs[r].grad.numpy(), rtol=2e-4, atol=2e-4)
            assert np.allclose(dg, gamma.grad.numpy(), rtol=2e-4, atol=2e-3)  # SUM over ranks
            assert np.allclose(db, beta.grad.numpy(), rtol=2e-4, atol=2e-3)
        else:  # different N per rank: cf differs per rank, forward values still follow the formula
            assert np.allclose(y, outs[r][0].detach().numpy(), atol=2e-5)
            assert np.isfinite(dx).all() and np.allclose(dg, got[1 - r][2]) and np.allclose(db, got[1 - r][3])
        assert np.allclose(rm, 0.01 * ex.detach().numpy(), atol=1e-6)
        assert np.allclose(rv, 0.99 + 0.01 * outs[r][1].detach().numpy(), atol=1e-5)


def _exchange_worker(rank, world, port, q):
    """the [2C]-vector algebra of the fused conv nodes (syncbn.exchange_moments / exchange_backward_sums) with the HIP
    sweeps' formulas written out in torch: d3d_bn_apply(training=2) normalises with the handed-in mean | variance,
    d3d_bn_bwd_dx computes gamma * invstd * (dz - S0 / N - xhat * S1 / N) from the sums it is given"""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from det3d_amd import syncbn
        from det3d_amd.syncbn import DistributedSyncBN, _SyncBNFn

        x = _data(rank, "sparse")
        n, c = x.shape
        mk = lambda: DistributedSyncBN(c, eps=1e-3, momentum=0.01).train()
        bn, ref = mk(), mk()
        for m in (bn, ref):
            with torch.no_grad():
                m.weight.copy_(torch.linspace(0.5, 2.0, c))
                m.bias.copy_(torch.linspace(-1.0, 1.0, c))
        assert syncbn.fused_sync(bn) and not syncbn.fused_sync(mk().eval()) and not syncbn.fused_sync(torch.nn.BatchNorm1d(c))
        # ---- the fused nodes' data flow
        sums = torch.cat([x.sum(0), (x * x).sum(0), torch.full((5,), 99.0)])       # conv epilogue: sum | sum of squares | slabs
        mv, cf = syncbn.exchange_moments(bn, sums, n)
        mean, invstd = mv[:c], (mv[c:] + bn.eps).rsqrt()                              # bn_apply_kernel, training == 2
        xhat = (x - mean) * invstd
        z = xhat * bn.weight + bn.bias
        y = z.clamp(min=0)                                                            # relu = 1
        dy = 2 * y * torch.linspace(0.5, 1.5, c) ** 2 * (1.0 + 0.5 * rank)
        dz = dy * (z > 0)
        partial = torch.cat([(dz * xhat).sum(0), dz.sum(0), torch.full((3,), -7.0)])  # dgrad epilogue / bn_bwd_reduce
        p2, g_gamma, g_beta = syncbn.exchange_backward_sums(bn, partial, cf, invstd)
        dx = bn.weight * invstd * (dz - p2[c:] / n - xhat * p2[:c] / n)                # bn_bwd_dx_kernel
        # ---- the torch-op formulation pinned to the reference formulas above
        xr = x.clone().requires_grad_(True)
        yr = torch.relu(_SyncBNFn.apply(xr, ref.weight, ref.bias, ref.running_mean, ref.running_var, ref.momentum, ref.eps, None))
        (((yr * torch.linspace(0.5, 1.5, c)) ** 2).sum() * (1.0 + 0.5 * rank)).backward()
        err = dict(y=float((y - yr.detach()).abs().max()), dx=float((dx.detach() - xr.grad).abs().max() / xr.grad.abs().max()),
                   dgamma=float((g_gamma.detach() - ref.weight.grad).abs().max() / ref.weight.grad.abs().max()),
                   dbeta=float((g_beta.detach() - ref.bias.grad).abs().max() / ref.bias.grad.abs().max()),
                   rmean=float((bn.running_mean - ref.running_mean).abs().max()),
                   rvar=float((bn.running_var - ref.running_var).abs().max()))
        q.put((rank, err))
    except Exception:
        import traceback

        q.put((rank, "error", traceback.format_exc()))
    finally:
        dist.destroy_process_group()


def test_fused_node_exchange_algebra_equals_the_reference_formulation_world2():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_exchange_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    for _ in range(2):
        item = q.get(timeout=180)
        assert not (isinstance(item[1], str) and item[1] == "error"), item[2]
        for k, v in item[1].items():
            assert v < 2e-5, (item[0], k, v)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0


def test_syncbn_without_process_group_is_plain_bn():
    from det3d_amd.nn_utils import build_norm_layer
    from det3d_amd.syncbn import DistributedSyncBN

    _, bn = build_norm_layer(dict(type="SyncBN", eps=1e-3, momentum=0.01), 8)
    assert isinstance(bn, DistributedSyncBN)
    ref = torch.nn.BatchNorm1d(8, eps=1e-3, momentum=0.01)
    x = torch.randn(40, 8)
    assert torch.allclose(bn(x), ref(x), atol=1e-6)
    assert torch.allclose(bn.running_var, ref.running_var)
    bn.eval()
    ref.eval()
    assert torch.allclose(bn(x), ref(x), atol=1e-6)
    assert set(bn.state_dict()) == set(ref.state_dict())
