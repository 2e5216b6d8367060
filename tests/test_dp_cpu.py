"""Data-parallel step on CPU: world_size-2 gloo processes.  The flat-bucket gradient mean must
equal the single-process gradient of the concatenated batch (no BN in the test net: exact)."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

from conftest import ROOT


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _net():
    # BatchNorm in eval mode: an affine layer for the gradient check, but it brings int64
    # (num_batches_tracked) and float32 buffers, so broadcast_params has several dtype groups
    torch.manual_seed(0)
    return torch.nn.Sequential(torch.nn.Conv2d(3, 8, 3, padding=1), torch.nn.BatchNorm2d(8), torch.nn.ReLU(),
                               torch.nn.Conv2d(8, 4, 1)).eval()


def _worker_overlapped(rank, world, port, q, n_buckets):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world),
                      MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    from det3d_amd import dist as D

    D.init_dist(backend="gloo")
    net = _net()
    D.broadcast_params(net)
    x = torch.randn(2, 3, 6, 6, generator=torch.Generator().manual_seed(100 + rank))
    bucket = D.OverlappedGradBuckets(net.parameters(), n_buckets)
    out = []
    for step in range(2):  # two steps: the hooks must re-arm cleanly
        bucket.zero()
        net(x * (1 + step)).pow(2).mean().backward()
        total = bucket.finish(max_norm=1e9 if step == 0 else 0.05)
        out.append(([p.grad.numpy().copy() for p in net.parameters()], float(total)))
    D.barrier()
    q.put((rank, out, [len(b) for b in bucket.buckets]))
    torch.distributed.destroy_process_group()


@pytest.mark.parametrize("n_buckets", [1, 3])
def test_overlapped_buckets_two_ranks_equal_single_process_mean(n_buckets):
    """OverlappedGradBuckets (all-reduces started from autograd hooks during backward, a15 / 8e): the gradients
    every rank ends with are the mean over ranks of the per-rank gradients, the norm is the global norm, the
    clip scales all buckets by one coefficient, and a second step works on re-armed hooks."""
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker_overlapped, args=(r, world, port, q, n_buckets)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=120) for _ in range(world)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert sum(res[0][2]) == len(list(_net().parameters())) and 1 <= len(res[0][2]) <= n_buckets
    assert n_buckets == 1 or len(res[0][2]) >= 2
    xs = [torch.randn(2, 3, 6, 6, generator=torch.Generator().manual_seed(100 + r)) for r in range(world)]
    for step in range(2):
        net = _net()
        (sum(net(x * (1 + step)).pow(2).mean() for x in xs) / world).backward()
        ref = [p.grad for p in net.parameters()]
        norm = float(torch.sqrt(sum((r ** 2).sum() for r in ref)))
        coef = 1.0 if step == 0 else min(1.0, 0.05 / (norm + 1e-6))
        for rank, out, _ in res:
            grads, total = out[step]
            assert total == pytest.approx(norm, rel=1e-5)
            for g, r in zip(grads, ref):
                assert np.allclose(g, (r * coef).numpy(), atol=1e-6)


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world),
                      MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    from det3d_amd import dist as D

    r, lr, w = D.init_dist(backend="gloo")
    assert (r, w) == (rank, world) and D.world_size() == world and D.get_rank() == rank
    net = _net()
    if rank == 1:  # diverge on purpose: broadcast_params must bring rank 1 back
        with torch.no_grad():
            for p in net.parameters():
                p.add_(1.0)
            for b in net.buffers():
                b.add_(3)
    D.broadcast_params(net)
    g = torch.Generator().manual_seed(100 + rank)
    x = torch.randn(2, 3, 6, 6, generator=g)
    bucket = D.FlatGradBucket(net.parameters())
    bucket.zero()
    # per-rank loss is a mean over the local batch; the DP gradient is the mean over ranks
    net(x).pow(2).mean().backward()
    bucket.gather()
    bucket.allreduce_mean()
    total = bucket.clip_norm_(1e9)
    objs = D.all_gather_objects(dict(rank=rank, n=int(x.shape[0])))
    D.barrier()
    # numpy through the queue: torch tensors are shared by fd and the worker may exit first
    q.put((rank, [p.grad.numpy().copy() for p in net.parameters()], float(total), objs,
           [p.detach().numpy().copy() for p in net.parameters()] + [b.numpy().copy() for b in net.buffers()]))
    torch.distributed.destroy_process_group()


def test_two_rank_gradient_mean_equals_single_process():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=120) for _ in range(world)], key=lambda t: t[0])
    res = [(r, [torch.from_numpy(g) for g in gs], tot, objs, [torch.from_numpy(np.asarray(p)) for p in ps])
           for r, gs, tot, objs, ps in res]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    # reference: one process, both shards
    net = _net()
    xs = [torch.randn(2, 3, 6, 6, generator=torch.Generator().manual_seed(100 + r)) for r in range(world)]
    loss = sum(net(x).pow(2).mean() for x in xs) / world
    loss.backward()
    ref = [p.grad for p in net.parameters()]
    for rank, grads, total, objs, params in res:
        for g, r in zip(grads, ref):
            assert torch.allclose(g, r, atol=1e-6)
        assert total == pytest.approx(float(torch.sqrt(sum((r ** 2).sum() for r in ref))), rel=1e-5)
        assert [o["rank"] for o in objs] == [0, 1]
        ref_net = _net()
        for p, q0 in zip(params, list(ref_net.parameters()) + list(ref_net.buffers())):
            assert torch.equal(p, q0)  # rank 1 was reset to rank 0's weights and buffers
    # both ranks ended with identical gradients
    for a, b in zip(res[0][1], res[1][1]):
        assert torch.equal(a, b)


def test_flat_bucket_clip_and_views_single_process():
    from det3d_amd import dist as D

    net = _net()
    b = D.FlatGradBucket(net.parameters())
    b.zero()
    net(torch.randn(2, 3, 5, 5)).sum().backward()
    flat = b.gather()
    assert flat.numel() == sum(p.numel() for p in net.parameters())
    for p in net.parameters():
        assert p.grad.data_ptr() >= flat.data_ptr() and p.grad.shape == p.shape
    n0 = float(b.clip_norm_(0.5))
    assert float(torch.linalg.vector_norm(flat)) == pytest.approx(min(n0, 0.5), rel=1e-4)


def test_one_cycle_schedule_matches_reference_formula():
    from det3d_amd.trainer import OneCycle

    s = OneCycle(1000, lr_max=3e-3, moms=(0.95, 0.85), div_factor=10.0, pct_start=0.4)
    assert s.at(0) == pytest.approx((3e-4, 0.95))
    assert s.at(400)[0] == pytest.approx(3e-3) and s.at(400)[1] == pytest.approx(0.85)
    lr_end, mom_end = s.at(1000)
    assert lr_end == pytest.approx(3e-4 / 1e4, rel=1e-3) and mom_end == pytest.approx(0.95)
    assert s.at(200)[0] == pytest.approx(3e-4 + (3e-3 - 3e-4) / 2)


def test_frame_partition_equals_reference_distributed_group_sampler():
    """det3d_amd.sampler vs indices produced by the reference's DistributedGroupSampler (tests/golden/sampler.npz):
    identical per rank and epoch; ranks tile the padded epoch without gaps."""
    import os

    import numpy as np

    from conftest import GOLDEN
    from det3d_amd import sampler

    z = np.load(os.path.join(GOLDEN, "sampler.npz"))
    for tag in ("a", "b", "c"):
        flag = z[tag + "_flag"]
        spg, world = (int(v) for v in z[tag + "_cfg"])

        class DS:
            pass

        ds = DS()
        ds.flag = flag
        for epoch in (0, 3):
            seen = []
            for rank in range(world):
                s = sampler.DistributedGroupSampler(ds, samples_per_gpu=spg, num_replicas=world, rank=rank)
                s.set_epoch(epoch)
                mine = list(iter(s))
                assert mine == z["%s_e%d_r%d" % (tag, epoch, rank)].tolist(), (tag, epoch, rank)
                assert len(mine) == len(s) and len(mine) % spg == 0
                seen += mine
            assert set(seen) == set(range(len(flag)))  # every frame is visited each epoch


def test_prefetcher_walks_nested_containers_and_is_transparent_without_a_gpu():
    """prefetch.GeometryPrefetcher: on a machine without a GPU it degrades to calling make_example inline; the
    tensor walker used for record_stream() reaches tensors inside dicts, lists and plain objects exactly once."""
    import torch

    from det3d_amd.prefetch import GeometryPrefetcher, _tensors

    class Box:
        def __init__(self):
            self.t = torch.ones(2)
            self.loop = self  # cycles must not hang the walk

    ex = {"a": torch.zeros(3), "b": [torch.zeros(1), (torch.zeros(4), 5)], "c": {"d": Box()}, "e": "text"}
    found = list(_tensors(ex, set()))
    assert sorted(t.numel() for t in found) == [1, 2, 3, 4]
    pf = GeometryPrefetcher(lambda i: {"i": i, "x": torch.full((2,), float(i))}, detector=None, enabled=True)
    if not torch.cuda.is_available():
        assert pf.enabled is False
    out = pf.prepare(7)
    assert out["i"] == 7 and out["x"].tolist() == [7.0, 7.0] and "sparse_plan" not in out


def test_sparse_stack_segments_only_cover_fusable_conv_bn_relu_runs():
    """sparse_stack.segment: which layers of a SparseSequential go through the single-node path is decided from the
    module list and the planned rulebooks alone -- checkable without a GPU with stand-in rulebooks."""
    import types

    import torch
    from torch import nn

    from det3d_amd import sparse_stack, spconv

    def rb(n_in, n_out, subm):
        return types.SimpleNamespace(n_in=n_in, n_out=n_out, subm=subm, kvol=27)

    convs = [spconv.SubMConv3d(4, 16, 3, bias=False, indice_key="s0"), spconv.SubMConv3d(16, 16, 3, bias=False, indice_key="s0"),
             spconv.SparseConv3d(16, 32, 3, 2, padding=1, bias=False), spconv.SubMConv3d(32, 32, 3, bias=False, indice_key="s1")]
    mods = []
    for c in convs:
        mods += [c, nn.BatchNorm1d(c.out_channels), nn.ReLU()]
    x = types.SimpleNamespace(indices=torch.zeros((100, 4), dtype=torch.int32), features=torch.zeros((100, 4), dtype=torch.bfloat16),
                              indice_dict={"s0": rb(100, 100, True), convs[2]._rulebook_key(): rb(100, 40, False),
                                           "s1": rb(40, 40, True)})
    for m in mods:
        m.train()
    layers, nxt = sparse_stack.segment(spconv, mods, 0, x)
    assert len(layers) == 4 and nxt == len(mods) and [ly.relu for ly in layers] == [True] * 4
    assert [ly.pad_cin for ly in layers] == [16, 16, 16, 32]        # the 4-channel stem is padded to an MFMA tile
    # a BatchNorm in eval mode ends the run in front of it
    mods[4].eval()
    layers, nxt = sparse_stack.segment(spconv, mods, 0, x)
    assert len(layers) == 1 and nxt == 3
    mods[4].train()
    # a missing rulebook, a row-count mismatch, an unsupported width or a conv without a following BatchNorm: no run
    x2 = types.SimpleNamespace(indices=x.indices, features=x.features, indice_dict={})
    assert sparse_stack.segment(spconv, mods, 0, x2)[0] == []
    x3 = types.SimpleNamespace(indices=torch.zeros((99, 4), dtype=torch.int32), features=x.features, indice_dict=x.indice_dict)
    assert sparse_stack.segment(spconv, mods, 0, x3)[0] == []
    wide = [spconv.SubMConv3d(16, 160, 3, bias=False, indice_key="s0"), nn.BatchNorm1d(160), nn.ReLU()]
    wide[1].train()
    assert sparse_stack.segment(spconv, wide, 0, x)[0] == []
    assert sparse_stack.segment(spconv, [convs[1], nn.ReLU()], 0, x)[0] == []
    # the parameter list handed to autograd follows the layer order: weight, [bias], gamma, beta
    layers, _ = sparse_stack.segment(spconv, mods, 0, x)
    ps = sparse_stack.params_of(layers)
    assert len(ps) == 12 and ps[0] is convs[0].weight and ps[1] is mods[1].weight and ps[2] is mods[1].bias


def test_single_rank_clip_equals_the_flat_bucket():
    """LocalGradClip (one rank: no concatenation) clips to the same gradients and reports the same norm as
    FlatGradBucket, including a parameter that received no gradient."""
    import torch

    from det3d_amd import dist as D

    torch.manual_seed(0)
    net_a = torch.nn.Sequential(torch.nn.Linear(6, 8), torch.nn.ReLU(), torch.nn.Linear(8, 3))
    unused = torch.nn.Parameter(torch.ones(4))
    net_b = torch.nn.Sequential(torch.nn.Linear(6, 8), torch.nn.ReLU(), torch.nn.Linear(8, 3))
    net_b.load_state_dict(net_a.state_dict())
    unused_b = torch.nn.Parameter(torch.ones(4))
    x = torch.randn(5, 6)
    outs = []
    for net, extra, cls in ((net_a, unused, D.FlatGradBucket), (net_b, unused_b, D.LocalGradClip)):
        params = list(net.parameters()) + [extra]
        b = cls(params)
        b.zero()
        (net(x) ** 2).sum().backward()
        b.gather()
        b.allreduce_mean()
        total = b.clip_norm_(0.5)
        outs.append((float(total), [p.grad.clone() for p in params]))
    assert outs[0][0] == pytest.approx(outs[1][0], rel=1e-6)
    for ga, gb in zip(outs[0][1], outs[1][1]):
        assert torch.allclose(ga, gb, rtol=1e-6, atol=1e-8)
    assert float(torch.linalg.vector_norm(torch.cat([g.flatten() for g in outs[1][1]]))) == pytest.approx(0.5, rel=1e-4)
    assert torch.equal(outs[1][1][-1], torch.zeros(4))
