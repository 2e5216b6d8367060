"""graph_tail.py: the RPN / head / loss tail replayed as two hipGraphs must be the same function as the eager tail
(losses, every parameter gradient, BatchNorm running statistics), in fp32 and under bf16 autocast, across
several steps with changing inputs, and must hand gradients to the data-parallel hooks."""
import copy
import os

import pytest
import torch

from conftest import ROOT

pytestmark = pytest.mark.gpu


def _model(name):
    from det3d_amd import Config, build_detector
    from oracle import model_ref as M

    cfg = Config.fromfile(os.path.join(ROOT, "det3d_amd", "configs", name))
    model = build_detector(cfg.model, cfg.train_cfg, cfg.test_cfg)
    M.fill_deterministic(model)
    return cfg, model.cuda().train()


def _kitti_examples(n, batch=2):
    from det3d_amd import synthetic

    ds = synthetic.KittiCarSynthetic(n_points=9000, n_seeds=4)
    dev = torch.device("cuda", 0)
    return [synthetic.make_example(ds.device_batch(i, batch, dev), synthetic.KITTI_CAR) for i in range(n)]


def _bias_before_bn(model):
    """conv biases directly followed by a train-mode BatchNorm (SparseSequential order conv, bn): zero true gradient"""
    from det3d_amd import spconv

    out = set()
    for name, m in model.named_modules():
        if isinstance(m, spconv.SparseSequential):
            mods = list(m._modules.items())
            for (k0, a), (_, b) in zip(mods, mods[1:]):
                if isinstance(a, spconv.SparseConvolution) and a.bias is not None and \
                        isinstance(b, (torch.nn.BatchNorm1d,)):
                    out.add("%s.%s.bias" % (name, k0))
    return out


def _rel_l2(a, b):
    return float((a - b).norm()) / (float(a.norm()) + 1e-20)


def _assert_same(a, a2, b, dtype):
    """a, a2: two runs of the reference path on identical copies of the model, b: the path under test.  Every
    reduction of the step is summed in a fixed order (round 3, csrc/d3d_fold.h), so the reference path repeats
    bit for bit -- and a path that issues the same kernels on the same data in the same order must be bit-equal to
    it as well (round 2 had to calibrate this comparison against the step's own run-to-run noise)."""
    for (la, lla, ga), (la2, _, ga2), (lb, llb, gb) in zip(a, a2, b):
        assert la == la2 and ga.keys() == ga2.keys() == gb.keys()
        for k in ga:
            assert torch.equal(ga[k], ga2[k]), "the reference path does not repeat: %s" % k
        assert lb == la, (la, lb)
        bad = [k for k in ga if not torch.equal(ga[k], gb[k])]
        assert not bad, "not bit-equal: %s" % bad[:6]


def _run(model, examples, graphed, dtype):
    from det3d_amd import graph_tail

    prev = graph_tail.GRAPH_TAIL
    graph_tail.GRAPH_TAIL = graphed
    out = []
    # everything this repo launches sums in a fixed order; the head's 1x1 convolutions are MIOpen's, whose default
    # weight-gradient solvers use atomics: ask for its deterministic ones so that whole steps can be compared bit for bit
    det = torch.backends.cudnn.flags(enabled=True, benchmark=False, deterministic=True)
    det.__enter__()
    try:
        for ex in examples:
            model.zero_grad(set_to_none=True)
            ctx = torch.autocast("cuda", dtype=dtype) if dtype is not None else torch.autocast("cuda", enabled=False)
            with ctx:
                losses = model(ex, return_loss=True)
            loss = sum(losses["loss"])
            loss.backward()
            out.append((float(loss), [float(v) for v in losses["loc_loss_reduced"]],
                        {k: p.grad.float().clone() for k, p in model.named_parameters() if p.grad is not None}))
    finally:
        det.__exit__(None, None, None)
        graph_tail.GRAPH_TAIL = prev
    return out


@pytest.mark.parametrize("dtype", [None, torch.bfloat16])
def test_graphed_tail_equals_the_eager_tail(dtype):
    cfg, eager = _model("kitti_car_second.py")
    eager2 = copy.deepcopy(eager)
    graphed = copy.deepcopy(eager)
    examples = _kitti_examples(3)
    a = _run(eager, examples, False, dtype)
    a2 = _run(eager2, examples, False, dtype)  # the run-to-run noise of the eager step itself
    b = _run(graphed, examples, True, dtype)
    assert "_graph_tails" in graphed.__dict__ and len(graphed.__dict__["_graph_tails"]) == 1, \
        "the tail was not captured: %s" % graphed.__dict__.get("_graph_tail_failed")
    assert "_graph_tails" not in eager.__dict__
    _assert_same(a, a2, b, dtype)
    # BatchNorm running statistics advanced exactly as in the eager run (the capture warm-up is rolled back)
    sa, sb = eager.state_dict(), graphed.state_dict()
    for k in sa:
        if k.endswith("running_mean") or k.endswith("running_var"):
            assert torch.equal(sa[k], sb[k]), k


def test_graphed_tail_multi_task_head_and_train_step():
    """CBGS head (several task groups, direction classifier) through the graphs inside TrainStep: the loss goes
    down on a repeated batch and the step leaves no gradient aliasing a graph-owned buffer behind."""
    from det3d_amd import graph_tail, synthetic
    from det3d_amd.trainer import TrainStep

    cfg, model = _model("nusc_cbgs.py")
    ds = synthetic.TaskGroupSynthetic(cfg, n_points=30000, n_seeds=2)
    dev = torch.device("cuda", 0)
    ex = ds.make_example(ds.device_batch(0, 2, dev))
    step = TrainStep(model, total_steps=40, lr_max=1e-3, autocast_dtype=torch.bfloat16)
    # MIOpen find mode on, as in bench.py: the solvers it selects must not end up inside the graphs (their split-K
    # weight gradients accumulated across replays -> gradient norm inf within 20 steps)
    prev, prev_g = torch.backends.cudnn.benchmark, graph_tail.GRAPH_TAIL
    torch.backends.cudnn.benchmark = True
    graph_tail.GRAPH_TAIL = True
    try:
        losses, norms = [], []
        for _ in range(25):
            losses.append(float(step(ex)))
            norms.append(float(step.last["grad_norm"]))
    finally:
        torch.backends.cudnn.benchmark = prev
        graph_tail.GRAPH_TAIL = prev_g
    assert "_graph_tails" in model.__dict__, model.__dict__.get("_graph_tail_failed")
    assert all(torch.isfinite(torch.tensor(losses))) and all(torch.isfinite(torch.tensor(norms))), (losses, norms)
    assert min(losses[1:]) < losses[0]


def test_graphed_tail_follows_the_optimizer_steps_bit_for_bit():
    """the packed weights of the neck's convolutions are cached per parameter version; a capture that
    hit the cache would bake the capture-time weights into the graph, and every replay after an optimizer step would
    convolve with stale weights.  Several whole TrainStep steps, graphed against eager, on copies of one model: the
    losses of every step and every parameter afterwards must be bit-equal (the pack launch is part of the graph)."""
    from det3d_amd import graph_tail
    from det3d_amd.trainer import TrainStep

    cfg, eager = _model("kitti_car_second.py")
    graphed = copy.deepcopy(eager)
    examples = _kitti_examples(2)

    def train(model, on):
        prev = graph_tail.GRAPH_TAIL
        graph_tail.GRAPH_TAIL = on
        try:
            step = TrainStep(model, total_steps=20, lr_max=3e-3, autocast_dtype=None)
            with torch.backends.cudnn.flags(enabled=True, benchmark=False, deterministic=True):
                return [float(step(examples[i % 2])) for i in range(5)]
        finally:
            graph_tail.GRAPH_TAIL = prev

    la = train(eager, False)
    lb = train(graphed, True)
    assert "_graph_tails" in graphed.__dict__, graphed.__dict__.get("_graph_tail_failed")
    assert la == lb, (la, lb)
    assert la[0] != la[2]  # the weights did move between two visits of the same batch
    pa, pb = dict(eager.named_parameters()), dict(graphed.named_parameters())
    bad = [k for k in pa if not torch.equal(pa[k], pb[k])]
    assert not bad, bad[:6]
    # an eager pass afterwards (validation) re-packs into the eager cache and must leave the graphs intact
    graph_tail.GRAPH_TAIL = False
    try:
        with torch.no_grad():
            graphed(examples[0], return_loss=True)
    finally:
        graph_tail.GRAPH_TAIL = True
    try:
        step = TrainStep(graphed, total_steps=20, lr_max=3e-3, autocast_dtype=None)
        assert torch.isfinite(torch.tensor(float(step(examples[0]))))
    finally:
        graph_tail.GRAPH_TAIL = False


def test_prefetched_geometry_is_the_inline_geometry():
    """prefetch.GeometryPrefetcher: example + rulebooks built on the side stream give the same step as the inline
    path, and the backbone really consumes the prefetched plan (no second planning pass)."""
    from det3d_amd import graph_tail, synthetic
    from det3d_amd.prefetch import GeometryPrefetcher

    cfg, model = _model("kitti_car_second.py")
    ds = synthetic.KittiCarSynthetic(n_points=9000, n_seeds=4)
    dev = torch.device("cuda", 0)
    make = lambda i: synthetic.make_example(ds.device_batch(i, 2, dev), synthetic.KITTI_CAR)
    pf = GeometryPrefetcher(make, model, dev)
    prev_g = graph_tail.GRAPH_TAIL
    graph_tail.GRAPH_TAIL = False
    det = torch.backends.cudnn.flags(enabled=True, benchmark=False, deterministic=True)  # see _run
    det.__enter__()
    try:
        for i in range(3):
            ex_a = make(i)
            model.zero_grad(set_to_none=True)
            la = sum(model(ex_a, return_loss=True)["loss"])
            la.backward()
            ga = {k: p.grad.clone() for k, p in model.named_parameters() if p.grad is not None}
            ex_b = pf.prepare(i)
            assert "sparse_plan" in ex_b and pf.stream is not None
            calls = []
            orig = model.backbone.middle_conv.plan
            model.backbone.middle_conv.plan = lambda x: calls.append(1) or orig(x)
            try:
                model.zero_grad(set_to_none=True)
                lb = sum(model(ex_b, return_loss=True)["loss"])
                lb.backward()
            finally:
                del model.backbone.middle_conv.plan
            assert not calls, "the backbone planned its rulebooks again"
            assert float(lb) == float(la)
            for k, g in ga.items():
                # same kernels on the same geometry, every sum in a fixed order: bit-equal
                assert torch.equal(g, model.get_parameter(k).grad), k
            # the same again with the build on the prefetcher's worker thread, two builds in flight
            pend = [pf.submit(i), pf.submit((i + 1) % 4)]
            ex_c = pend[0].result()
            assert "sparse_plan" in ex_c and torch.equal(ex_c["coordinates"], ex_b["coordinates"])
            model.zero_grad(set_to_none=True)
            lc = sum(model(ex_c, return_loss=True)["loss"])
            lc.backward()
            assert float(lc) == float(la)
            for k, g in ga.items():
                assert torch.equal(g, model.get_parameter(k).grad), k
            assert torch.equal(pend[1].result()["coordinates"], make((i + 1) % 4)["coordinates"])
    finally:
        pf.close()
        det.__exit__(None, None, None)
        graph_tail.GRAPH_TAIL = prev_g


@pytest.mark.parametrize("dtype", [None, torch.bfloat16])
def test_fused_sparse_stack_equals_the_per_layer_path(dtype):
    """sparse_stack.py: the whole SpMiddleFHD stack as one autograd node issues the same kernels as the per-layer
    Functions -- the same loss, gradients and running statistics, bit for bit."""
    from det3d_amd import graph_tail, sparse_stack

    cfg, a_model = _model("kitti_car_second.py")
    a2_model = copy.deepcopy(a_model)
    b_model = copy.deepcopy(a_model)
    examples = _kitti_examples(2)
    calls = []
    orig = sparse_stack.run
    try:
        sparse_stack.FUSE_STACK = False
        a = _run(a_model, examples, False, dtype)
        a2 = _run(a2_model, examples, False, dtype)
        sparse_stack.FUSE_STACK = True
        sparse_stack.run = lambda sp, layers, x: calls.append(len(layers)) or orig(sp, layers, x)
        b = _run(b_model, examples, False, dtype)
    finally:
        sparse_stack.FUSE_STACK = True
        sparse_stack.run = orig
    assert calls == [14, 14], calls  # all 14 conv-BN-ReLU triples of SpMiddleFHD in one node, once per step
    _assert_same(a, a2, b, dtype)
    sa, sb = a_model.state_dict(), b_model.state_dict()
    for k in sa:
        if k.endswith("running_mean") or k.endswith("running_var"):
            assert torch.equal(sa[k], sb[k]), k
