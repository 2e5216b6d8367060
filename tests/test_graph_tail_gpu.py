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


def _run(model, examples, graphed, dtype):
    from det3d_amd import graph_tail

    graph_tail.GRAPH_TAIL = graphed
    out = []
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
        graph_tail.GRAPH_TAIL = True
    return out


@pytest.mark.parametrize("dtype", [None, torch.bfloat16])
def test_graphed_tail_equals_the_eager_tail(dtype):
    cfg, eager = _model("kitti_car_second.py")
    graphed = copy.deepcopy(eager)
    examples = _kitti_examples(3)
    a = _run(eager, examples, False, dtype)
    b = _run(graphed, examples, True, dtype)
    assert "_graph_tails" in graphed.__dict__ and len(graphed.__dict__["_graph_tails"]) == 1, \
        "the tail was not captured: %s" % graphed.__dict__.get("_graph_tail_failed")
    assert "_graph_tails" not in eager.__dict__
    # same kernels on the same data; MIOpen's split-K weight gradients and the BN atomics are order-dependent
    tol = 2e-4 if dtype is None else 2e-2
    for (la, lla, ga), (lb, llb, gb) in zip(a, b):
        assert lb == pytest.approx(la, rel=tol)
        assert llb == pytest.approx(lla, rel=tol)
        assert ga.keys() == gb.keys()
        for k in ga:
            scale = float(ga[k].abs().max()) + 1e-12
            assert float((ga[k] - gb[k]).abs().max()) <= tol * 5 * scale + 1e-7, k
    # BatchNorm running statistics advanced exactly as in the eager run (the capture warm-up is rolled back)
    sa, sb = eager.state_dict(), graphed.state_dict()
    for k in sa:
        if k.endswith("running_mean") or k.endswith("running_var"):
            assert torch.allclose(sa[k], sb[k], rtol=1e-3 if dtype is None else 2e-2, atol=1e-5), k


def test_graphed_tail_multi_task_head_and_train_step():
    """CBGS head (several task groups, direction classifier) through the graphs inside TrainStep: the loss goes
    down on a repeated batch and the step leaves no gradient aliasing a graph-owned buffer behind."""
    from det3d_amd import synthetic
    from det3d_amd.trainer import TrainStep

    cfg, model = _model("nusc_cbgs.py")
    ds = synthetic.TaskGroupSynthetic(cfg, n_points=30000, n_seeds=2)
    dev = torch.device("cuda", 0)
    ex = ds.make_example(ds.device_batch(0, 2, dev))
    step = TrainStep(model, total_steps=20, lr_max=1e-3, autocast_dtype=torch.bfloat16)
    losses = [float(step(ex)) for _ in range(6)]
    assert "_graph_tails" in model.__dict__, model.__dict__.get("_graph_tail_failed")
    assert all(torch.isfinite(torch.tensor(losses))) and min(losses[1:]) < losses[0]
