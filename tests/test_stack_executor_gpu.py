"""csrc/d3d_stack.hip (d3d_sparse_stack_forward / _backward, ABI 12): the fused sparse stack's launches from one foreign call
per direction.  The call walks the same entry points in the same order as det3d_amd/sparse_stack.py does from Python, so
everything must be BIT-identical: losses, every parameter gradient, BatchNorm running statistics, over several steps, in
both precisions, with and without the weight-gradient stream -- and the executor must really have run."""
import copy
import os

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _model():
    from det3d_amd import Config, build_detector
    from oracle import model_ref as M

    cfg = Config.fromfile(os.path.join(ROOT, "det3d_amd", "configs", "kitti_car_second.py"))
    model = build_detector(cfg.model, cfg.train_cfg, cfg.test_cfg)
    M.fill_deterministic(model)
    return model.cuda().train()


def _examples(n, batch=2):
    from det3d_amd import synthetic

    ds = synthetic.KittiCarSynthetic(n_points=9000, n_seeds=4)
    dev = torch.device("cuda", 0)
    return [synthetic.make_example(ds.device_batch(i, batch, dev), synthetic.KITTI_CAR) for i in range(n)]


def _steps(model, examples, executor, dtype, calls=None):
    from det3d_amd import _lib, runtime

    prev = runtime.get("stack_executor")
    runtime.configure(stack_executor=executor)
    L = _lib.lib()
    out = []
    try:
        for ex in examples:
            model.zero_grad(set_to_none=True)
            ctx = torch.autocast("cuda", dtype=dtype) if dtype is not None else torch.autocast("cuda", enabled=False)
            with ctx:
                losses = model(ex, return_loss=True)
            loss = sum(losses["loss"])
            loss.backward()
            out.append((float(loss.detach()), {k: p.grad.clone() for k, p in model.named_parameters() if p.grad is not None}))
    finally:
        runtime.configure(stack_executor=prev)
    return out


@pytest.mark.parametrize("dtype", [None, torch.bfloat16])
@pytest.mark.parametrize("wgrad_stream", ["all", "off"])
def test_c_executor_is_bit_identical_to_the_python_sequence(dtype, wgrad_stream, monkeypatch):
    from det3d_amd import runtime, sparse_stack

    prev = runtime.get("wgrad_stream")
    runtime.configure(wgrad_stream=wgrad_stream)
    # count the executor's calls: the comparison means nothing if both runs took the same path
    calls = {"fwd": 0, "bwd": 0}
    orig = sparse_stack._StackFn._backward_c

    def counted(*a, **k):
        calls["bwd"] += 1
        return orig(*a, **k)

    monkeypatch.setattr(sparse_stack._StackFn, "_backward_c", staticmethod(counted))
    try:
        a_model = _model()
        b_model = copy.deepcopy(a_model)
        examples = _examples(3)
        a = _steps(a_model, examples, "python", dtype)
        assert calls["bwd"] == 0
        b = _steps(b_model, examples, "c", dtype)
        assert calls["bwd"] >= 3, "the C executor did not run"
    finally:
        runtime.configure(wgrad_stream=prev)
    for (la, ga), (lb, gb) in zip(a, b):
        assert la == lb, (la, lb)
        assert ga.keys() == gb.keys()
        bad = [k for k in ga if not torch.equal(ga[k], gb[k])]
        assert not bad, "not bit-equal: %s" % bad[:6]
    sa, sb = a_model.state_dict(), b_model.state_dict()
    for k in sa:
        if k.endswith("running_mean") or k.endswith("running_var"):
            assert torch.equal(sa[k], sb[k]), k


def test_c_executor_refuses_missing_buffers():
    """the entry points validate their descriptors instead of launching on null pointers"""
    from det3d_amd import _lib

    L = _lib.lib()
    desc = (_lib.StackLayer * 1)()
    st = _lib.stream_ptr(torch.device("cuda", 0))
    assert L.d3d_sparse_stack_forward(desc, 1, _lib.D3D_F32, st) != 0
    assert L.d3d_sparse_stack_forward(None, 0, _lib.D3D_F32, st) != 0
    x = torch.zeros(4, device="cuda")
    assert L.d3d_sparse_stack_backward(desc, 1, _lib.D3D_F32, x.data_ptr(), st, None) != 0
