"""ABI 10 contexts on the GPU: two host threads, each with its own context and its own stream, run reducing launches (sparse
forward with the BatchNorm-statistics epilogue: ticket words from the context's ring) at the same time; the knobs of one
context do not reach the other."""
import ctypes
import threading

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _problem():
    from det3d_amd import _lib, spconv
    from test_spconv_gpu import _build, _random_sparse

    rng = np.random.default_rng(5)
    shape = [9, 40, 36]
    idx, _ = _random_sparse(rng, 3, shape, 1500, 1)
    idx = idx[np.lexsort((idx[:, 3], idx[:, 2], idx[:, 1], idx[:, 0]))]
    rb = _build(idx, 3, shape, [3, 3, 3], [1, 1, 1], [0, 0, 0], True)
    dev = torch.device("cuda", 0)
    torch.manual_seed(3)
    f = torch.randn(rb.n_out, 64, device=dev)
    w = torch.randn(27, 64, 64, device=dev) / 24.0
    return rb, f, w, dev


def _forward(L, rb, f, w, dev, stream, reps=1):
    """pack under the calling thread's context (the pack format follows knob 10), forward + statistics `reps` times"""
    from det3d_amd import _lib

    n = rb.n_out
    st = ctypes.c_void_p(stream.cuda_stream)
    outs = []
    with torch.cuda.stream(stream):
        wpack = torch.empty((27, 64, 64), device=dev)
        _lib.check(L.d3d_pack_conv_weights(w.data_ptr(), 27, 64, 64, _lib.D3D_F32, 1, wpack.data_ptr(), None, st), "pack")
        has_ep = L.d3d_sparse_conv_has_epilogue(64, 64, _lib.D3D_F32)
        nws = L.d3d_conv_epilogue_ws_floats(n, 64)
        for _ in range(reps):
            o = torch.full((n, 64), float("nan"), device=dev)
            ws = torch.zeros(nws, device=dev)
            ep = _lib.ConvEpilogue(stats=ws.data_ptr(), ws_floats=nws) if has_ep else None
            _lib.check(L.d3d_sparse_conv_fwd_ex(f.data_ptr(), wpack.data_ptr(), None, rb.nbr_out.data_ptr(), rb.cap_out, n, 27,
                                                64, 64, _lib.D3D_F32, o.data_ptr(), ctypes.byref(ep) if ep else None, st),
                       "fwd")
            outs.append((o, ws[:128].clone()))
    stream.synchronize()
    return outs


def test_two_threads_two_contexts_two_streams():
    from det3d_amd import _lib

    L = _lib.lib()
    rb, f, w, dev = _problem()
    ref_split = _forward(L, rb, f, w, dev, torch.cuda.Stream(dev))[0]
    assert L.d3d_set_tuning(10, 1) == 0      # the default context, exact-f32 test kernels
    try:
        ref_exact = _forward(L, rb, f, w, dev, torch.cuda.Stream(dev))[0]
    finally:
        L.d3d_set_tuning(10, 0)
    assert torch.isfinite(ref_split[0]).all() and not torch.equal(ref_split[0], ref_exact[0])
    assert (ref_split[0] - ref_exact[0]).abs().max() < 1e-3

    res, errs = {}, []

    def worker(name, exact):
        try:
            torch.cuda.set_device(dev)
            ctx = ctypes.c_void_p()
            _lib.check(L.d3d_ctx_create(ctypes.byref(ctx)), "create")
            _lib.check(L.d3d_ctx_bind(ctx), "bind")
            if exact:
                _lib.check(L.d3d_set_tuning(10, 1), "knob")
            res[name] = _forward(L, rb, f, w, dev, torch.cuda.Stream(dev), reps=40)
            _lib.check(L.d3d_ctx_destroy(ctx), "destroy")
        except Exception as e:  # noqa: BLE001
            errs.append((name, repr(e)))

    ts = [threading.Thread(target=worker, args=("split", False)), threading.Thread(target=worker, args=("exact", True))]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    assert not errs, errs
    v = ctypes.c_int(-7)
    assert L.d3d_ctx_get_tuning(10, ctypes.byref(v)) == 0 and v.value == 0   # this thread's (default) context: untouched
    for o, st in res["split"]:
        assert torch.equal(o, ref_split[0]) and torch.equal(st, ref_split[1])
    for o, st in res["exact"]:
        assert torch.equal(o, ref_exact[0])
