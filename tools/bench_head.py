"""d3d_head_conv_fwd / _dgrad / _wgrad alone on the device (HIP events), f32 map of the KITTI head
[4, 200, 176, 128] -> 14 + 2 + 4 channels.  Run twice to compare the f32-MFMA kernels (default) with the round-3 VALU
kernels:  python tools/bench_head.py ; D3D_HEAD_VALU=1 python tools/bench_head.py
Also checks the three results against float64 matmuls."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from det3d_amd import _lib  # noqa: E402
from det3d_amd.bbox_heads import _ptr_array  # noqa: E402


def main():
    L = _lib.lib()
    dev = torch.device("cuda")
    torch.manual_seed(0)
    b, h, w, cin = 4, 200, 176, 128
    n = [14, 2, 4]
    m = b * h * w
    x = torch.randn(m, cin, device=dev)
    ws_ = [torch.randn(k, cin, device=dev) * 0.05 for k in n]
    bs = [torch.randn(k, device=dev) for k in n]
    outs = [torch.empty(m, k, device=dev) for k in n]
    gs = [torch.randn(m, k, device=dev) for k in n]
    dx = torch.empty(m, cin, device=dev)
    nws = L.d3d_head_conv_wgrad_ws_floats(m)
    wsb = torch.empty(nws, device=dev)
    st = _lib.stream_ptr(dev)
    narr = _lib.i32_array(n)
    calls = {
        "fwd": lambda: L.d3d_head_conv_fwd(x.data_ptr(), m, cin, _lib.D3D_F32, _ptr_array(ws_), _ptr_array(bs), _ptr_array(outs),
                                           narr, 3, st),
        "dgrad": lambda: L.d3d_head_conv_dgrad(_ptr_array(gs), narr, 3, m, cin, _lib.D3D_F32, _ptr_array(ws_), dx.data_ptr(), st),
        "wgrad": lambda: L.d3d_head_conv_wgrad(x.data_ptr(), _ptr_array(gs), narr, 3, m, cin, _lib.D3D_F32, wsb.data_ptr(), nws, st),
    }
    mode = "VALU (D3D_HEAD_VALU=1)" if os.environ.get("D3D_HEAD_VALU", "0") == "1" else "f32 MFMA"
    for name, fn in calls.items():
        for _ in range(3):
            _lib.check(fn(), name)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = 20
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        us = 1e3 * e0.elapsed_time(e1) / reps
        mb = (m * cin * 4 + sum(n) * m * 4) / 1e6
        label = "VALU (only kernel)" if name == "dgrad" else mode
        print("%-6s %-24s %7.1f us  (%.0f MB -> %.2f TB/s)" % (name, label, us, mb, mb / us))
    # parity against float64
    xd = x.double()
    for k in range(3):
        ref = xd @ ws_[k].double().t() + bs[k].double()
        err = float((outs[k].double() - ref).abs().max()) / float(ref.abs().max())
        print("fwd conv %d: max rel err %.2e" % (k, err))
    ref = sum(gs[k].double() @ ws_[k].double() for k in range(3))
    print("dgrad: max rel err %.2e" % (float((dx.double() - ref).abs().max()) / float(ref.abs().max())))
    first = 0
    for k in range(3):
        ref = gs[k].double().t() @ xd
        got = wsb[: 32 * cin].view(32, cin)[first: first + n[k]].double()
        refb = gs[k].double().sum(0)
        gotb = wsb[32 * cin: 32 * cin + 32][first: first + n[k]].double()
        print("wgrad conv %d: dW max rel err %.2e, db %.2e" % (k, float((got - ref).abs().max()) / float(ref.abs().max()),
                                                              float((gotb - refb).abs().max()) / float(refb.abs().max())))
        first += n[k]
    assert float(wsb[: 32 * cin].view(32, cin)[first:].abs().max()) == 0.0, "padded rows of dW must be zero"


if __name__ == "__main__":
    main()
