"""dense conv kernels (csrc/d3d_conv2d.hip) against MIOpen at the RPN's shapes: us per launch, TFLOP/s.
    python tools/bench_conv2d.py [--shape B H W CIN COUT K S P]
    env: VARIANTS=9,16,17 (d3d_set_tuning knob 7), WGRAD_WAVES=8,4, MODES=bf16,split, NO_MIOPEN=1"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from det3d_amd import dense_conv as dc  # noqa: E402


def timeit(fn, reps=20, warm=4):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps


def main():
    shapes = [(4, 200, 176, 128, 128, 3, 1, 1), (4, 200, 176, 128, 128, 1, 1, 0), (4, 128, 128, 256, 256, 3, 1, 1),
              (4, 128, 128, 256, 128, 3, 1, 1)]
    if "--shape" in sys.argv:
        i = sys.argv.index("--shape")
        shapes = [tuple(int(v) for v in sys.argv[i + 1: i + 9])]
    out = []
    for b, h, w, cin, cout, k, s, p in shapes:
        x = torch.randn((b, h, w, cin), device="cuda")
        wt = torch.nn.Parameter(torch.randn((cout, cin, k, k), device="cuda") * 0.03)
        ho, wo = dc._out_hw(h, w, k, s, p)
        gy = torch.randn((b, ho, wo, cout), device="cuda")
        flop = 2.0 * b * ho * wo * cout * cin * k * k
        row = {"shape": [b, h, w, cin, cout, k, s, p], "gflop": flop / 1e9}
        from det3d_amd import _lib
        variants = [int(v) for v in os.environ.get("VARIANTS", "8").split(",")]
        for mode in os.environ.get("MODES", "bf16,split").split(","):
            w_hi, w_lo = dc._packed(wt, "fwd", mode)
            if mode == "bf16":
                x_hi, x_lo, g_hi, g_lo = x.to(torch.bfloat16), None, gy.to(torch.bfloat16), None
            else:
                (x_hi, x_lo), (g_hi, g_lo) = dc.split_planes(x), dc.split_planes(gy)
            for v in variants:
                for tail in [int(t) for t in os.environ.get("TAILS", "1").split(",")]:
                    _lib.lib().d3d_set_tuning(7, v)
                    _lib.lib().d3d_set_tuning(9, tail)
                    t = timeit(lambda: dc.conv2d_raw(x_hi, x_lo, w_hi, w_lo, k, s, p, mode))
                    ts = timeit(lambda: dc.conv2d_raw(x_hi, x_lo, w_hi, w_lo, k, s, p, mode, want_stats=True))
                    tag = mode + "_v%d" % v + ("" if tail else "_whole")
                    row[tag + "_fwd_us"] = round(t, 1)
                    row[tag + "_fwd_stats_us"] = round(ts, 1)
                    row[tag + "_fwd_TF"] = round(flop / t / 1e6, 1)
            _lib.lib().d3d_set_tuning(9, 1)
            _lib.lib().d3d_set_tuning(7, variants[0])
            if cin % 128 == 0:
                for wv in [int(v) for v in os.environ.get("WGRAD_WAVES", "8").split(",")]:
                    if wv in (16, 17) and not (mode == "split" and k == 3 and s == 1):
                        continue
                    _lib.lib().d3d_set_tuning(8, wv)
                    tw = timeit(lambda: dc.wgrad_raw(g_hi, g_lo, x_hi, x_lo, k, s, p, mode))
                    row[mode + "_wgrad%d_us" % wv] = round(tw, 1)
                    row[mode + "_wgrad%d_TF" % wv] = round(flop / tw / 1e6, 1)
                _lib.lib().d3d_set_tuning(8, 16)
        # MIOpen through torch, channels-last
        for dt, nm in (() if os.environ.get("NO_MIOPEN") else ((torch.bfloat16, "miopen_bf16"), (torch.float32, "miopen_f32"))):
            xm = x.permute(0, 3, 1, 2).to(dt).contiguous(memory_format=torch.channels_last).requires_grad_(True)
            wm = wt.detach().to(dt).contiguous(memory_format=torch.channels_last).requires_grad_(True)
            gm = gy.permute(0, 3, 1, 2).to(dt).contiguous(memory_format=torch.channels_last)
            row[nm + "_fwd_us"] = round(timeit(lambda: torch.nn.functional.conv2d(xm, wm, None, s, p)), 1)
            ym = torch.nn.functional.conv2d(xm, wm, None, s, p)
            row[nm + "_bwd_us"] = round(timeit(lambda: torch.autograd.grad(ym, (xm, wm), gm, retain_graph=True)), 1)
        print(json.dumps(row), flush=True)
        out.append(row)


if __name__ == "__main__":
    main()
