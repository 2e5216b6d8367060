"""Micro-benchmark of the four BatchNorm sweeps on the RPN's channels-last maps ([140800, 128] and [140800, 256]
bf16, 36 / 72 MB) and on a sparse-layer sized input: every reduction variant of d3d_bn.hip (d3d_set_tuning knob 3),
the dx kernel's rows in flight (knob 5), the streaming grid cap (knob 6), and the cost of the global atomics (knob 4).
Buffers rotate through > 256 MB so that nothing is served from the Infinity Cache.

    python tools/bench_bn.py [out.json]
"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from det3d_amd import _lib  # noqa: E402

L = _lib.lib()
dev = torch.device("cuda", 0)
ST = _lib.stream_ptr(dev)
COP = _lib.D3D_BN_COPIES


def sets(n, c, dtype, k):
    g = torch.Generator(device=dev).manual_seed(1)
    out = []
    for _ in range(k):
        x = (torch.randn((n, c), device=dev, generator=g) * 1.5 + 0.3).to(dtype)
        dy = torch.randn((n, c), device=dev, generator=g).to(dtype)
        out.append((x, dy, torch.empty_like(x)))
    return out


def timeit(fn, bufs, reps=24):
    for i in range(4):
        fn(*bufs[i % len(bufs)])
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(reps):
        fn(*bufs[i % len(bufs)])
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps  # us per call (includes the accumulator fill launch where there is one)


def main():
    rows = []
    for n, c, dtype in [(140800, 128, torch.bfloat16), (140800, 256, torch.bfloat16), (140800, 128, torch.float32),
                        (60000, 64, torch.bfloat16)]:
        dt = _lib.D3D_BF16 if dtype == torch.bfloat16 else _lib.D3D_F32
        esz = 2 if dtype == torch.bfloat16 else 4
        mb = n * c * esz / 1e6
        k = max(3, int(900 / (3 * mb)) + 1)
        bufs = sets(n, c, dtype, k)
        gamma = torch.linspace(0.5, 1.5, c, device=dev)
        beta = torch.linspace(-0.5, 0.5, c, device=dev)
        # sized for the largest grid any variant uses (512 blocks, 32 groups)
        stats = torch.empty((2 * c * (1 + 512 + 32),), device=dev)
        part = torch.empty((2 * c * (1 + 512 + 32),), device=dev)
        save = torch.empty((2, c), device=dev)
        rm, rv = torch.zeros(c, device=dev), torch.ones(c, device=dev)
        g2 = torch.empty((2, c), device=dev)

        def f_stats(x, dy, o):
            _lib.check(L.d3d_bn_stats(x.data_ptr(), n, c, dt, stats.data_ptr(), ST), "stats")

        def f_apply(x, dy, o):
            _lib.check(L.d3d_bn_apply(x.data_ptr(), n, c, dt, gamma.data_ptr(), beta.data_ptr(), rm.data_ptr(),
                                      rv.data_ptr(), 0.01, 1e-3, 1, 1, stats.data_ptr(), COP, o.data_ptr(),
                                      save[0].data_ptr(), save[1].data_ptr(), ST), "apply")

        def f_red(x, dy, o):
            _lib.check(L.d3d_bn_bwd_reduce(x.data_ptr(), dy.data_ptr(), n, c, dt, gamma.data_ptr(), beta.data_ptr(),
                                           save[0].data_ptr(), save[1].data_ptr(), 1, part.data_ptr(), ST), "red")

        def f_dx(x, dy, o):
            _lib.check(L.d3d_bn_bwd_dx(x.data_ptr(), dy.data_ptr(), n, c, dt, gamma.data_ptr(), beta.data_ptr(),
                                       save[0].data_ptr(), save[1].data_ptr(), 1, part.data_ptr(), COP, o.data_ptr(),
                                       g2[0].data_ptr(), g2[1].data_ptr(), ST), "dx")

        # reference sums for the correctness check of every variant
        x0, dy0, o0 = bufs[0]
        ref_s = torch.cat([x0.float().sum(0), (x0.float() ** 2).sum(0)])
        big = n * c >= (1 << 23)
        for var in (range(10) if big else [0]):
            L.d3d_set_tuning(3, var)
            f_stats(x0, dy0, o0)
            got = stats[:2 * c]
            err = float((got - ref_s).abs().max() / ref_s.abs().max())
            f_apply(x0, dy0, o0)
            t_s = timeit(f_stats, bufs)
            t_r = timeit(f_red, bufs)
            L.d3d_set_tuning(4, 1)
            t_s0 = timeit(f_stats, bufs)
            t_r0 = timeit(f_red, bufs)
            L.d3d_set_tuning(4, 0)
            rows.append(dict(n=n, c=c, dtype=str(dtype)[6:], MB=round(mb, 1), variant=var, stats_us=round(t_s, 1),
                             stats_GBs=round(mb / t_s * 1e3, 0), reduce_us=round(t_r, 1),
                             reduce_GBs=round(2 * mb / t_r * 1e3, 0), stats_us_no_global_atomics=round(t_s0, 1),
                             reduce_us_no_global_atomics=round(t_r0, 1), stats_rel_err=err))
            print(rows[-1], flush=True)
        L.d3d_set_tuning(3, 0)
        f_stats(x0, dy0, o0)
        f_apply(x0, dy0, o0)
        f_red(x0, dy0, o0)
        for cap in (0, 512, 1024, 4096):
            L.d3d_set_tuning(6, cap)
            t_a = timeit(f_apply, bufs)
            row = dict(n=n, c=c, dtype=str(dtype)[6:], MB=round(mb, 1), stream_cap=cap, apply_us=round(t_a, 1),
                       apply_GBs=round(2 * mb / t_a * 1e3, 0))
            for ur in (1, 2, 4):
                L.d3d_set_tuning(5, ur)
                t_d = timeit(f_dx, bufs)
                row["dx_us_rows%d" % ur] = round(t_d, 1)
                row["dx_GBs_rows%d" % ur] = round(3 * mb / t_d * 1e3, 0)
            rows.append(row)
            print(row, flush=True)
        L.d3d_set_tuning(6, 0)
        L.d3d_set_tuning(5, 2)
        del bufs
        torch.cuda.empty_cache()
    if len(sys.argv) > 1:
        with open(sys.argv[1], "w") as f:
            json.dump(rows, f, indent=1)


if __name__ == "__main__":
    main()
