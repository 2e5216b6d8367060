"""host cost of one C-ABI call from Python: a no-op entry point, a launching one, and a torch op for comparison"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from det3d_amd import _lib
L = _lib.lib()
dev = torch.device("cuda", 0)
x = torch.zeros(64 * 1024, device=dev)
st = _lib.stream_ptr(dev)
def bench(name, fn, n=3000):
    for _ in range(50): fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n): fn()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print("%-46s %6.2f us/call enqueue, %6.2f us/call incl. drain" % (name, (t1 - t0) / n * 1e6, (t2 - t0) / n * 1e6))
bench("ctypes no-op (d3d_set_tuning)", lambda: L.d3d_set_tuning(2, 0))
stats = torch.zeros(2 * 8 * 64, device=dev)
xb = torch.randn(4096, 64, device=dev).bfloat16()
bench("ctypes + 2 launches (d3d_bn_stats, 4096x64)", lambda: L.d3d_bn_stats(xb.data_ptr(), 4096, 64, _lib.D3D_BF16, stats.data_ptr(), st))
bench("torch op (x.add_(1), 64k floats)", lambda: x.add_(1.0))
bench("torch.empty((4096, 64))", lambda: torch.empty((4096, 64), device=dev))
bench("_lib.stream_ptr", lambda: _lib.stream_ptr(dev))
