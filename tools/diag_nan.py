"""train the bench workload step by step and report the first non-finite loss / gradient / buffer (diagnostic)"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402

sys.argv = ["bench.py", "--steps", "400", "--warmup", "0", "--no-cpu-baseline", "--no-extras"]
args = bench.parse()
dev = torch.device("cuda", 0)
torch.backends.cudnn.benchmark = os.environ.get("D3D_MIOPEN_FIND", "1") == "1"
STEPS = int(os.environ.get("DIAG_STEPS", "400"))
wl = bench.Workload("c2-20k", args, dev, 0, 1, "bf16", STEPS)
model = wl.model
for i in range(STEPS):
    ex = wl.example(i)
    loss = wl.step_fn(ex)
    lv = float(loss)
    last = wl.step_fn.last
    bad = not (lv == lv and abs(lv) < 1e30)
    if i % 20 == 0 or bad:
        print("step %d loss %.4f gnorm %.4f lr %.2e" % (i, lv, float(last["grad_norm"]), last["lr"]), flush=True)
    if bad:
        for k, p in model.named_parameters():
            if p.grad is not None and not torch.isfinite(p.grad).all():
                print("  non-finite grad:", k, int((~torch.isfinite(p.grad)).sum()), "of", p.grad.numel())
            if not torch.isfinite(p).all():
                print("  non-finite param:", k)
        for k, b in model.named_buffers():
            if b.is_floating_point() and not torch.isfinite(b).all():
                print("  non-finite buffer:", k)
        print("  losses", {k: [float(x) for x in v] if isinstance(v, list) and v and torch.is_tensor(v[0]) and v[0].numel() == 1 else None
                           for k, v in last["losses"].items()})
        break
print("done")
