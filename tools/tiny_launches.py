"""Which Python line launches each of the step's small kernels?  torch.profiler with stacks over a few training steps
(same settings as tools/host_profile.py); prints, per kernel that is not one of the library's big ones, the innermost
det3d_amd / bench frames of the launching op.   usage: python tools/tiny_launches.py [fp32|bf16]"""
import collections
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from det3d_amd import Config, build_detector, synthetic  # noqa: E402
from det3d_amd.trainer import TrainStep  # noqa: E402

DTYPE = sys.argv[1] if len(sys.argv) > 1 else "fp32"
dev = torch.device("cuda", 0)
cfg = Config.fromfile(os.path.join(ROOT, "det3d_amd", "configs", "kitti_car_second.py"))
torch.manual_seed(0)
model = build_detector(cfg.model, cfg.train_cfg, cfg.test_cfg).to(dev).train()
model.backbone.dense_channels_last = True
ds = synthetic.KittiCarSynthetic(n_points=20000, n_seeds=8)
batches = [ds.device_batch(i, 4, dev) for i in range(2)]
step = TrainStep(model, total_steps=100, autocast_dtype=torch.bfloat16 if DTYPE == "bf16" else None)


def one(i):
    return step(synthetic.make_example(batches[i % 2], synthetic.KITTI_CAR, device_targets=True))


for i in range(6):
    one(i)
torch.cuda.synchronize()
torch.autograd.set_multithreading_enabled(False)
STEPS = 4
with torch.profiler.profile(activities=[torch.profiler.ProfilerActivity.CPU, torch.profiler.ProfilerActivity.CUDA],
                            with_stack=True) as prof:
    for i in range(STEPS):
        one(i)
    torch.cuda.synchronize()

BIG = ("spconv_", "conv2d_halo", "conv2d_wgrad", "conv2d_nhwc", "bn_", "wgrad_reduce", "rc_", "rb_", "head_conv", "vox_",
       "d3d_scan", "asg_", "adam_", "pack_all", "loss_", "dense_move", "conv2d_split")
events = prof.events()
rows = collections.Counter()
dur = collections.Counter()
for c in events:
    if c.device_type != torch.autograd.DeviceType.CPU or not c.kernels:
        continue
    for k in c.kernels:
        if any(b in k.name for b in BIG):
            continue
        frames = [f for f in (c.stack or []) if "det3d_amd" in f or "tools/" in f or "bench.py" in f]
        where = " <- ".join(f.split("/")[-1].strip() for f in frames[:3]) or ((c.stack or ["?"])[0])
        key = (k.name[:70], "%s @ %s" % (c.name, where))
        rows[key] += 1
        dur[key] += k.duration
print("%d steps; launches per step | total us per step | kernel | launched from" % STEPS)
for k, n in sorted(rows.items(), key=lambda kv: -dur[kv[0]]):
    print("%5.1f  %7.1f  %-70s  %s" % (n / STEPS, dur[k] / STEPS, k[0], k[1]))
