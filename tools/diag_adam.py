"""fused optimizer step vs the PyTorch sequence on the real detector: parameters after 1 and 2 steps, per tensor.
   python tools/diag_adam.py [fp32|bf16]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from det3d_amd import Config, build_detector, synthetic  # noqa: E402
from det3d_amd.trainer import TrainStep  # noqa: E402

dtype = sys.argv[1] if len(sys.argv) > 1 else "fp32"
dev = torch.device("cuda", 0)
cfg = Config.fromfile(os.path.join(ROOT, "det3d_amd", "configs", "kitti_car_second.py"))
ds = synthetic.KittiCarSynthetic(n_points=6000, n_seeds=2)
ex = synthetic.make_example(ds.device_batch(0, 1, dev), synthetic.KITTI_CAR)
runs = {}
for fused in ("1", "0"):
    os.environ["D3D_FUSED_ADAM"] = fused
    torch.manual_seed(0)
    model = build_detector(cfg.model, cfg.train_cfg, cfg.test_cfg).to(dev).train()
    step = TrainStep(model, total_steps=10, autocast_dtype=torch.bfloat16 if dtype == "bf16" else None)
    snaps, losses, norms = [], [], []
    for i in range(3):
        losses.append(float(step(ex)))
        norms.append(float(step.last["grad_norm"]))
        snaps.append([p.detach().clone() for p in model.parameters()])
    runs[fused] = (snaps, losses, norms, [n for n, _ in model.named_parameters()])
    print("fused=%s losses %s grad norms %s" % (fused, losses, norms))
a, b = runs["1"], runs["0"]
for s in range(3):
    worst = sorted(((float((x - y).abs().max()), float(y.abs().max()), n) for x, y, n in zip(a[0][s], b[0][s], a[3])), reverse=True)[:4]
    print("after step %d: largest |fused - torch| per tensor:" % (s + 1), worst)
