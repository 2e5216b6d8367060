"""the body of tests/test_optim_gpu.py::test_training_steps_with_the_fused_optimizer_follow_the_torch_sequence, printing
losses and gradient norms of both optimizer paths (usage: python tools/optim_follow.py [repo_root] [steps])"""
import os
import sys

ROOT = os.path.abspath(sys.argv[1]) if len(sys.argv) > 1 else os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from det3d_amd import Config, build_detector, synthetic  # noqa: E402
from det3d_amd.trainer import TrainStep  # noqa: E402

steps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
cfg = Config.fromfile(os.path.join(ROOT, "det3d_amd", "configs", "kitti_car_second.py"))
dev = torch.device("cuda", 0)
ds = synthetic.KittiCarSynthetic(n_points=6000, n_seeds=2)
ex = synthetic.make_example(ds.device_batch(0, 1, dev), synthetic.KITTI_CAR)
for fused in ("1", "0", "1"):
    os.environ["D3D_FUSED_ADAM"] = fused
    torch.manual_seed(0)
    model = build_detector(cfg.model, cfg.train_cfg, cfg.test_cfg).to(dev).train()
    step = TrainStep(model, total_steps=10, autocast_dtype=None)
    losses, norms = [], []
    for i in range(steps):
        losses.append(float(step(ex)))
        norms.append(float(step.last["grad_norm"]))
    print(ROOT, "fused" if fused == "1" else "torch", ["%.6f" % v for v in losses], ["%.4f" % v for v in norms])
