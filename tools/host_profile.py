"""host-side view of the training step (run on the GPU box): issue time vs. synchronised time per step and a
cProfile of 20 steps, with the same settings as bench.py (NHWC, MIOpen find mode, device AssignTarget).
usage: python tools/host_profile.py [repo_root] [fp32|bf16]   (repo_root: another checkout of this repo to profile)"""
import cProfile, io, os, pstats, sys, time
os.environ.setdefault("PYTORCH_MIOPEN_SUGGEST_NHWC", "1")
os.environ.setdefault("PYTORCH_MIOPEN_SUGGEST_NHWC_BATCHNORM", "1")
import torch
DTYPE = sys.argv[2] if len(sys.argv) > 2 else "bf16"
ROOT = os.path.abspath(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1] != "." else os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from det3d_amd import Config, build_detector, synthetic
from det3d_amd.trainer import TrainStep
print("profiling", ROOT)
dev = torch.device("cuda", 0)
torch.backends.cudnn.benchmark = True
cfg = Config.fromfile(os.path.join(ROOT, "det3d_amd", "configs", "kitti_car_second.py"))
torch.manual_seed(0)
model = build_detector(cfg.model, cfg.train_cfg, cfg.test_cfg).to(dev).train()
model.backbone.dense_channels_last = True
ds = synthetic.KittiCarSynthetic(n_points=20000, n_seeds=8)
batches = [ds.device_batch(i, 4, dev) for i in range(2)]
step = TrainStep(model, total_steps=100, autocast_dtype=torch.bfloat16 if DTYPE == "bf16" else None)
def one(i):
    return step(synthetic.make_example(batches[i % 2], synthetic.KITTI_CAR, device_targets=True))
for i in range(8):
    one(i)
torch.cuda.synchronize()
for trial in range(3):
    t0 = time.perf_counter()
    for i in range(20):
        one(i)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print("20 steps: host issue %.2f ms/step, +final sync %.2f ms/step" % ((t1 - t0) * 50, (t2 - t0) * 50))
# the backward functions run on the autograd engine's device thread, where cProfile does not look: one thread for this pass
torch.autograd.set_multithreading_enabled(False)
pr = cProfile.Profile()
pr.enable()
for i in range(20):
    one(i)
torch.cuda.synchronize()
pr.disable()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(45)
print(s.getvalue()[:12000])
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("cumtime").print_stats(45)
print(s.getvalue()[:12000])
