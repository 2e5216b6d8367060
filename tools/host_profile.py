"""host-side cProfile of the training step (run on the GPU box)."""
import cProfile, io, os, pstats, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from det3d_amd import Config, build_detector, synthetic
from det3d_amd.trainer import TrainStep
dev = torch.device("cuda", 0)
cfg = Config.fromfile(os.path.join(ROOT, "det3d_amd", "configs", "kitti_car_second.py"))
torch.manual_seed(0)
model = build_detector(cfg.model, cfg.train_cfg, cfg.test_cfg).to(dev).train()
ds = synthetic.KittiCarSynthetic(n_points=20000, n_seeds=8)
batches = [ds.device_batch(i, 4, dev) for i in range(2)]
step = TrainStep(model, total_steps=100, autocast_dtype=torch.bfloat16)
def one(i):
    return step(synthetic.make_example(batches[i % 2], synthetic.KITTI_CAR))
for i in range(4):
    one(i)
torch.cuda.synchronize()
for trial in range(2):
    t0 = time.perf_counter()
    for i in range(10):
        one(i)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print("10 steps: host issue %.1f ms/step, +sync %.1f ms/step" % ((t1 - t0) * 100, (t2 - t0) * 100))
pr = cProfile.Profile()
pr.enable()
for i in range(10):
    one(i)
torch.cuda.synchronize()
pr.disable()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(35)
print(s.getvalue()[:6000])
