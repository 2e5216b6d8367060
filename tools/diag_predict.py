"""fused vs per-frame predict() on the bench model: detection counts and box agreement (diagnostic)"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from det3d_amd import Config, bbox_heads, build_detector, synthetic  # noqa: E402

dev = torch.device("cuda", 0)
cfg = Config.fromfile(os.path.join(ROOT, "det3d_amd", "configs", "kitti_car_second.py"))
torch.manual_seed(0)
model = build_detector(cfg.model, cfg.train_cfg, cfg.test_cfg).to(dev).eval()
ds = synthetic.KittiCarSynthetic(n_points=20000, n_seeds=4)
ex = synthetic.make_example(ds.device_batch(0, 4, dev), synthetic.KITTI_CAR)
for dt in (None, torch.bfloat16):
    with torch.no_grad(), torch.autocast("cuda", dtype=dt, enabled=dt is not None):
        data = dict(features=ex["voxel_mean"], num_voxels=ex["num_points"], coors=ex["coordinates"], batch_size=4,
                    input_shape=ex["shape"][0])
        preds = model.bbox_head(model.extract_feat(data))
        cls = preds[0]["cls_preds"].float()
        print("dtype", dt, "cls logits: min %.4f max %.4f, distinct values %d of %d" % (
            float(cls.min()), float(cls.max()), int(torch.unique(cls).numel()), cls.numel()))
        out = {}
        for fused in (True, False):
            bbox_heads.FUSE_PREDICT = fused
            dets = model.bbox_head.predict(ex, preds, model.test_cfg)
            out[fused] = dets
            print("  fused" if fused else "  per-frame", [int(d["scores"].shape[0]) for d in dets],
                  "top scores", [round(float(d["scores"][0]), 4) if len(d["scores"]) else None for d in dets])
        bbox_heads.FUSE_PREDICT = True
