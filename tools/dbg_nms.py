import os, sys, numpy as np, torch
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
from det3d_amd import Config, build_detector, synthetic, box_ops
from oracle import model_ref as M, clib
ROOT='/root/repo'
cfg = Config.fromfile(os.path.join(ROOT, "det3d_amd", "configs", "kitti_car_second.py"))
torch.manual_seed(1)
prod = build_detector(cfg.model, cfg.train_cfg, cfg.test_cfg)
for m in prod.modules():
    if isinstance(m, (torch.nn.BatchNorm1d, torch.nn.BatchNorm2d)):
        m.running_mean.normal_(0, 0.1); m.running_var.uniform_(0.5, 1.5)
torch.nn.init.constant_(prod.bbox_head.tasks[0].conv_cls.bias, 0.5)
prod = prod.cuda().eval()
ds = synthetic.KittiCarSynthetic(n_points=6000, n_seeds=2)
ex = synthetic.make_example(ds.device_batch(0, 2, torch.device("cuda")), synthetic.KITTI_CAR)
with torch.no_grad():
    data = dict(features=ex["voxel_mean"], num_voxels=ex["num_points"], coors=ex["coordinates"], batch_size=2, input_shape=ex["shape"][0])
    preds = prod.bbox_head(prod.extract_feat(data))
    p = preds[0]
    an = ex["anchors"][0][0].view(-1,7)
    box = box_ops.second_box_decode(p["box_preds"][0].reshape(-1,7).float(), an)
    sc = torch.sigmoid(p["cls_preds"][0].reshape(-1).float())
    keep = sc >= 0.3
    box, sc = box[keep], sc[keep]
    print("candidates", len(sc), "unique scores", len(torch.unique(sc)))
    k = min(len(sc), 1000)
    s_k, idx = torch.topk(sc, k)
    b5 = box[idx][:, [0,1,3,4,6]].contiguous()
    d6 = torch.cat([b5, s_k[:,None]],1).cpu().numpy()
    ref, margin = clib.rotate_nms_cc(d6, 0.01, return_margin=True)
    g = box_ops._device_nms("d3d_rotate_nms", b5, 0.01, 100).cpu().tolist()
    print("oracle keep", len(ref), ref[:20], "margin", margin)
    print("gpu keep", len(g), g[:20])
    print("score sorted desc?", bool((s_k[:-1] >= s_k[1:]).all()), "ties", int((s_k[:-1] == s_k[1:]).sum()))
    # pairwise iou compare
    iou_g = box_ops.rotate_iou(b5[:200], b5[:200]).cpu().numpy()
    iou_o = clib.rotate_iou(d6[:200,:5], d6[:200,:5])
    print("max iou diff", np.abs(iou_g-iou_o).max(), "at", np.unravel_index(np.abs(iou_g-iou_o).argmax(), iou_g.shape))
    i,j = np.unravel_index(np.abs(iou_g-iou_o).argmax(), iou_g.shape)
    print(d6[i], d6[j], iou_g[i,j], iou_o[i,j])
