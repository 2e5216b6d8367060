"""Regenerate tests/golden/*.npz|*.pt by RUNNING THE REFERENCE'S OWN PYTHON CODE
(/root/reference/det3d, loaded through ref_loader's stubs) on small seeded inputs.

    python tests/golden/make_golden.py

Only works in the build container (needs /root/reference).  The produced files are
committed; tests never import the reference.  What each file pins:

  voxel_cases.npz   det3d/ops/point_cloud/point_cloud_ops.py:7-55,112-184 run as plain
                    Python (numba.jit stubbed to identity) -> voxels/coors/num_points
  rotate_iou.npz    det3d/ops/nms/nms_gpu.py:180-416 (devRotateIoU, the numba-CUDA
                    rotated IoU) run as plain Python -> IoU of box pairs
  rotate_iou_eval.npz  datasets/utils/kitti_object_eval_python/rotate_iou.py:246-258
                    (devRotateIoUEval) run as plain Python, criteria -1 / 0 / 1 / 2
  task_targets.npz  TargetAssigner.assign_v2 (core/anchor/target_assigner.py:68-142) on two-class task
                    groups: nuScenes 9-dim boxes + angle-vector coder, Lyft 7-dim boxes
  augment.npz       random_flip(_both) + global_rotation + global_scaling_v2 (core/sampler/preprocess.py:796-861)
                    with a seeded NumPy stream, 7-dim (KITTI) and 9-dim (nuScenes) boxes
  assign_pipeline.npz  the AssignTarget pipeline step (datasets/pipelines/preprocess.py:306-483) built from the
                    nuScenes-CBGS and Lyft assigner configs: per-task labels / positive targets / anchor checksums
  range_filter.npz  filter_gt_box_outside_range (core/sampler/preprocess.py:133-147)
  collate.pt        collate_kitti (torchie/parallel/collate.py:90-150) on ragged frames, every key class
  sampler.npz       DistributedGroupSampler (datasets/loader/sampler.py:138-224): per-rank indices, two epochs
  vfe_pfn.pt        VoxelFeatureExtractorV3 (voxel_encoder.py:197-211), PillarFeatureNet +
                    PointPillarsScatter (pillar_encoder.py:17-211)
  rpn.pt            RPN (necks/rpn.py:22-159) in three stride configurations
  head.pt           MultiGroupHead forward / loss / predict (mg_head.py:385-1085), losses
                    (losses.py), second_box_decode (box_torch_ops.py:80-148).  predict()'s
                    rotate_nms_cc needs the boost extension, which cannot be built here:
                    it is replaced by this repo's oracle NMS, so head.pt pins everything
                    in predict() EXCEPT the polygon step.
  anchors_targets.npz  AnchorGeneratorRange + create_target_np (core/anchor/*)
  detectors.pt      the reference's WHOLE detectors -- VoxelNet (detectors/voxelnet.py) / PointPillars
                    (detectors/point_pillars.py) built by the reference's own registries from the four
                    shipped model configs (KITTI SECOND, nuScenes CBGS, Lyft CBGS, nuScenes PointPillars):
                    readers, scn.py (SpMiddleFHD / SpMiddleResNetFHD + SparseBasicBlock), PointPillarsScatter,
                    RPN, MultiGroupHead forward / loss / predict -- on cropped ranges (detector_specs.py).
                    scn.py runs over oracle/spconv_shim.py (spconv is un-vendored; its arithmetic is the
                    oracle's, pinned to F.conv3d) and rotate_nms_cc is the oracle NMS (boost absent).
  nms_gpu.npz       det3d/ops/nms/nms_gpu.py:22-126 (iou_device, nms_kernel, nms_postprocess: the +1-pixel
                    axis-aligned NMS) run as plain Python through the numba stub
"""
import collections
import collections.abc
import logging
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

import ref_loader  # noqa: E402

for _n in ("Iterable", "Mapping", "Sequence"):
    if not hasattr(collections, _n):
        setattr(collections, _n, getattr(collections.abc, _n))

KITTI_RANGE = [0, -40.0, -3.0, 70.4, 40.0, 1.0]
KITTI_VS = [0.05, 0.05, 0.1]
PP_RANGE = [0, -39.68, -3, 69.12, 39.68, 1]
PP_VS = [0.16, 0.16, 4]


def voxel_cases():
    pc = ref_loader.ref("det3d.ops.point_cloud.point_cloud_ops")
    rng = np.random.default_rng(1234)
    out = {}

    def run(name, pts, vs, rg, T, M):
        v, c, n = pc.points_to_voxel(pts, vs, rg, T, True, M)
        out[name + "_points"] = pts
        out[name + "_cfg"] = np.array(list(vs) + list(rg) + [T, M], dtype=np.float64)
        out[name + "_voxels"] = v
        out[name + "_coors"] = c
        out[name + "_num"] = n

    # (i) sparse random cloud, no cap, incl. out-of-range points
    pts = rng.uniform([-2, -42, -3.5, 0], [72, 42, 1.5, 1], size=(1200, 4)).astype(np.float32)
    run("sparse", pts, KITTI_VS, KITTI_RANGE, 5, 20000)
    # (ii) max_voxels cap + early break
    run("capped", pts, KITTI_VS, KITTI_RANGE, 5, 150)
    # (iii) clustered: many points per voxel (> T), points exactly on cell/range boundaries
    centers = rng.uniform([5, -10, -2, 0], [30, 10, 0, 1], size=(40, 4))
    cl = (centers[rng.integers(0, 40, 1500)] + rng.normal(0, 0.03, (1500, 4))).astype(np.float32)
    edge = np.array([[0, -40, -3, .5], [70.4, 40, 1, .5], [70.39999, 39.99999, 0.99999, .5],
                     [0.05, -39.95, -2.9, .1], [0.1, 0.0, 0.0, .2], [35.2, 0.05, -1.0, .3],
                     [-0.0, -0.0, -0.0, .4]], dtype=np.float32)
    run("clustered", np.concatenate([cl, edge]).astype(np.float32), KITTI_VS, KITTI_RANGE, 3, 20000)
    # (iv) pillars (z collapses to one cell), T large, 5-channel points
    p5 = rng.uniform([0, -39.68, -3, 0, 0], [69.12, 39.68, 1, 1, 0.5], size=(900, 5)).astype(np.float32)
    p5[:300, :2] = p5[0, :2] + rng.normal(0, 0.05, (300, 2)).astype(np.float32)
    run("pillars", p5, PP_VS, PP_RANGE, 100, 12000)
    np.savez_compressed(os.path.join(HERE, "voxel_cases.npz"), **out)
    print("voxel_cases:", {k: v.shape for k, v in out.items() if k.endswith("_voxels")})


def rotate_iou_cases():
    ng = ref_loader.ref("det3d.ops.nms.nms_gpu")
    rng = np.random.default_rng(7)
    n = 160
    a = np.stack([rng.uniform(0, 6, n), rng.uniform(0, 6, n), rng.uniform(1, 3, n),
                  rng.uniform(2, 5, n), rng.uniform(-3.2, 3.2, n)], 1).astype(np.float32)
    b = a.copy()
    b[:, :2] += rng.normal(0, 1.0, (n, 2)).astype(np.float32)
    b[:, 2:4] *= rng.uniform(0.7, 1.3, (n, 2)).astype(np.float32)
    b[:, 4] += rng.normal(0, 0.6, n).astype(np.float32)
    # special cases: identical, nested, 90 degrees, disjoint
    a[:4] = [[1, 1, 2, 4, 0.3], [1, 1, 4, 4, 0], [1, 1, 2, 4, 0], [0, 0, 1, 1, 0]]
    b[:4] = [[1, 1, 2, 4, 0.3], [1, 1, 1, 1, 0.5], [1, 1, 2, 4, np.pi / 2], [5, 5, 1, 1, 0.2]]
    iou = np.array([ng.devRotateIoU(a[i].copy(), b[i].copy()) for i in range(n)], dtype=np.float64)
    np.savez_compressed(os.path.join(HERE, "rotate_iou.npz"), a=a, b=b, iou=iou)
    print("rotate_iou:", iou[:6])


def rotate_iou_eval_cases():
    """devRotateIoUEval (kitti_object_eval_python/rotate_iou.py:246-258) run as Python on float32 arrays,
    all four criteria.  Pairs whose intersection polygon would overrun the reference's 8-point scratch
    (IndexError here, undefined behaviour on the device) are dropped."""
    ri = ref_loader.ref("det3d.datasets.utils.kitti_object_eval_python.rotate_iou")
    rng = np.random.default_rng(11)
    n = 120
    a = np.stack([rng.uniform(0, 6, n), rng.uniform(0, 6, n), rng.uniform(1, 3, n),
                  rng.uniform(2, 5, n), rng.uniform(-3.2, 3.2, n)], 1).astype(np.float32)
    b = a.copy()
    b[:, :2] += rng.normal(0, 1.0, (n, 2)).astype(np.float32)
    b[:, 2:4] *= rng.uniform(0.7, 1.3, (n, 2)).astype(np.float32)
    b[:, 4] += rng.normal(0, 0.6, n).astype(np.float32)
    a[:5] = [[1, 1, 2, 4, 0.3], [1, 1, 4, 4, 0], [1, 1, 2, 4, 0], [0, 0, 1, 1, 0], [0, 0, 2, 2, 0]]
    b[:5] = [[1, 1, 2, 4, 0.3], [1, 1, 1, 1, 0.5], [1, 1, 2, 4, np.pi / 2], [5, 5, 1, 1, 0.2], [2, 0, 2, 2, 0]]
    keep, vals = [], []
    for i in range(n):
        try:
            row = [ri.devRotateIoUEval(a[i].copy(), b[i].copy(), c) for c in (-1, 0, 1, 2)]
        except IndexError:
            continue
        keep.append(i)
        vals.append(row)
    keep = np.array(keep)
    np.savez_compressed(os.path.join(HERE, "rotate_iou_eval.npz"), rbox1=a[keep], rbox2=b[keep],
                        criteria=np.array([-1, 0, 1, 2]), value=np.array(vals, dtype=np.float64))
    print("rotate_iou_eval:", len(keep), "pairs", np.array(vals)[:6, 0])


def vfe_pfn():
    ve = ref_loader.ref("det3d.models.readers.voxel_encoder")
    pe = ref_loader.ref("det3d.models.readers.pillar_encoder")
    torch.manual_seed(0)
    out = {}
    M, T, C = 50, 5, 4
    num = torch.randint(1, T + 1, (M,), dtype=torch.int32)
    feats = torch.randn(M, T, C)
    feats = feats * (torch.arange(T)[None, :] < num[:, None]).float()[:, :, None]
    vfe = ve.VoxelFeatureExtractorV3(num_input_features=4)
    out["vfe"] = dict(features=feats, num_voxels=num, out=vfe(feats, num))
    vfe3 = ve.VoxelFeatureExtractorV3(num_input_features=3)
    out["vfe3"] = dict(out=vfe3(feats, num))

    M, T, C = 60, 12, 4
    num = torch.randint(1, T + 1, (M,), dtype=torch.int32)
    feats = torch.rand(M, T, C) * torch.tensor([69.0, 79.0, 4.0, 1.0]) + torch.tensor([0, -39.6, -3.0, 0])
    feats = feats * (torch.arange(T)[None, :] < num[:, None]).float()[:, :, None]
    coors = torch.stack([torch.randint(0, 2, (M,)), torch.zeros(M, dtype=torch.long),
                         torch.randint(0, 496, (M,)), torch.randint(0, 432, (M,))], 1).int()
    pfn = pe.PillarFeatureNet(num_input_features=4, num_filters=(16,), with_distance=False,
                              voxel_size=PP_VS, pc_range=PP_RANGE)
    pfn.train()
    y_train = pfn(feats, num, coors)
    pfn.eval()
    y_eval = pfn(feats, num, coors)
    # unique coords per batch for the scatter
    lin = coors[:, 0].long() * 10 ** 7 + coors[:, 2].long() * 432 + coors[:, 3].long()
    _, first = np.unique(lin.numpy(), return_index=True)
    keep = torch.from_numpy(np.sort(first))
    sc = pe.PointPillarsScatter(num_input_features=16)
    canvas = sc(y_eval[keep].detach(), coors[keep], 2, [432, 496, 1])
    out["pfn"] = dict(state=pfn.state_dict(), features=feats, num_voxels=num, coors=coors,
                      out_train=y_train.detach(), out_eval=y_eval.detach(), keep=keep,
                      canvas_nonzero=canvas.nonzero(), canvas_vals=canvas[canvas != 0],
                      canvas_shape=torch.tensor(canvas.shape))
    torch.save(out, os.path.join(HERE, "vfe_pfn.pt"))
    print("vfe_pfn:", y_train.shape, canvas.shape)


RPN_CFGS = {
    "second": dict(layer_nums=[2], ds_layer_strides=[1], ds_num_filters=[16],
                   us_layer_strides=[1], us_num_filters=[16], num_input_features=16),
    "pillars": dict(layer_nums=[1, 2, 2], ds_layer_strides=[2, 2, 2], ds_num_filters=[8, 16, 32],
                    us_layer_strides=[1, 2, 4], us_num_filters=[8, 8, 8], num_input_features=8),
    "nusc_pp": dict(layer_nums=[1, 1, 1], ds_layer_strides=[2, 2, 2], ds_num_filters=[8, 16, 32],
                    us_layer_strides=[0.5, 1, 2], us_num_filters=[8, 8, 8], num_input_features=8),
    "cbgs": dict(layer_nums=[2, 2], ds_layer_strides=[1, 2], ds_num_filters=[8, 16],
                 us_layer_strides=[1, 2], us_num_filters=[16, 16], num_input_features=16),
}


def rpn_cases():
    rpn = ref_loader.ref("det3d.models.necks.rpn")
    out = {}
    for name, cfg in RPN_CFGS.items():
        torch.manual_seed(hash(name) % 1000)
        m = rpn.RPN(logger=logging.getLogger("RPN"), **cfg)
        x = torch.randn(2, cfg["num_input_features"], 16, 24)
        m.train()
        y_train = m(x)
        state_after_train = {k: v.clone() for k, v in m.state_dict().items()}
        m.eval()
        y_eval = m(x)
        out[name] = dict(cfg=cfg, state=state_after_train, x=x, out_train=y_train.detach(),
                         out_eval=y_eval.detach(), downsample_factor=float(m.downsample_factor))
        print("rpn", name, tuple(y_train.shape))
    torch.save(out, os.path.join(HERE, "rpn.pt"))


def _patch_nms():
    """predict() -> box_torch_ops.rotate_nms -> rotate_nms_cc needs the boost extension;
    substitute this repo's oracle for that single function (documented above)."""
    from oracle import clib

    bto = ref_loader.ref("det3d.core.bbox.box_torch_ops")
    bto.rotate_nms_cc = lambda dets, thresh: clib.rotate_nms_cc(dets, thresh)
    return bto


def head_cases():
    bto = _patch_nms()
    mg = ref_loader.ref("det3d.models.bbox_heads.mg_head")
    coders = ref_loader.ref("det3d.core.bbox.box_coders")
    ref_loader.ref("det3d.models.losses.losses")
    out = {}

    class Cfg(dict):
        __getattr__ = dict.__getitem__

    def make(name, tasks, coder, loss_aux, in_ch, H, W, direction_offset, test_cfg):
        torch.manual_seed(3)
        head = mg.MultiGroupHead(
            mode="3d", in_channels=in_ch, norm_cfg=None, tasks=tasks, weights=[1] * len(tasks),
            box_coder=coder, encode_background_as_zeros=True,
            loss_norm=dict(type="NormByNumPositives", pos_cls_weight=1.0, neg_cls_weight=2.0),
            loss_cls=dict(type="SigmoidFocalLoss", alpha=0.25, gamma=2.0, loss_weight=1.0),
            use_sigmoid_score=True,
            loss_bbox=dict(type="WeightedSmoothL1Loss", sigma=3.0,
                           code_weights=[1.0] * coder.code_size, codewise=True, loss_weight=2.0),
            encode_rad_error_by_sin=not coder.vec_encode, loss_aux=loss_aux,
            direction_offset=direction_offset, logger=logging.getLogger("head"))
        B = 2
        x = torch.randn(B, in_ch, H, W)
        # make some logits large so predict() has detections
        for t in head.tasks:
            torch.nn.init.normal_(t.conv_cls.bias, 0.0, 1.5)
        preds = head(x)
        anchors, labels, reg_targets = [], [], []
        for ti, t in enumerate(tasks):
            nc = len(t["class_names"])
            A = H * W * 2 * nc
            an = torch.zeros(B, A, coder.n_dim)
            ys, xs = torch.meshgrid(torch.arange(H).float(), torch.arange(W).float(), indexing="ij")
            base = torch.stack([xs * 2.0 + 1.0, ys * 2.0 - 10.0], -1).reshape(H * W, 1, 2)
            an[:, :, :2] = base.expand(H * W, 2 * nc, 2).reshape(1, A, 2)
            an[:, :, 2] = -1.0
            an[:, :, 3:6] = torch.tensor([1.6, 3.9, 1.56])
            rot = torch.tensor([0.0, 1.57]).repeat(H * W * nc)
            an[:, :, -1] = rot
            anchors.append(an)
            lab = torch.randint(-1, nc + 1, (B, A), dtype=torch.int32)
            lab[torch.rand(B, A) < 0.8] = 0
            labels.append(lab)
            reg_targets.append(torch.randn(B, A, coder.code_size) * 0.3)
        example = dict(voxels=None, num_points=None, coordinates=None, anchors=anchors, labels=labels,
                       reg_targets=reg_targets, metadata=[dict(token=i) for i in range(B)])
        head.train()
        losses = head.loss(example, preds)
        total = sum(losses["loss"])
        grads = torch.autograd.grad(total, [x.requires_grad_(False)] if False else list(head.parameters()),
                                    allow_unused=True)
        head.eval()
        with torch.no_grad():
            dets = head.predict(example, head(x), test_cfg)
        rec = dict(
            tasks=tasks, coder=dict(n_dim=coder.n_dim, vec=coder.vec_encode, linear_dim=coder.linear_dim),
            loss_aux=loss_aux, in_ch=in_ch, direction_offset=direction_offset, test_cfg=dict(test_cfg, nms=dict(test_cfg["nms"])),
            state=head.state_dict(), x=x, preds=[{k: v.detach() for k, v in p.items()} for p in preds],
            anchors=anchors, labels=labels, reg_targets=reg_targets,
            loss=[l.detach() for l in losses["loss"]],
            loss_parts={k: [torch.as_tensor(v).detach().float() for v in losses[k]]
                        for k in ("cls_pos_loss", "cls_neg_loss", "dir_loss_reduced",
                                  "cls_loss_reduced", "loc_loss_reduced")},
            grad_norms=[float(g.norm()) if g is not None else 0.0 for g in grads],
            dets=[{k: (v.clone() if torch.is_tensor(v) else v) for k, v in d.items()} for d in dets])
        out[name] = rec
        print("head", name, "loss", [float(l) for l in losses["loss"]], "dets", [d["scores"].shape[0] for d in dets])

    kitti_test = Cfg(nms=Cfg(use_rotate_nms=True, use_multi_class_nms=False, nms_pre_max_size=1000,
                             nms_post_max_size=100, nms_iou_threshold=0.01),
                     score_threshold=0.3, post_center_limit_range=[0, -40.0, -5.0, 70.4, 40.0, 5.0],
                     max_per_img=100)
    make("kitti", [dict(num_class=1, class_names=["Car"])],
         coders.GroundBox3dCoderTorch(False, False, n_dim=7),
         dict(type="WeightedSoftmaxClassificationLoss", name="direction_classifier", loss_weight=0.2),
         16, 8, 10, 0.0, kitti_test)
    nusc_test = Cfg(nms=Cfg(use_rotate_nms=True, use_multi_class_nms=False, nms_pre_max_size=1000,
                            nms_post_max_size=83, nms_iou_threshold=0.2),
                    score_threshold=0.1, post_center_limit_range=[-61.2, -61.2, -10.0, 61.2, 61.2, 10.0],
                    max_per_img=500)
    make("nusc", [dict(num_class=1, class_names=["car"]),
                  dict(num_class=2, class_names=["truck", "construction_vehicle"])],
         coders.GroundBox3dCoderTorch(False, True, n_dim=9), None, 16, 6, 8, 0.0, nusc_test)
    lyft_test = Cfg(nms=Cfg(use_rotate_nms=True, use_multi_class_nms=False, nms_pre_max_size=1000,
                            nms_post_max_size=80, nms_iou_threshold=0.2),
                    score_threshold=0.1, post_center_limit_range=[-100, -100, -10.0, 100, 100, 10.0],
                    max_per_img=500)
    make("lyft", [dict(num_class=2, class_names=["car", "bus"])],
         coders.GroundBox3dCoderTorch(False, False, n_dim=7),
         dict(type="WeightedSoftmaxClassificationLoss", name="direction_classifier", loss_weight=0.2),
         16, 6, 8, 0.785, lyft_test)
    torch.save(out, os.path.join(HERE, "head.pt"))


def anchors_targets():
    bnp = ref_loader.ref("det3d.core.bbox.box_np_ops")
    tops = ref_loader.ref("det3d.core.anchor.target_ops")
    rng = np.random.default_rng(21)
    fs = [1, 25, 22]
    rg = [0, -40.0, -1.0, 70.4, 40.0, -1.0]
    # numpy >= 2 returns a tuple from meshgrid; the reference (written for numpy 1.x) assigns
    # into it (box_np_ops.py:791) -- shim the return type only
    _mg = np.meshgrid
    np.meshgrid = lambda *a, **k: list(_mg(*a, **k))
    try:
        anc = bnp.create_anchors_3d_range(fs, rg, [1.6, 3.9, 1.56], [0, 1.57])
    finally:
        np.meshgrid = _mg
    anchors = anc.reshape(-1, 7)
    gt = np.zeros((9, 7), np.float32)
    gt[:, 0] = rng.uniform(5, 65, 9)
    gt[:, 1] = rng.uniform(-35, 35, 9)
    gt[:, 2] = -1.0
    gt[:, 3:6] = np.array([1.6, 3.9, 1.56]) * rng.uniform(0.85, 1.15, (9, 3))
    gt[:, 6] = rng.uniform(-3.1, 3.1, 9)
    gt[0, :2] = anchors[200, :2]  # one exactly on an anchor centre
    gt[0, 6] = 0.0

    def sim(a, g):
        return bnp.iou_jit(bnp.rbbox2d_to_near_bbox(a[:, [0, 1, 3, 4, 6]]),
                           bnp.rbbox2d_to_near_bbox(g[:, [0, 1, 3, 4, 6]]), eps=0.0)

    ret = tops.create_target_np(anchors, gt, sim, bnp.second_box_encode, matched_threshold=0.6,
                                unmatched_threshold=0.45, positive_fraction=None, box_code_size=7)
    np.savez_compressed(os.path.join(HERE, "anchors_targets.npz"), feature_size=np.array(fs),
                        anchor_range=np.array(rg), anchors6d=anc, gt=gt, labels=ret["labels"],
                        bbox_targets=ret["bbox_targets"], weights=ret["bbox_outside_weights"])
    print("anchors_targets:", anc.shape, int((ret["labels"] > 0).sum()), "pos",
          int((ret["labels"] == 0).sum()), "neg")


def task_targets():
    """TargetAssigner.generate_anchors / generate_anchors_dict / assign_v2 (target_assigner.py:68-200) for the
    nuScenes [bus, trailer] task group: 9-dim boxes, angle-vector box coder, per-class anchors + thresholds, a
    NaN velocity, a class-less frame; and the Lyft [motorcycle, bicycle] group with 7-dim boxes."""
    bnp = ref_loader.ref("det3d.core.bbox.box_np_ops")
    ta = ref_loader.ref("det3d.core.anchor.target_assigner")
    ag = ref_loader.ref("det3d.core.anchor.anchor_generator")
    bc = ref_loader.ref("det3d.core.bbox.box_coders")
    rs = ref_loader.ref("det3d.core.bbox.region_similarity")
    _mg = np.meshgrid
    np.meshgrid = lambda *a, **k: list(_mg(*a, **k))
    out = {}
    try:
        for tag, nd, vec, rng_xy, gens in (
                ("nusc", 9, True, 51.2, [("bus", [2.94, 10.5, 3.47], -0.085, 0.55, 0.4),
                                         ("trailer", [2.90, 12.29, 3.87], 0.115, 0.5, 0.35)]),
                ("lyft", 7, False, 100.8, [("motorcycle", [0.97, 2.36, 1.60], -0.9, 0.55, 0.4),
                                           ("bicycle", [0.64, 1.76, 1.46], -1.04, 0.55, 0.4)])):
            generators = [ag.AnchorGeneratorRange(sizes=sz, anchor_ranges=[-rng_xy, -rng_xy, z, rng_xy, rng_xy, z],
                                                  rotations=[0, 1.57], velocities=[0, 0] if nd == 9 else None,
                                                  match_threshold=hi, unmatch_threshold=lo, class_name=name)
                          for name, sz, z, hi, lo in gens]
            assigner = ta.TargetAssigner(box_coder=bc.GroundBox3dCoder(vec_encode=vec, n_dim=nd),
                                         anchor_generators=generators,
                                         region_similarity_calculator=rs.NearestIouSimilarity(),
                                         positive_fraction=None, sample_size=512)
            fmap = [1, 32, 32]
            anchors = assigner.generate_anchors(fmap)["anchors"]
            adict = assigner.generate_anchors_dict(fmap)
            out[tag + "_anchors"] = anchors.reshape(-1, nd)
            rng = np.random.default_rng(5 if nd == 9 else 6)
            for frame, (n0, n1) in enumerate([(4, 3), (0, 5), (0, 0)]):
                n = n0 + n1
                gt = np.zeros((n, nd), np.float32)
                names = np.array([gens[0][0]] * n0 + [gens[1][0]] * n1)
                cls = np.array([1] * n0 + [2] * n1, np.int32)
                gt[:, 0] = rng.uniform(-0.8 * rng_xy, 0.8 * rng_xy, n)
                gt[:, 1] = rng.uniform(-0.8 * rng_xy, 0.8 * rng_xy, n)
                for i in range(n):
                    gt[i, 2] = gens[cls[i] - 1][2]
                    gt[i, 3:6] = np.array(gens[cls[i] - 1][1]) * rng.uniform(0.85, 1.15, 3)
                gt[:, -1] = rng.uniform(-3.1, 3.1, n)
                if nd == 9:
                    gt[:, 6:8] = rng.normal(0, 2, (n, 2))
                    if n:
                        gt[0, 6:8] = np.nan                      # unlabeled velocity (nan_to_num in assign_v2)
                if n:
                    flat = anchors.reshape(-1, nd)
                    gt[0, :2] = flat[(16 * 32 + 16) * 4 + 2 * (cls[0] - 1), :2]   # exactly on an anchor centre
                    gt[0, -1] = 0.0
                ret = assigner.assign_v2(adict, gt.copy(), None, gt_classes=cls, gt_names=names)
                out["%s_f%d_gt" % (tag, frame)] = gt
                out["%s_f%d_cls" % (tag, frame)] = cls
                out["%s_f%d_labels" % (tag, frame)] = ret["labels"]
                out["%s_f%d_targets" % (tag, frame)] = ret["bbox_targets"]
                out["%s_f%d_weights" % (tag, frame)] = ret["bbox_outside_weights"]
                print("task_targets", tag, frame, anchors.shape, "pos per class",
                      [(ret["labels"] == c).sum() for c in (1, 2)], "ignored", (ret["labels"] < 0).sum())
    finally:
        np.meshgrid = _mg
    np.savez_compressed(os.path.join(HERE, "task_targets.npz"), **out)


def augment_cases():
    """random_flip / random_flip_both + global_rotation + global_scaling_v2 (core/sampler/preprocess.py:796-861)
    exactly as Preprocess.__call__ chains them (pipelines/preprocess.py:197-212), seeded NumPy stream."""
    prep = ref_loader.ref("det3d.core.sampler.preprocess")
    out = {}
    for tag, nd, both in (("kitti", 7, False), ("nusc", 9, True)):
        for frame in range(4):
            rng = np.random.default_rng(100 + frame + (10 if both else 0))
            pts = rng.uniform(-30, 30, (500, 5 if both else 4)).astype(np.float32)
            gt = rng.uniform(-20, 20, (6, nd)).astype(np.float32)
            gt[:, 3:6] = rng.uniform(0.5, 5, (6, 3))
            gt[:, -1] = rng.uniform(-3.1, 3.1, 6)
            out["%s_f%d_points_in" % (tag, frame)] = pts.copy()
            out["%s_f%d_gt_in" % (tag, frame)] = gt.copy()
            np.random.seed(7 + frame)
            if both:
                gt, pts = prep.random_flip_both(gt, pts)
            else:
                gt, pts = prep.random_flip(gt, pts)
            gt, pts = prep.global_rotation(gt, pts, rotation=[-0.3925, 0.3925] if both else [-0.78539816, 0.78539816])
            gt, pts = prep.global_scaling_v2(gt, pts, 0.95, 1.05)
            out["%s_f%d_points_out" % (tag, frame)] = pts
            out["%s_f%d_gt_out" % (tag, frame)] = gt
    np.savez_compressed(os.path.join(HERE, "augment.npz"), **out)
    print("augment:", sorted(out)[:4], "...")


def object_noise_cases():
    """noise_per_object_v3_ (core/sampler/preprocess.py:587-709) with the KITTI config's settings, seeded NumPy
    stream, three frames with boxes close enough that candidates collide.  numba compiles `x is True` on a boolean
    to a value comparison; run as plain Python the same text compares identities with a numpy.bool_ and is always
    False, which would silently skip box_collision_test's containment branch -- so that one function is re-executed
    with `is True/False` spelled `== True/False` (numba's semantics) before the reference code runs."""
    import inspect
    import re

    prep = ref_loader.ref("det3d.core.sampler.preprocess")
    src = inspect.getsource(prep.box_collision_test)
    src = re.sub(r"^@numba\.njit\n", "", src)
    src = src.replace(" is True", " == True").replace(" is False", " == False")
    ns = prep.__dict__
    exec(compile(src, "<box_collision_test with numba bool semantics>", "exec"), ns)
    out = {}
    for frame in range(3):
        rng = np.random.default_rng(40 + frame)
        n = 8
        gt = np.zeros((n, 7), np.float32)
        gx, gy = np.meshgrid(np.arange(4), np.arange(2))
        gt[:, 0] = 10 + 4.2 * gx.ravel() + rng.uniform(-0.5, 0.5, n)
        gt[:, 1] = -3 + 4.5 * gy.ravel() + rng.uniform(-0.5, 0.5, n)
        gt[:, 2] = rng.uniform(-1.2, -0.8, n)
        gt[:, 3:6] = np.array([1.7, 4.0, 1.6], np.float32) * rng.uniform(0.9, 1.1, (n, 3)).astype(np.float32)
        gt[:, 6] = rng.uniform(-3.1, 3.1, n)
        inside = []
        for b in gt:  # points inside every box (they must move with it) + clutter around
            loc = rng.uniform(-0.45, 0.45, (60, 3)) * b[3:6]
            c, s_ = np.cos(b[6]), np.sin(b[6])
            x = loc[:, 0] * c + loc[:, 1] * s_
            y = -loc[:, 0] * s_ + loc[:, 1] * c
            inside.append(np.stack([x + b[0], y + b[1], loc[:, 2] + b[2]], 1))
        clutter = np.concatenate([rng.uniform(0, 30, (600, 1)), rng.uniform(-10, 10, (600, 1)),
                                  rng.uniform(-2, 0.5, (600, 1))], 1)
        pts = np.concatenate(inside + [clutter]).astype(np.float32)
        pts = np.concatenate([pts, rng.uniform(0, 1, (len(pts), 1)).astype(np.float32)], 1)
        valid = np.ones((n,), np.bool_)
        valid[5] = False
        out["f%d_gt_in" % frame], out["f%d_points_in" % frame], out["f%d_valid" % frame] = gt.copy(), pts.copy(), valid
        np.random.seed(11 + frame)
        state = np.random.get_state()
        prep.noise_per_object_v3_(gt, pts, valid, rotation_perturb=[-0.78539816, 0.78539816],
                                  center_noise_std=[1.0, 1.0, 0.5], global_random_rot_range=[0.0, 0.0],
                                  group_ids=None, num_try=40)
        out["f%d_gt_out" % frame], out["f%d_points_out" % frame] = gt, pts
        del state
    np.savez_compressed(os.path.join(HERE, "object_noise.npz"), **out)
    moved = [int((out["f%d_gt_out" % f][:, :2] != out["f%d_gt_in" % f][:, :2]).any(1).sum()) for f in range(3)]
    print("object_noise: boxes moved per frame", moved)


def _numba_bool_semantics(prep):
    """see object_noise_cases: box_collision_test with `is True/False` as numba compiles them"""
    import inspect
    import re

    src = inspect.getsource(prep.box_collision_test)
    src = re.sub(r"^@numba\.njit\n", "", src)
    src = src.replace(" is True", " == True").replace(" is False", " == False")
    exec(compile(src, "<box_collision_test with numba bool semantics>", "exec"), prep.__dict__)


def gt_sampler_cases():
    """DataBaseSamplerV2 (core/sampler/sample_ops.py:13-298) on a synthetic three-class database written to a temp
    directory, KITTI-style groups and filters, four consecutive frames (the BatchSamplers are stateful) -- plus the
    paste step of the train pipeline (pipelines/preprocess.py:163-181)."""
    import logging
    import tempfile

    prep = ref_loader.ref("det3d.core.sampler.preprocess")
    _numba_bool_semantics(prep)
    sample_ops = ref_loader.ref("det3d.core.sampler.sample_ops")
    box_np_ops = ref_loader.ref("det3d.core.bbox.box_np_ops")
    rng = np.random.default_rng(77)
    sizes = {"Car": (1.7, 4.0, 1.6), "Pedestrian": (0.6, 0.8, 1.75), "Cyclist": (0.6, 1.8, 1.7)}
    out, db_infos = {}, {}
    tmp = tempfile.mkdtemp()
    for name, (w, l, h) in sizes.items():
        infos = []
        for k in range(40):
            box = np.array([rng.uniform(2, 60), rng.uniform(-25, 25), rng.uniform(-1.5, -0.5), w * rng.uniform(0.9, 1.1),
                            l * rng.uniform(0.9, 1.1), h * rng.uniform(0.9, 1.1), rng.uniform(-3.1, 3.1)], np.float32)
            npts = int(rng.integers(2, 60))
            pts = np.concatenate([rng.uniform(-0.5, 0.5, (npts, 3)) * box[3:6], rng.uniform(0, 1, (npts, 1))], 1).astype(np.float32)
            path = "%s_%d.bin" % (name, k)
            pts.tofile(os.path.join(tmp, path))
            infos.append(dict(name=name, path=path, box3d_lidar=box, num_points_in_gt=npts,
                              difficulty=int(rng.integers(-1, 3)), gt_idx=k, image_idx=k))
            out["db_%s_%d_points" % (name, k)] = pts
        db_infos[name] = infos
        out["db_%s_boxes" % name] = np.stack([i["box3d_lidar"] for i in infos])
        out["db_%s_npts" % name] = np.array([i["num_points_in_gt"] for i in infos])
        out["db_%s_difficulty" % name] = np.array([i["difficulty"] for i in infos])
    groups = [dict(Car=15), dict(Pedestrian=8), dict(Cyclist=8)]
    prepor = prep.DataBasePreprocessor([
        prep.DBFilterByMinNumPoint(dict(Car=5, Pedestrian=8, Cyclist=8), logger=logging.getLogger("g")),
        prep.DBFilterByDifficulty([-1], logger=logging.getLogger("g"))])
    np.random.seed(123)
    sampler = sample_ops.DataBaseSamplerV2(db_infos, groups, prepor, 1.0, None, logger=logging.getLogger("g"))
    for frame in range(4):
        n = 3 + frame
        gt = np.zeros((n, 7), np.float32)
        gt[:, 0], gt[:, 1] = rng.uniform(5, 50, n), rng.uniform(-20, 20, n)
        gt[:, 2] = -1.0
        gt[:, 3:6] = sizes["Car"]
        gt[:, 6] = rng.uniform(-3, 3, n)
        names = np.array(["Car"] * (n - 1) + ["Pedestrian"])
        pts = np.concatenate([rng.uniform(0, 65, (3000, 1)), rng.uniform(-30, 30, (3000, 1)), rng.uniform(-2, 0.5, (3000, 1)),
                              rng.uniform(0, 1, (3000, 1))], 1).astype(np.float32)
        out["f%d_gt" % frame], out["f%d_names" % frame], out["f%d_points" % frame] = gt, names, pts
        s = sampler.sample_all(tmp, gt, names, 4, False, gt_group_ids=None, calib=None)
        assert s is not None
        for k in ("gt_names", "difficulty", "gt_boxes", "points", "gt_masks", "group_ids"):
            out["f%d_s_%s" % (frame, k)] = s[k]
        # the paste step, as Preprocess.__call__ does it
        masks = box_np_ops.points_in_rbbox(pts, s["gt_boxes"])
        kept = pts[np.logical_not(masks.any(-1))]
        out["f%d_pasted_points" % frame] = np.concatenate([s["points"], kept], axis=0)
    np.savez_compressed(os.path.join(HERE, "gt_sampler.npz"), **out)
    print("gt_sampler: sampled per frame", [len(out["f%d_s_gt_names" % f]) for f in range(4)])


def loading_cases():
    """LoadPointCloudFromFile (datasets/pipelines/loading.py:66-163) on synthetic files: a nuScenes key frame with six
    candidate sweeps (transform matrices, time lags, points near the sensor to be removed, one file with a partial
    trailing row), KITTI with and without a `_reduced` twin, Lyft.  The files are not kept: the test rewrites them
    from the arrays stored here."""
    import tempfile

    loading = ref_loader.ref("det3d.datasets.pipelines.loading")
    rng = np.random.default_rng(5)
    tmp = tempfile.mkdtemp()
    out = {}

    def cloud(n):
        p = np.concatenate([rng.uniform(-40, 40, (n, 2)), rng.uniform(-3, 2, (n, 1)), rng.uniform(0, 255, (n, 1)),
                            rng.uniform(0, 31, (n, 1))], 1).astype(np.float32)
        p[: n // 10, :2] = rng.uniform(-0.9, 0.9, (n // 10, 2))  # inside the 1 m exclusion square
        return p

    key = cloud(700)
    key.tofile(os.path.join(tmp, "key.bin"))
    out["nusc_key"] = key
    sweeps = []
    for i in range(6):
        p = cloud(300 + 40 * i)
        raw = p.ravel()
        if i == 2:
            raw = np.concatenate([raw, np.array([1.0, 2.0, 3.0], np.float32)])  # partial row at the end
        raw.tofile(os.path.join(tmp, "sweep%d.bin" % i))
        ang = rng.uniform(-0.2, 0.2)
        m = np.eye(4)
        m[:2, :2] = [[np.cos(ang), -np.sin(ang)], [np.sin(ang), np.cos(ang)]]
        m[:3, 3] = rng.uniform(-2, 2, 3)
        sweeps.append(dict(lidar_path=os.path.join(tmp, "sweep%d.bin" % i), transform_matrix=m if i != 4 else None,
                           time_lag=0.05 * (i + 1)))
        out["nusc_sweep%d_raw" % i], out["nusc_sweep%d_matrix" % i] = raw, m
        out["nusc_sweep%d_lag" % i] = np.float64(0.05 * (i + 1))
    step = loading.LoadPointCloudFromFile(dataset="NuScenesDataset")
    for nsweeps in (1, 4):
        res = {"lidar": {"nsweeps": nsweeps}, "metadata": {}}
        np.random.seed(9)
        res, _ = step(res, dict(lidar_path=os.path.join(tmp, "key.bin"), sweeps=sweeps))
        out["nusc_n%d_points" % nsweeps] = res["lidar"]["points"]
        out["nusc_n%d_times" % nsweeps] = res["lidar"]["times"]
        out["nusc_n%d_combined" % nsweeps] = res["lidar"]["combined"]
    kitti = rng.uniform(-30, 30, (500, 4)).astype(np.float32)
    os.makedirs(os.path.join(tmp, "training", "velodyne"))
    kitti.tofile(os.path.join(tmp, "training", "velodyne", "000001.bin"))
    out["kitti_raw"] = kitti
    step = loading.LoadPointCloudFromFile(dataset="KittiDataset")
    info = dict(point_cloud=dict(velodyne_path="training/velodyne/000001.bin"))
    res, _ = step({"lidar": {}, "metadata": {"image_prefix": tmp, "num_point_features": 4}}, info)
    out["kitti_points"] = res["lidar"]["points"]
    os.makedirs(os.path.join(tmp, "training", "velodyne_reduced"))
    kitti[:200].tofile(os.path.join(tmp, "training", "velodyne_reduced", "000001.bin"))
    res, _ = step({"lidar": {}, "metadata": {"image_prefix": tmp, "num_point_features": 4}}, info)
    out["kitti_reduced_points"] = res["lidar"]["points"]
    step = loading.LoadPointCloudFromFile(dataset="LyftDataset")
    res, _ = step({"lidar": {}, "metadata": {}}, dict(ref_info=dict(LIDAR_TOP=dict(lidar_path=os.path.join(tmp, "key.bin")))))
    out["lyft_points"] = res["lidar"]["points"]
    np.savez_compressed(os.path.join(HERE, "loading.npz"), **out)
    print("loading:", {k: v.shape for k, v in out.items() if k.endswith("points")})


def preprocess_cases():
    """The reference's Preprocess step itself (datasets/pipelines/preprocess.py:29-255), train mode, KITTI settings
    (GT-AUG sampler from a pickled synthetic database, per-object noise, flip, rotation, scaling, shuffle) and
    nuScenes settings (double flip, 5-column `combined` cloud, no sampler hits), plus val mode.  The output cloud is
    stored in full: its row ORDER is part of the contract (the voxelizer is order sensitive)."""
    import pickle
    import tempfile

    from det3d_amd.config import ConfigDict

    prep = ref_loader.ref("det3d.core.sampler.preprocess")
    _numba_bool_semantics(prep)
    pp = ref_loader.ref("det3d.datasets.pipelines.preprocess")
    rng = np.random.default_rng(321)
    tmp = tempfile.mkdtemp()
    sizes = {"Car": (1.7, 4.0, 1.6), "Pedestrian": (0.6, 0.8, 1.75)}
    db, out = {}, {}
    for name, (w, l, h) in sizes.items():
        infos = []
        for k in range(30):
            box = np.array([rng.uniform(5, 60), rng.uniform(-30, 30), -1.0, w, l, h, rng.uniform(-3, 3)], np.float32)
            npts = int(rng.integers(6, 40))
            pts = np.concatenate([rng.uniform(-0.5, 0.5, (npts, 3)) * box[3:6], rng.uniform(0, 1, (npts, 1))], 1).astype(np.float32)
            pts.tofile(os.path.join(tmp, "%s_%d.bin" % (name, k)))
            infos.append(dict(name=name, path="%s_%d.bin" % (name, k), box3d_lidar=box, num_points_in_gt=npts,
                              difficulty=int(rng.integers(0, 3)), gt_idx=k, image_idx=k))
            out["db_%s_%d_points" % (name, k)] = pts
        db[name] = infos
        out["db_%s_boxes" % name] = np.stack([i["box3d_lidar"] for i in infos])
        out["db_%s_npts" % name] = np.array([i["num_points_in_gt"] for i in infos])
        out["db_%s_difficulty" % name] = np.array([i["difficulty"] for i in infos])
    with open(os.path.join(tmp, "dbinfos.pkl"), "wb") as f:
        pickle.dump(db, f)
    sampler_cfg = ConfigDict(type="GT-AUG", enable=True, db_info_path=os.path.join(tmp, "dbinfos.pkl"),
                             sample_groups=[dict(Car=10), dict(Pedestrian=6)],
                             db_prep_steps=[dict(filter_by_min_num_points=dict(Car=5, Pedestrian=5)),
                                            dict(filter_by_difficulty=[-1])],
                             global_random_rotation_range_per_object=[0, 0], rate=1.0)

    def cfg(mode, names, **kw):
        base = dict(mode=mode, shuffle_points=True, gt_loc_noise=[1.0, 1.0, 0.5], gt_rot_noise=[-0.785, 0.785],
                    global_rot_noise=[-0.785, 0.785], global_scale_noise=[0.95, 1.05], global_rot_per_obj_range=[0, 0],
                    global_trans_noise=[0.0, 0.0, 0.0], remove_points_after_sample=True, gt_drop_percentage=0.0,
                    gt_drop_max_keep_points=15, remove_unknown_examples=False, remove_environment=False,
                    db_sampler=sampler_cfg, class_names=names)
        base.update(kw)
        return ConfigDict(base)

    def frame(nd, n_box, n_pts, names_pool):
        gt = np.zeros((n_box, nd), np.float32)
        gt[:, 0], gt[:, 1], gt[:, 2] = rng.uniform(5, 55, n_box), rng.uniform(-25, 25, n_box), -1.0
        gt[:, 3:6] = [1.7, 4.0, 1.6]
        gt[:, -1] = rng.uniform(-3, 3, n_box)
        names = np.array([names_pool[i % len(names_pool)] for i in range(n_box)])
        cols = 5 if nd == 9 else 4
        pts = np.concatenate([rng.uniform(0, 65, (n_pts, 1)), rng.uniform(-35, 35, (n_pts, 1)), rng.uniform(-2.5, 0.5, (n_pts, 1)),
                              rng.uniform(0, 1, (n_pts, cols - 3))], 1).astype(np.float32)
        return gt, names, pts

    # ---- KITTI-style: sampler active, two frames in a row (stateful sampler), DontCare dropped, Van kept out
    np.random.seed(2024)
    step = pp.Preprocess(cfg=cfg("train", ["Car", "Pedestrian"]))
    for f in range(2):
        gt, names, pts = frame(7, 5, 2500, ["Car", "DontCare", "Van", "Car", "Pedestrian"])
        out["kitti%d_gt" % f], out["kitti%d_names" % f], out["kitti%d_points" % f] = gt.copy(), names.copy(), pts.copy()
        res = dict(type="KittiDataset", metadata=dict(image_prefix=tmp, num_point_features=4),
                   lidar=dict(points=pts, annotations=dict(boxes=gt, names=names)))
        res, _ = step(res, {})
        a = res["lidar"]["annotations"]
        out["kitti%d_out_points" % f], out["kitti%d_out_boxes" % f] = res["lidar"]["points"], a["gt_boxes"]
        out["kitti%d_out_names" % f], out["kitti%d_out_classes" % f] = a["gt_names"], a["gt_classes"]
    # ---- nuScenes-style: 9-dim boxes, `combined` 5-column cloud, double flip, a sampler with nothing to add
    np.random.seed(2025)
    step = pp.Preprocess(cfg=cfg("train", ["Car", "Pedestrian"], gt_loc_noise=[0.0, 0.0, 0.0], gt_rot_noise=[0.0, 0.0],
                                 db_sampler=ConfigDict(sampler_cfg, sample_groups=[dict(Car=1)])))
    gt, names, pts = frame(9, 4, 1500, ["Car", "Pedestrian"])
    gt[:, 6:8] = rng.normal(0, 2, (4, 2))
    out["nusc_gt"], out["nusc_names"], out["nusc_points"] = gt.copy(), names.copy(), pts.copy()
    res = dict(type="NuScenesDataset", metadata=dict(image_prefix=tmp, num_point_features=5),
               lidar=dict(combined=pts, annotations=dict(boxes=gt, names=names)))
    res, _ = step(res, {})
    out["nusc_out_points"], out["nusc_out_boxes"] = res["lidar"]["points"], res["lidar"]["annotations"]["gt_boxes"]
    out["nusc_out_names"] = res["lidar"]["annotations"]["gt_names"]
    # ---- val mode: only the shuffle flag matters (off in the configs)
    step = pp.Preprocess(cfg=ConfigDict(mode="val", shuffle_points=False, remove_environment=False,
                                        remove_unknown_examples=False))
    res, _ = step(dict(type="KittiDataset", metadata={}, lidar=dict(points=out["nusc_points"][:, :4].copy())), {})
    out["val_out_points"] = res["lidar"]["points"]
    np.savez_compressed(os.path.join(HERE, "preprocess.npz"), **out)
    print("preprocess:", {k: v.shape for k, v in out.items() if "_out_" in k})


def annotation_cases():
    """LoadPointCloudAnnotations (datasets/pipelines/loading.py:166-222) on a synthetic KITTI info (camera-frame boxes,
    calibration matrices, a DontCare row) and a nuScenes info."""
    loading = ref_loader.ref("det3d.datasets.pipelines.loading")
    rng = np.random.default_rng(8)
    n = 6
    ang = 0.01
    r0 = np.eye(4)
    r0[:3, :3] = [[np.cos(ang), -np.sin(ang), 0], [np.sin(ang), np.cos(ang), 0], [0, 0, 1]]
    v2c = np.eye(4)
    v2c[:3, :3] = [[0, -1, 0], [0, 0, -1], [1, 0, 0]]  # lidar x-forward -> camera z-forward
    v2c[:3, 3] = [0.0, -0.08, -0.27]
    p2 = np.array([[721.5, 0, 609.6, 44.9], [0, 721.5, 172.9, 0.2], [0, 0, 1, 0.003], [0, 0, 0, 1]])
    annos = dict(name=np.array(["Car", "DontCare", "Pedestrian", "Car", "Van", "Cyclist"]),
                 location=rng.uniform(-10, 30, (n, 3)), dimensions=rng.uniform(0.5, 4, (n, 3)),
                 rotation_y=rng.uniform(-3, 3, n), bbox=rng.uniform(0, 300, (n, 4)), difficulty=rng.integers(0, 3, n))
    info = dict(calib=dict(R0_rect=r0, Tr_velo_to_cam=v2c, P2=p2), annos={k: v.copy() for k, v in annos.items()})
    res, _ = loading.LoadPointCloudAnnotations()(dict(type="KittiDataset", lidar={}, cam={}, metadata={}), info)
    out = {"kitti_%s" % k: v for k, v in annos.items()}
    out.update(kitti_r0=r0, kitti_v2c=v2c, kitti_p2=p2, kitti_out_boxes=res["lidar"]["annotations"]["boxes"],
               kitti_out_names=res["lidar"]["annotations"]["names"], kitti_out_cam_boxes=res["cam"]["annotations"]["boxes"])
    gt = rng.uniform(-40, 40, (5, 7))
    info = dict(gt_boxes=gt, gt_names=np.array(["car", "bus", "car", "truck", "pedestrian"]),
                gt_boxes_token=np.array(["a", "b", "c", "d", "e"]), gt_boxes_velocity=rng.normal(0, 2, (5, 3)))
    res, _ = loading.LoadPointCloudAnnotations()(dict(type="NuScenesDataset", lidar={}, metadata={}), info)
    out.update(nusc_gt=gt, nusc_vel=info["gt_boxes_velocity"], nusc_out_boxes=res["lidar"]["annotations"]["boxes"],
               nusc_out_velocities=res["lidar"]["annotations"]["velocities"])
    np.savez_compressed(os.path.join(HERE, "annotations.npz"), **out)
    print("annotations:", out["kitti_out_boxes"].shape, out["kitti_out_boxes"].dtype)


def sampler_cases():
    """DistributedGroupSampler (datasets/loader/sampler.py:138-224): indices per rank and epoch."""
    sm = ref_loader.ref("det3d.datasets.loader.sampler")

    class _DS:
        def __init__(self, flag):
            self.flag = flag

        def __len__(self):
            return len(self.flag)

    out = {}
    rng = np.random.default_rng(4)
    for tag, n, spg, world in (("a", 37, 4, 3), ("b", 64, 4, 8), ("c", 53, 6, 2)):
        flag = (rng.random(n) < 0.3).astype(np.uint8) if tag != "b" else np.zeros(n, np.uint8)
        out[tag + "_flag"] = flag
        out[tag + "_cfg"] = np.array([spg, world])
        for epoch in (0, 3):
            for rank in range(world):
                s = sm.DistributedGroupSampler(_DS(flag), samples_per_gpu=spg, num_replicas=world, rank=rank)
                s.set_epoch(epoch)
                out["%s_e%d_r%d" % (tag, epoch, rank)] = np.array(list(iter(s)), np.int64)
    np.savez_compressed(os.path.join(HERE, "sampler.npz"), **out)
    print("sampler:", {k: v.shape for k, v in out.items() if k.endswith("_r0")})


def collate_cases():
    """collate_kitti (torchie/parallel/collate.py:90-150) on three ragged frames with every key class."""
    col = ref_loader.ref("det3d.torchie.parallel.collate")
    rng = np.random.default_rng(8)
    batch = []
    for i, (m, n, g) in enumerate([(5, 40, 2), (3, 25, 0), (7, 33, 4)]):
        batch.append(dict(
            voxels=rng.random((m, 4, 4)).astype(np.float32), num_points=rng.integers(1, 5, m).astype(np.int32),
            num_voxels=np.array([m], np.int64), coordinates=rng.integers(0, 30, (m, 3)).astype(np.int32),
            points=rng.random((n, 4)).astype(np.float32), shape=np.array([70, 80, 10], np.int64),
            anchors=[rng.random((12, 7)).astype(np.float32), rng.random((6, 7)).astype(np.float32)],
            labels=[rng.integers(-1, 2, 12).astype(np.int32), rng.integers(-1, 3, 6).astype(np.int32)],
            reg_targets=[rng.random((12, 7)).astype(np.float32), rng.random((6, 7)).astype(np.float32)],
            reg_weights=[rng.random(12).astype(np.float32), rng.random(6).astype(np.float32)],
            gt_boxes=[rng.random((g, 7)), rng.random((max(g - 1, 0), 7))],
            calib=dict(rect=rng.random((4, 4)).astype(np.float32), P2=rng.random((4, 4)).astype(np.float32)),
            metadata=dict(token="frame%d" % i, num_point_features=4)))
    out = col.collate_kitti(batch)
    torch.save(dict(batch=batch, out=out), os.path.join(HERE, "collate.pt"))
    print("collate:", {k: (tuple(v.shape) if hasattr(v, "shape") else type(v).__name__) for k, v in out.items()})


def assign_pipeline():
    """The reference's AssignTarget pipeline step itself (datasets/pipelines/preprocess.py:306-483) built from the
    nuScenes CBGS assigner config (6 task groups, 10 classes, 128x128 feature map) and from the Lyft one, on
    synthetic annotations.  Anchors are 12 MB per config, so the golden keeps labels (int8), the positive rows of
    the regression targets and a checksum + samples of the anchors."""
    pp = ref_loader.ref("det3d.datasets.pipelines.preprocess")
    from det3d_amd import Config

    root = os.path.dirname(os.path.dirname(HERE))
    _mg = np.meshgrid
    np.meshgrid = lambda *a, **k: list(_mg(*a, **k))
    out = {}
    try:
        for tag, cfg_name, grid, nd in (("nusc", "nusc_cbgs.py", [1024, 1024, 40], 9), ("lyft", "lyft_cbgs.py", [2016, 2016, 40], 7)):
            cfg = Config.fromfile(os.path.join(root, "det3d_amd", "configs", cfg_name))
            at = pp.AssignTarget(cfg=cfg.assigner)
            names_all = list(cfg.class_names)
            gens = {g.class_name: g for g in cfg.target_assigner.anchor_generators}
            rng = np.random.default_rng(31 if nd == 9 else 32)
            n = 24
            cls = rng.integers(1, len(names_all) + 1, n).astype(np.int32)
            cls[:2] = [1, len(names_all)]
            gt = np.zeros((n, nd), np.float32)
            lim = 45.0 if nd == 9 else 90.0
            gt[:, :2] = rng.uniform(-lim, lim, (n, 2))
            for i in range(n):
                g = gens[names_all[cls[i] - 1]]
                gt[i, 2] = g.anchor_ranges[2]
                gt[i, 3:6] = np.array(g.sizes) * rng.uniform(0.85, 1.2, 3)
            gt[:, -1] = rng.uniform(-6.0, 6.0, n)             # outside [-pi, pi]: exercises limit_period
            if nd == 9:
                gt[:, 6:8] = rng.normal(0, 2, (n, 2))
                gt[3, 6:8] = np.nan
            res = dict(mode="train", lidar=dict(voxels=dict(shape=np.array(grid)),
                                                annotations=dict(gt_boxes=gt.copy(), gt_classes=cls.copy(),
                                                                 gt_names=np.array([names_all[c - 1] for c in cls]))))
            res, _ = at(res, {})
            ex = res["lidar"]["targets"]
            out[tag + "_gt"], out[tag + "_cls"], out[tag + "_grid"] = gt, cls, np.array(grid)
            out[tag + "_ntask"] = np.array(len(ex["anchors"]))
            for t in range(len(ex["anchors"])):
                a, lab, tgt, w = ex["anchors"][t], ex["labels"][t], ex["reg_targets"][t], ex["reg_weights"][t]
                pos = np.where(lab > 0)[0]
                assert np.array_equal(np.where(w > 0)[0], pos) and not tgt[lab <= 0].any()
                out["%s_t%d_labels" % (tag, t)] = lab.astype(np.int8)
                out["%s_t%d_pos_targets" % (tag, t)] = tgt[pos]
                out["%s_t%d_anchor_sum" % (tag, t)] = a.astype(np.float64).sum(0)
                out["%s_t%d_anchor_rows" % (tag, t)] = a[:: max(len(a) // 64, 1)]
                out["%s_t%d_task_gt" % (tag, t)] = res["lidar"]["annotations"]["gt_boxes"][t]
                print("assign_pipeline", tag, t, a.shape, "pos", len(pos), "ignored", int((lab < 0).sum()))
    finally:
        np.meshgrid = _mg
    np.savez_compressed(os.path.join(HERE, "assign_pipeline.npz"), **out)


def range_filter_cases():
    """filter_gt_box_outside_range (core/sampler/preprocess.py:133-147) incl. boxes touching / straddling the border."""
    prep = ref_loader.ref("det3d.core.sampler.preprocess")
    rng = np.random.default_rng(12)
    n = 200
    gt = np.zeros((n, 7), np.float32)
    gt[:, 0] = rng.uniform(-10, 80, n)
    gt[:, 1] = rng.uniform(-50, 50, n)
    gt[:, 3:6] = rng.uniform(0.5, 5, (n, 3))
    gt[:, 6] = rng.uniform(-3.2, 3.2, n)
    gt[:4] = [[0, 0, 0, 2, 2, 1, 0], [-1, 0, 0, 2, 2, 1, 0], [70.4, 40, 0, 2, 4, 1, 0.3], [71.5, 0, 0, 2, 2, 1, 0]]
    lim = np.array([0, -40.0, 70.4, 40.0], np.float32)
    mask = prep.filter_gt_box_outside_range(gt, lim)
    np.savez_compressed(os.path.join(HERE, "range_filter.npz"), gt=gt, limit=lim, mask=mask)
    print("range_filter:", int(mask.sum()), "of", n, mask[:4])


def _dump_specs(path):
    """the in-repo configs are loaded in a CHILD process (they import through det3d_amd's `det3d.*` aliases,
    which must not coexist with ref_loader's real `det3d.*` modules here)."""
    import subprocess

    code = r"""
import pickle, sys
sys.path.insert(0, %r); sys.path.insert(0, %r)
import detector_specs as S
out = {}
for name in S.CASES:
    cfg = S.load_cfg(name)
    m = cfg.model.to_dict()
    coder = cfg.model.bbox_head.box_coder
    m["bbox_head"]["box_coder"] = dict(linear_dim=bool(coder.linear_dim), vec=bool(coder.vec_encode), n_dim=int(coder.n_dim))
    m["neck"].pop("logger", None); m["bbox_head"].pop("logger", None)
    out[name] = dict(model=m, test_cfg=cfg.test_cfg.to_dict())
pickle.dump(out, open(%r, "wb"))
""" % (HERE, os.path.dirname(os.path.dirname(HERE)), path)
    subprocess.run([sys.executable, "-c", code], check=True)


def detector_cases(sane=False):
    """sane=True -> detectors_sane.pt: the same detectors with model_ref.tame_head applied (trained-like head magnitudes),
    eval only: raw predictions and the reference's OWN predict() output, so that the GPU test can compare EVERY
    detection."""
    import pickle
    import tempfile

    import detector_specs as S
    from oracle import clib, model_ref, spconv_shim

    with tempfile.TemporaryDirectory() as td:
        _dump_specs(os.path.join(td, "specs.pkl"))
        specs = pickle.load(open(os.path.join(td, "specs.pkl"), "rb"))
    sys.modules["spconv"] = spconv_shim  # scn.py: `import spconv; from spconv import SparseConv3d, SubMConv3d`
    _patch_nms()
    for mod in ("backbones.scn", "readers.voxel_encoder", "readers.pillar_encoder", "necks.rpn",
                "bbox_heads.mg_head", "losses.losses"):
        ref_loader.ref("det3d.models." + mod)
    assert sys.modules["det3d.models.backbones.scn"].spconv is spconv_shim
    dets_mod = {"VoxelNet": ref_loader.ref("det3d.models.detectors.voxelnet").VoxelNet,
                "PointPillars": ref_loader.ref("det3d.models.detectors.point_pillars").PointPillars}
    coders = ref_loader.ref("det3d.core.bbox.box_coders")

    class Cfg(dict):
        __getattr__ = dict.__getitem__

    out = {}
    for name, case in S.CASES.items():
        spec = specs[name]
        m = spec["model"]
        c = m["bbox_head"]["box_coder"]
        m["bbox_head"]["box_coder"] = coders.GroundBox3dCoderTorch(c["linear_dim"], c["vec"], n_dim=c["n_dim"])
        m["bbox_head"]["logger"] = logging.getLogger("head")
        m["neck"]["logger"] = logging.getLogger("RPN")
        tc = spec["test_cfg"]
        test_cfg = Cfg(tc, nms=Cfg(tc["nms"]))
        kind = m.pop("type")
        m.pop("pretrained", None)
        torch.manual_seed(0)
        net = dets_mod[kind](train_cfg=None, test_cfg=test_cfg, **m)
        model_ref.fill_deterministic(net)
        if sane:
            model_ref.tame_head(net, **S.SANE_HEAD[name])
        code_size = m["bbox_head"]["box_coder"].code_size
        ex, fmap, nv = S.make_example(name, m["bbox_head"]["tasks"], c["n_dim"], code_size)
        anchors, labels, reg_targets = ex["anchors"], ex["labels"], ex["reg_targets"]
        # ---- eval: raw predictions, neck input, detections
        net.eval()
        with torch.no_grad():
            data = dict(features=ex["voxels"], num_voxels=ex["num_points"], coors=ex["coordinates"],
                        batch_size=S.BATCH, input_shape=ex["shape"][0])
            if kind == "PointPillars":
                f = net.reader(data["features"], data["num_voxels"], data["coors"])
            else:
                f = net.reader(data["features"], data["num_voxels"])
            bev = net.backbone(f, data["coors"], data["batch_size"], data["input_shape"])
            preds = net.bbox_head(net.neck(bev))
            dets = net.bbox_head.predict(ex, preds, test_cfg)
        assert preds[0]["box_preds"].shape[1:3] == fmap, (preds[0]["box_preds"].shape, fmap)
        if sane:
            out[name] = dict(preds=[{k: v[:, ::2, 1::2].clone() for k, v in p.items()} for p in preds],
                             dets=[{k: v.clone() for k, v in d.items() if torch.is_tensor(v)} for d in dets])
            print("detector (sane head)", name, "dets", [d["scores"].shape[0] for d in dets], "max |box|",
                  [round(float(d["box3d_lidar"].abs().max()), 2) if d["scores"].shape[0] else 0 for d in dets],
                  "score range", [(round(float(d["scores"].min()), 3), round(float(d["scores"].max()), 3)) if d["scores"].shape[0] else 0 for d in dets])
            continue
        # ---- train: loss per task group, all gradients (norms) + a few whole gradient tensors
        net.train()
        losses = net(ex, return_loss=True)
        total = sum(losses["loss"])
        params = dict(net.named_parameters())
        grads = torch.autograd.grad(total, list(params.values()), allow_unused=True)
        gnorm = {k: (float(g.norm()) if g is not None else 0.0) for k, g in zip(params, grads)}
        small = {k: g.clone() for k, g in zip(params, grads) if g is not None and g.numel() <= 7000}
        # the inputs are NOT stored (12k padded voxels + anchors would be 20 MB): the tests regenerate them with
        # detector_specs.make_example (seeded) and check these sums first
        sums = dict(voxels=float(ex["voxels"].double().sum()), coordinates=int(ex["coordinates"].long().sum()),
                    num_points=int(ex["num_points"].long().sum()), num_voxels=[int(v) for v in nv],
                    anchors=[float(a.double().sum()) for a in anchors], labels=[int(l.long().sum()) for l in labels],
                    reg_targets=[float(t.double().abs().sum()) for t in reg_targets])
        out[name] = dict(
            fingerprint=S.param_fingerprint(net), input_sums=sums,
            bev_shape=tuple(bev.shape), bev_sample=bev[:, :, 1::4, ::8].clone(), bev_abs_mean=float(bev.abs().mean()),
            # raw head outputs [B, H, W, ch] on a quarter of the cells (every cell feeds `dets` anyway)
            preds=[{k: v[:, ::2, 1::2].clone() for k, v in p.items()} for p in preds],
            dets=[{k: v.clone() for k, v in d.items() if torch.is_tensor(v)} for d in dets],
            loss=[l.detach().clone() for l in losses["loss"]], grad_norms=gnorm, grads=small)
        print("detector", name, "voxels", nv, "bev", tuple(bev.shape), "loss", [round(float(l), 4) for l in losses["loss"]],
              "dets", [d["scores"].shape[0] for d in dets], "small grads", len(small))
    torch.save(out, os.path.join(HERE, "detectors_sane.pt" if sane else "detectors.pt"))


def nms_gpu_cases():
    """det3d/ops/nms/nms_gpu.py: iou_device (:22-33, +1 pixel convention), nms_kernel (:67-101, one CUDA block
    per 64x64 tile, `>` threshold), nms_postprocess (:109-126) -- executed as plain Python: the kernel body is
    called once per (block, thread) with a fake numba.cuda context."""
    ng = ref_loader.ref("det3d.ops.nms.nms_gpu")
    rng = np.random.default_rng(11)
    out = {}
    for name, n, thr in (("dense", 150, 0.5), ("sparse", 130, 0.3), ("tiny", 3, 0.5), ("dupes", 70, 0.7)):
        xy = rng.uniform(0, 60 if name != "sparse" else 400, (n, 2))
        wh = rng.uniform(4, 30, (n, 2))
        dets = np.concatenate([xy, xy + wh, rng.uniform(0, 1, (n, 1))], 1).astype(np.float32)
        if name == "dupes":
            dets[n // 2:, :4] = dets[: n - n // 2, :4]  # exact duplicates and
            dets[n // 2:: 3, 2] += 1.0                     # near duplicates
        if name == "tiny":
            dets = np.array([[0, 0, 9, 9, 0.9], [1, 1, 10, 10, 0.8], [20, 20, 29, 29, 0.7]], np.float32)
        order = dets[:, 4].argsort()[::-1]
        boxes = np.ascontiguousarray(dets[order])
        nb = -(-n // 64)
        mask = np.zeros((n * nb,), dtype=np.uint64)
        cuda = sys.modules["numba.cuda"]

        class _Idx:
            pass

        for by in range(nb):
            for bx in range(nb):
                shared = {}
                cuda.shared = type("S", (), {"array": staticmethod(
                    lambda shape, dtype=None, _s=shared: _s.setdefault("a", np.zeros(shape, np.float32)))})
                cuda.syncthreads = lambda: None
                # phase 1 for all threads (shared-memory fill), then phase 2: the kernel has one barrier, so
                # running every thread to completion in order with a persistent shared array is equivalent
                for tx in range(64):
                    cuda.threadIdx = _Idx(); cuda.threadIdx.x = tx
                    cuda.blockIdx = _Idx(); cuda.blockIdx.x = bx; cuda.blockIdx.y = by
                    ng.nms_kernel(n, np.float32(thr), boxes.reshape(-1), mask)
                for tx in range(64):
                    cuda.threadIdx.x = tx
                    ng.nms_kernel(n, np.float32(thr), boxes.reshape(-1), mask)
        keep = np.zeros((n,), np.int32)
        num = ng.nms_postprocess(keep, mask, n)
        out[name + "_dets"] = dets
        out[name + "_thr"] = np.float32(thr)
        out[name + "_keep"] = order[keep[:num]].astype(np.int64)
        print("nms_gpu", name, n, "->", num)
    np.savez_compressed(os.path.join(HERE, "nms_gpu.npz"), **out)


def main():
    ref_loader.install()
    which = sys.argv[1:] or ["voxel", "iou", "ioueval", "vfe", "rpn", "head", "anchors", "tasks", "augment", "objnoise", "gtsampler", "loading", "annotations", "preprocess", "sampler", "collate", "pipeline", "rangefilter", "nmsgpu", "detectors", "detectors_sane"]
    if "anchors" in which:
        anchors_targets()
    if "tasks" in which:
        task_targets()
    if "augment" in which:
        augment_cases()
    if "objnoise" in which:
        object_noise_cases()
    if "gtsampler" in which:
        gt_sampler_cases()
    if "loading" in which:
        loading_cases()
    if "annotations" in which:
        annotation_cases()
    if "preprocess" in which:
        preprocess_cases()
    if "sampler" in which:
        sampler_cases()
    if "collate" in which:
        collate_cases()
    if "pipeline" in which:
        assign_pipeline()
    if "rangefilter" in which:
        range_filter_cases()
    if "voxel" in which:
        voxel_cases()
    if "iou" in which:
        rotate_iou_cases()
    if "ioueval" in which:
        rotate_iou_eval_cases()
    if "vfe" in which:
        vfe_pfn()
    if "rpn" in which:
        rpn_cases()
    if "head" in which:
        head_cases()
    if "nmsgpu" in which:
        nms_gpu_cases()
    if "detectors" in which:
        detector_cases()
    if "detectors_sane" in which:
        detector_cases(sane=True)


if __name__ == "__main__":
    main()
