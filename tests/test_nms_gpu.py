"""HIP rotated NMS / rotated IoU / axis-aligned NMS vs the CPU oracle."""
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN

pytestmark = pytest.mark.gpu


def _dets(rng, n, spread=30.0):
    return np.stack([rng.uniform(0, spread, n), rng.uniform(-spread / 2, spread / 2, n),
                     rng.uniform(1.4, 2.0, n), rng.uniform(3.4, 4.6, n), rng.uniform(-3.2, 3.2, n),
                     rng.permutation(n).astype(np.float64) / n + 0.001], 1).astype(np.float32)


def test_rotate_iou_vs_oracle_and_reference_golden(oracle_lib):
    from det3d_amd.box_ops import rotate_iou

    z = np.load(os.path.join(GOLDEN, "rotate_iou.npz"))
    a, b = torch.from_numpy(z["a"]).cuda(), torch.from_numpy(z["b"]).cuda()
    iou = rotate_iou(a, b).cpu().numpy()
    assert np.abs(np.diag(iou) - z["iou"]).max() < 2e-4   # reference numba-CUDA algorithm
    assert np.abs(iou - oracle_lib.rotate_iou(z["a"], z["b"])).max() < 2e-5


@pytest.mark.parametrize("n,thr,post", [(1, 0.01, 100), (63, 0.5, 100), (64, 0.01, 100), (65, 0.2, 83),
                                        (500, 0.01, 100), (1000, 0.2, 83), (1000, 0.5, 300), (1000, 0.01, 5)])
def test_rotate_nms_keep_set_equals_oracle(oracle_lib, n, thr, post):
    from det3d_amd.box_ops import rotate_nms

    rng = np.random.default_rng(n + int(thr * 100))
    dets = _dets(rng, n)
    ref, margin = oracle_lib.rotate_nms_cc(dets, thr, return_margin=True)
    ref = ref[:post]
    boxes, scores = torch.from_numpy(dets[:, :5]).cuda(), torch.from_numpy(dets[:, 5]).cuda()
    keep = rotate_nms(boxes, scores, pre_max_size=1000, post_max_size=post, iou_threshold=thr)
    assert keep.dtype == torch.int64 and keep.device.type == "cuda"
    if margin < 1e-5:
        pytest.skip("threshold tie within 1e-5 in this draw (SURVEY 8c)")
    assert keep.cpu().tolist() == ref


def test_rotate_nms_pre_max_and_empty():
    from det3d_amd.box_ops import rotate_nms

    e = rotate_nms(torch.zeros((0, 5)).cuda(), torch.zeros((0,)).cuda(), 1000, 100, 0.5)
    assert e.shape == (0,) and e.dtype == torch.int64
    # far apart boxes: nothing suppressed, pre_max truncates to the top scores
    n = 50
    boxes = torch.zeros((n, 5)).cuda()
    boxes[:, 0] = torch.arange(n).cuda() * 10.0
    boxes[:, 2:4] = 1.0
    scores = torch.rand(n).cuda()
    keep = rotate_nms(boxes, scores, pre_max_size=10, post_max_size=100, iou_threshold=0.1)
    assert keep.cpu().tolist() == scores.topk(10)[1].cpu().tolist()


def test_rotate_nms_adversarial(oracle_lib):
    from det3d_amd.box_ops import rotate_nms

    base = np.array([[10, 0, 1.6, 3.9, 0.3]], np.float32)
    boxes = np.concatenate([
        base, base,                                  # duplicate
        base + [0, 0, 0, 0, np.pi / 2],              # 90 degrees
        base * [1, 1, .5, .5, 1],                    # nested
        base + [1.6, 0, 0, 0, -0.3],                 # near touching
        base + [40, 0, 0, 0, 0]]).astype(np.float32)  # disjoint
    scores = np.linspace(0.9, 0.4, len(boxes)).astype(np.float32)
    dets = np.concatenate([boxes, scores[:, None]], 1)
    for thr in (0.05, 0.3, 0.7):
        ref = oracle_lib.rotate_nms_cc(dets, thr)
        keep = rotate_nms(torch.from_numpy(boxes).cuda(), torch.from_numpy(scores).cuda(), 1000, 100, thr)
        assert keep.cpu().tolist() == ref


def test_rotate_nms_and_iou3d_on_duplicated_and_nearly_duplicated_boxes(oracle_lib):
    """clip-scratch overflow regression: many exact duplicates and boxes that differ by a few ulps /
    micro-rotations -- near-collinear edges make Sutherland-Hodgman emit extra vertices after rounding."""
    from det3d_amd.box_ops import rotate_nms
    from det3d_amd import iou3d_utils

    rng = np.random.default_rng(3)
    base = np.concatenate([rng.uniform(-20, 20, (40, 2)), rng.uniform(0.5, 5, (40, 2)), rng.uniform(-3.2, 3.2, (40, 1))], 1)
    rep = np.repeat(base, 6, axis=0)
    rep[1::6] = np.nextafter(rep[1::6].astype(np.float32), np.float32(np.inf))       # 1 ulp off
    rep[2::6, 4] += 1e-6                                                             # micro rotation
    rep[3::6, :2] += 1e-5                                                            # micro shift
    rep[4::6, 4] += np.pi / 2 + 1e-6                                                 # ~90 degrees, same centre
    rep[5::6, 2:4] *= 1 + 1e-6                                                       # micro scale
    boxes = rep.astype(np.float32)
    scores = (rng.permutation(len(boxes)) / len(boxes)).astype(np.float32)
    dets = np.concatenate([boxes, scores[:, None]], 1)
    for thr in (0.1, 0.5, 0.9):
        (ref, margin) = oracle_lib.rotate_nms_cc(dets, thr, return_margin=True)
        keep = rotate_nms(torch.from_numpy(boxes).cuda(), torch.from_numpy(scores).cuda(), 1000, 1000, thr).cpu().tolist()
        if keep != ref:
            assert margin < 1e-5, (thr, margin)  # an IoU sat within rounding of the threshold
    # iou3d overlap on the same set: finite, symmetric-ish and equal to the oracle
    b5 = np.stack([boxes[:, 0] - boxes[:, 2] / 2, boxes[:, 1] - boxes[:, 3] / 2, boxes[:, 0] + boxes[:, 2] / 2,
                   boxes[:, 1] + boxes[:, 3] / 2, boxes[:, 4]], 1).astype(np.float32)
    got = iou3d_utils.boxes_iou_bev(torch.from_numpy(b5).cuda(), torch.from_numpy(b5).cuda()).cpu().numpy()
    ref = oracle_lib.iou3d_pairwise(b5, b5, "iou_bev")
    assert np.isfinite(got).all() and np.abs(got - ref).max() < 1e-4


def test_axis_aligned_nms(oracle_lib):
    from det3d_amd.box_ops import nms

    rng = np.random.default_rng(5)
    n = 700
    xy = rng.uniform(0, 200, (n, 2))
    wh = rng.uniform(5, 40, (n, 2))
    boxes = np.concatenate([xy, xy + wh], 1).astype(np.float32)
    scores = (rng.permutation(n) / n).astype(np.float32)
    order = scores.argsort()[::-1]
    ref = [int(order[i]) for i in oracle_lib.nms_sorted(boxes[order], 0.4)][:120]
    keep = nms(torch.from_numpy(boxes).cuda(), torch.from_numpy(scores).cuda(), 1000, 120, 0.4)
    assert keep.cpu().tolist() == ref


@pytest.mark.parametrize("case", ["dense", "sparse", "tiny", "dupes"])
def test_axis_aligned_nms_vs_reference_nms_gpu_golden(case):
    """a12: d3d_nms (through box_ops.nms) == the reference's nms_gpu.py run as plain Python (nms_gpu.npz)."""
    from det3d_amd.box_ops import nms

    g = np.load(os.path.join(GOLDEN, "nms_gpu.npz"))
    dets, thr, ref = g[case + "_dets"], float(g[case + "_thr"]), g[case + "_keep"]
    n = len(dets)
    keep = nms(torch.from_numpy(dets[:, :4]).cuda(), torch.from_numpy(dets[:, 4]).cuda(), n, n, thr)
    # score ties (the duplicated boxes keep distinct random scores) do not occur: exact order equality
    assert keep.cpu().tolist() == ref.tolist()


@pytest.mark.parametrize("fused", [True, False])
@pytest.mark.parametrize("name", ["kitti", "nusc", "lyft"])
def test_head_predict_matches_reference_golden(name, fused):
    """MultiGroupHead.predict on the GPU (decode, score filter, top-k, HIP rotated NMS, direction
    flip, range mask, multi-task merge) vs the detections the REFERENCE's predict() produced on the
    same predictions (tests/golden/head.pt; its NMS step used this repo's oracle, see make_golden.py)."""
    import logging

    from det3d_amd import bbox_heads, builder
    from det3d_amd.bbox_heads import MultiGroupHead
    from det3d_amd.config import ConfigDict

    # fused = d3d_decode_filter_topk + d3d_rotate_nms_batched for the whole batch; unfused = per-frame torch ops
    bbox_heads.FUSE_PREDICT = fused
    g = torch.load(os.path.join(GOLDEN, "head.pt"), weights_only=False)[name]
    coder = builder.GroundBox3dCoderTorch(g["coder"]["linear_dim"], g["coder"]["vec"], n_dim=g["coder"]["n_dim"])
    head = MultiGroupHead(
        mode="3d", in_channels=g["in_ch"], norm_cfg=None, tasks=g["tasks"], weights=[1] * len(g["tasks"]),
        box_coder=coder, encode_background_as_zeros=True,
        loss_norm=dict(type="NormByNumPositives", pos_cls_weight=1.0, neg_cls_weight=2.0),
        loss_cls=dict(type="SigmoidFocalLoss", alpha=0.25, gamma=2.0, loss_weight=1.0), use_sigmoid_score=True,
        loss_bbox=dict(type="WeightedSmoothL1Loss", sigma=3.0, code_weights=[1.0] * coder.code_size,
                       codewise=True, loss_weight=2.0),
        encode_rad_error_by_sin=not coder.vec_encode, loss_aux=g["loss_aux"],
        direction_offset=g["direction_offset"], logger=logging.getLogger("head"))
    head.load_state_dict(g["state"])
    head = head.cuda().eval()
    example = dict(anchors=[a.cuda() for a in g["anchors"]], metadata=[dict(token=i) for i in range(2)])
    try:
        with torch.no_grad():
            preds = head(g["x"].cuda())
            assert head._fused_predict_ok(example, preds, ConfigDict(g["test_cfg"])) == fused
            dets = head.predict(example, preds, ConfigDict(g["test_cfg"]))
    finally:
        bbox_heads.FUSE_PREDICT = True
    assert len(dets) == len(g["dets"])
    for d, r in zip(dets, g["dets"]):
        assert d["box3d_lidar"].shape == r["box3d_lidar"].shape
        assert torch.allclose(d["box3d_lidar"].cpu(), r["box3d_lidar"], atol=1e-4)
        assert torch.allclose(d["scores"].cpu(), r["scores"], atol=1e-5)
        assert torch.equal(d["label_preds"].cpu(), r["label_preds"])
        assert d["metadata"] == r["metadata"]


@pytest.mark.parametrize("nd,vec,use_dir,nc,dtype", [(7, False, True, 1, torch.float32), (9, True, False, 2, torch.bfloat16),
                                                     (7, False, True, 2, torch.bfloat16), (9, True, False, 1, torch.float32)])
def test_decode_filter_topk_vs_torch_formulation(nd, vec, use_dir, nc, dtype):
    """d3d_decode_filter_topk against sigmoid -> max over classes -> threshold -> sort (score desc, anchor asc)
    -> top-k -> second_box_decode in torch ops, incl. massive score ties (empty cells of a random-init head all
    share one logit) and a frame with fewer candidates than K."""
    from det3d_amd import box_ops

    torch.manual_seed(nd + nc)
    B, A, K = 3, 20000, 1000
    code = nd + (1 if vec else 0)
    dev = torch.device("cuda")
    box = (torch.randn(B, A, code, device=dev) * 0.3).to(dtype)
    cls = torch.randn(B, A, nc, device=dev) * 2 - 1.0
    cls[0, 5000:, :] = 0.25          # 15000 anchors with identical scores above the threshold: ties at the cut
    cls[2] = cls[2] - 6.0            # few candidates
    cls = cls.to(dtype)
    dirp = torch.randn(B, A, 2, device=dev).to(dtype) if use_dir else None
    anchors = torch.zeros(A, nd, device=dev)
    anchors[:, :2] = torch.rand(A, 2, device=dev) * 80 - 40
    anchors[:, 2] = -1.0
    anchors[:, 3:6] = torch.tensor([1.6, 3.9, 1.56], device=dev) * (1 + 0.2 * torch.rand(A, 3, device=dev))
    anchors[:, -1] = torch.rand(A, device=dev) * 3.14
    thr = 0.3
    boxes, scores, labels, dirs, count = box_ops.decode_filter_topk(box, cls, dirp, anchors, vec, thr, K)
    sc_all, lab_all = torch.sigmoid(cls.float()).max(-1)
    for b in range(B):
        cand = (sc_all[b] >= thr).nonzero()[:, 0]
        s_sorted, order = torch.sort(sc_all[b][cand], descending=True, stable=True)  # ties keep ascending anchors
        sel = cand[order][:K]
        n = int(count[b])
        assert n == min(len(cand), K)
        assert torch.equal(scores[b, :n], s_sorted[:n])
        assert torch.equal(labels[b, :n].long(), lab_all[b][sel])
        ref = box_ops.second_box_decode(box[b, sel].float(), anchors[sel], vec, False)
        assert torch.allclose(boxes[b, :n], ref, atol=1e-5, rtol=1e-5)
        if use_dir:
            assert torch.equal(dirs[b, :n].long(), dirp[b, sel].float().max(-1)[1])
    assert int(count[2]) < K and int(count[0]) == K


def test_rotate_nms_batched_equals_the_single_list_kernel():
    from det3d_amd import box_ops

    rng = np.random.default_rng(8)
    S, K, nd = 5, 600, 9
    counts = np.array([600, 0, 37, 129, 512], np.int32)
    boxes = np.zeros((S, K, nd), np.float32)
    boxes[..., :2] = rng.uniform(-30, 30, (S, K, 2))
    boxes[..., 3:5] = rng.uniform(1, 6, (S, K, 2))
    boxes[..., -1] = rng.uniform(-3.1, 3.1, (S, K))
    bt = torch.from_numpy(boxes).cuda()
    keep, nkeep = box_ops.rotate_nms_batched(bt, torch.from_numpy(counts).cuda(), 0.2, 83)
    for s in range(S):
        n = int(counts[s])
        b5 = bt[s, :n][:, [0, 1, 3, 4, -1]]
        sc = torch.arange(n, 0, -1, device="cuda").float()  # already sorted: descending scores
        ref = box_ops.rotate_nms(b5, sc, 1000, 83, 0.2).cpu().tolist() if n else []
        got = keep[s, : int(nkeep[s])].cpu().tolist()
        assert got == ref and (keep[s, int(nkeep[s]):] == -1).all()
