"""End-to-end parity of the HIP path (det3d_amd VoxelNet on the GPU) against the CPU oracle
model (oracle/model_ref.SecondRef, pinned to the reference by goldens) on the same synthetic
cloud with the same weights.  fp32 mode: box regressions within 1e-3 (north_star tolerance)."""
import os

import numpy as np
import pytest
import torch

from conftest import ROOT

pytestmark = pytest.mark.gpu


def _setup(n_points=6000, batch=2, seed=0):
    from det3d_amd import Config, build_detector, synthetic
    from oracle import model_ref as M

    cfg = Config.fromfile(os.path.join(ROOT, "det3d_amd", "configs", "kitti_car_second.py"))
    torch.manual_seed(seed)
    prod = build_detector(cfg.model, cfg.train_cfg, cfg.test_cfg)
    # non-trivial BN statistics / head biases so eval-mode outputs are informative
    for m in prod.modules():
        if isinstance(m, (torch.nn.BatchNorm1d, torch.nn.BatchNorm2d)):
            m.running_mean.normal_(0, 0.1)
            m.running_var.uniform_(0.5, 1.5)
    torch.nn.init.constant_(prod.bbox_head.tasks[0].conv_cls.bias, 0.5)
    ref = M.SecondRef(cfg)
    ref.load_state_dict(prod.state_dict())
    ds = synthetic.KittiCarSynthetic(n_points=n_points, n_seeds=batch)
    return cfg, prod.cuda(), ref, ds


def test_forward_loss_and_grads_fp32():
    from det3d_amd import synthetic
    from oracle import model_ref as M

    cfg, prod, ref, ds = _setup()
    prod.train()
    ref.train()
    hb = ds.host_batch(0, 2)
    ex_cpu = M.cpu_example(hb, synthetic.KITTI_CAR)
    ex_gpu = synthetic.make_example(ds.device_batch(0, 2, torch.device("cuda")), synthetic.KITTI_CAR,
                                    fused_mean=True, want_voxels=True)
    # the data path: device voxelizer output == reference CPU voxelizer output, bit exact
    assert np.array_equal(ex_gpu["coordinates"].cpu().numpy(), ex_cpu["coordinates"].numpy())
    assert np.array_equal(ex_gpu["voxels"].cpu().numpy(), ex_cpu["voxels"].numpy())
    loss_ref = sum(ref(ex_cpu)["loss"])
    loss_ref.backward()
    losses = prod(ex_gpu, return_loss=True)
    loss = sum(losses["loss"])
    loss.backward()
    assert float(loss) == pytest.approx(float(loss_ref), rel=2e-4)
    worst = []
    for (n1, p1), (n2, p2) in zip(prod.named_parameters(), ref.named_parameters()):
        assert n1 == n2
        g1, g2 = p1.grad.cpu(), p2.grad
        rel = float((g1 - g2).abs().max()) / (float(g2.abs().max()) + 1e-12)
        cos = float(torch.nn.functional.cosine_similarity(g1.flatten(), g2.flatten(), dim=0))
        worst.append((rel, 1.0 - cos, n1))
    worst.sort(reverse=True)
    print("worst grad rel errors:", [(round(r, 5), n) for r, _, n in worst[:6]])
    # two fp32 implementations through 20 train-mode BN + ReLU layers: summation order differs and
    # near-zero pre-activations flip ReLU gates, so element-wise agreement is bounded at 5 % of each
    # tensor's max while the direction of every gradient tensor must agree to 1e-3
    assert worst[0][0] < 5e-2, worst[:6]
    assert max(c for _, c, _ in worst) < 1e-3, sorted(worst, key=lambda t: -t[1])[:6]


def test_box_regressions_within_1e3_fp32():
    from det3d_amd import synthetic
    from oracle import model_ref as M

    cfg, prod, ref, ds = _setup(seed=1)
    prod.eval()
    ref.eval()
    hb = ds.host_batch(0, 2)
    ex_cpu = M.cpu_example(hb, synthetic.KITTI_CAR)
    ex_gpu = synthetic.make_example(ds.device_batch(0, 2, torch.device("cuda")), synthetic.KITTI_CAR)
    with torch.no_grad():
        x = ref.features(ex_cpu)
        preds_ref = [t(x) for t in ref.bbox_head.tasks]
        data = dict(features=ex_gpu["voxel_mean"], num_voxels=ex_gpu["num_points"], coors=ex_gpu["coordinates"],
                    batch_size=2, input_shape=ex_gpu["shape"][0])
        preds = prod.bbox_head(prod.extract_feat(data))
    for k in ("box_preds", "cls_preds", "dir_cls_preds"):
        a, b = preds[0][k].cpu(), preds_ref[0][k]
        assert float((a - b).abs().max()) < 1e-3, k
    # detections: NMS is discontinuous, so feed the oracle's decode+NMS the GPU's own raw
    # predictions (equal to the oracle's within 1e-3, checked above).  A random-init net gives
    # identical logits on all empty cells and tie order is implementation-defined in the
    # reference (topk, then a non-stable argsort: box_torch_ops.py:534, nms_cpu.py:36), so break
    # ties with the same fixed jitter on both sides.
    with torch.no_grad():
        g = torch.Generator().manual_seed(0)
        jitter = torch.rand(preds[0]["cls_preds"].shape, generator=g) * 0.5
        preds[0]["cls_preds"] = preds[0]["cls_preds"] + jitter.cuda()
        dets = prod.bbox_head.predict(ex_gpu, preds, prod.test_cfg)
        preds_cpu = [{k: v.float().cpu() for k, v in preds[0].items()}]
        dets_ref = ref.predict(ex_cpu, preds_cpu)
    for d, r in zip(dets, dets_ref):
        a, b = d["box3d_lidar"].cpu(), r["box3d_lidar"]
        assert a.shape[0] > 0 and abs(a.shape[0] - b.shape[0]) <= 1
        # match by nearest centre: every GPU box has an oracle twin within 1e-3, allowing one
        # borderline NMS decision (IoU threshold 0.01 sits on touching boxes)
        dist = torch.cdist(a[:, :2], b[:, :2])
        nn = dist.argmin(1)
        err = (a - b[nn]).abs().max(1)[0]
        assert int((err >= 1e-3).sum()) <= 1, err
        assert float((d["scores"].cpu() - r["scores"][nn]).abs()[err < 1e-3].max()) < 1e-4


def test_bf16_budget_is_bounded_layer_by_layer():
    """bf16 autocast (BASELINE configs[1] names bf16) cannot meet north_star's 1e-3 on O(1) box regressions --
    bf16 activations carry 2^-9 relative rounding per layer -- so the contract is carried by the fp32 mode
    (test_box_regressions_within_1e3_fp32, the detector goldens, bench.py's `fp32` line) and this test pins the
    bf16 deviation instead of waving it through: every sparse layer's bf16 output against the SAME
    HIP path in fp32 (relative RMS <= 2 %, growing slowly with depth) and the raw box regressions against the CPU
    oracle (<= 2 % of the largest regression)."""
    from det3d_amd import spconv, synthetic
    from oracle import model_ref as M

    cfg, prod, ref, ds = _setup(seed=2)
    prod.eval()
    ref.eval()
    ex_cpu = M.cpu_example(ds.host_batch(0, 2), synthetic.KITTI_CAR)
    ex_gpu = synthetic.make_example(ds.device_batch(0, 2, torch.device("cuda")), synthetic.KITTI_CAR)
    data = dict(features=ex_gpu["voxel_mean"], num_voxels=ex_gpu["num_points"], coors=ex_gpu["coordinates"],
                batch_size=2, input_shape=ex_gpu["shape"][0])
    taps = {}

    def tap(tag):
        def hook(mod, inp, out):
            taps.setdefault(tag, []).append(out.features.detach().float())
        return hook

    hooks = [m.register_forward_hook(tap(i)) for i, m in enumerate(prod.backbone.middle_conv.conv_layers())]
    with torch.no_grad():
        x = ref.features(ex_cpu)
        ref_box = ref.bbox_head.tasks[0](x)["box_preds"]
        box32 = prod.bbox_head(prod.extract_feat(data))[0]["box_preds"].float().cpu()
        with torch.autocast("cuda", dtype=torch.bfloat16):
            box = prod.bbox_head(prod.extract_feat(data))[0]["box_preds"].float().cpu()
    for h in hooks:
        h.remove()
    rel = []
    for i in sorted(taps):
        a32, a16 = taps[i]
        rel.append(float((a16 - a32).norm() / (a32.norm() + 1e-12)))
    print("bf16 vs fp32 HIP path, relative RMS per sparse conv layer:", [round(r, 4) for r in rel])
    assert max(rel) < 2e-2, rel
    err = float((box - ref_box).abs().max())
    err32 = float((box32 - ref_box).abs().max())
    scale = float(ref_box.abs().max())
    print("box_preds max abs err: bf16 %.4f, fp32 %.2e (max |ref| %.3f)" % (err, err32, scale))
    assert err32 < 1e-3 and err < 2e-2 * max(scale, 1e-6)


def test_training_reduces_loss_bf16():
    from det3d_amd import synthetic
    from det3d_amd.trainer import TrainStep

    cfg, prod, ref, ds = _setup(n_points=8000)
    prod.train()
    step = TrainStep(prod, total_steps=40, autocast_dtype=torch.bfloat16)
    batch = ds.device_batch(0, 2, torch.device("cuda"))
    losses = [float(step(synthetic.make_example(batch, synthetic.KITTI_CAR))) for _ in range(12)]
    assert all(np.isfinite(losses)) and losses[-1] < losses[0]


def test_point_pillars_gpu_matches_its_cpu_plumbing_path():
    """BASELINE configs[0] (kitti_point_pillars): PFN -> scatter (HIP dense kernel on GPU, index_put
    on CPU) -> RPN -> head; GPU fp32 vs the same modules on CPU."""
    from det3d_amd import Config, build_detector, synthetic
    from det3d_amd.point_cloud_ops import voxelize_batch, grid_size_of

    cfg = Config.fromfile(os.path.join(ROOT, "det3d_amd", "configs", "kitti_point_pillars.py"))
    torch.manual_seed(0)
    m_cpu = build_detector(cfg.model, cfg.train_cfg, cfg.test_cfg).eval()
    import copy
    m_gpu = copy.deepcopy(m_cpu).cuda().eval()
    vg = cfg.voxel_generator
    pts = [synthetic.kitti_like_cloud(s, 16384)[0] for s in range(2)]
    off = np.cumsum([0] + [len(p) for p in pts]).astype(np.int32)
    vb = voxelize_batch(torch.from_numpy(np.concatenate(pts)).cuda(), torch.from_numpy(off).cuda(),
                        vg.voxel_size, vg.range, vg.max_points_in_voxel, vg.max_voxel_num).sync()
    grid = grid_size_of(vg.voxel_size, vg.range)
    ex = dict(voxels=vb["voxels"], num_points=vb["num_points"], coordinates=vb["coordinates"],
              num_voxels=vb["num_voxels"], shape=np.stack([grid] * 2))
    ex_cpu = {k: (v.cpu() if torch.is_tensor(v) else v) for k, v in ex.items()}
    with torch.no_grad():
        data = lambda e: dict(features=e["voxels"], num_voxels=e["num_points"], coors=e["coordinates"],
                              batch_size=2, input_shape=e["shape"][0])
        a = m_gpu.bbox_head(m_gpu.extract_feat(data(ex)))[0]
        b = m_cpu.bbox_head(m_cpu.extract_feat(data(ex_cpu)))[0]
    assert a["box_preds"].shape == (2, 248, 216, 14)
    for k in a:
        assert float((a[k].cpu() - b[k]).abs().max()) < 1e-3, k


def test_resnet_fhd_backbone_runs_and_is_self_consistent():
    """SpMiddleResNetFHD (nuScenes / Lyft CBGS backbone): planned + fused path vs the unplanned,
    unfused path of the same weights; fwd output and input gradient."""
    import copy

    from det3d_amd import spconv, synthetic
    from det3d_amd.backbones import SpMiddleResNetFHD
    from det3d_amd.point_cloud_ops import voxelize_batch

    torch.manual_seed(0)
    net = SpMiddleResNetFHD(num_input_features=5).cuda().train()
    net2 = copy.deepcopy(net)
    rng = np.random.default_rng(0)
    pts = rng.uniform([-51, -51, -5, 0, 0], [51, 51, 3, 1, 0.5], size=(30000, 5)).astype(np.float32)
    off = torch.tensor([0, 14000, 30000], dtype=torch.int32).cuda()
    vb = voxelize_batch(torch.from_numpy(pts).cuda(), off, [0.1, 0.1, 0.2], [-51.2, -51.2, -5.0, 51.2, 51.2, 3.0],
                        10, 60000, want_voxels=False, mean_channels=5).sync()
    feats = vb["mean"].clone().requires_grad_(True)
    out = net(feats, vb["coordinates"], 2, [1024, 1024, 40])
    assert out.shape == (2, 256, 128, 128)
    g = torch.randn_like(out)
    out.backward(g)
    spconv.FUSE_BN_RELU = False
    try:
        feats2 = vb["mean"].clone().requires_grad_(True)
        net2.middle_conv.plan = lambda x: None  # force the per-layer (lazy) rulebook path
        out2 = net2(feats2, vb["coordinates"], 2, [1024, 1024, 40])
        out2.backward(g)
    finally:
        spconv.FUSE_BN_RELU = True
    assert torch.allclose(out, out2, atol=2e-3, rtol=2e-3)
    # 27 train-mode BN + ReLU layers: compare gradient directions (see test_forward_loss_and_grads)
    cos = torch.nn.functional.cosine_similarity
    worst = [(float(cos(feats.grad.flatten(), feats2.grad.flatten(), dim=0)), "input")]
    gmax = max(float(p.grad.abs().max()) for p in net2.parameters())
    for (n1, p1), (n2, p2) in zip(net.named_parameters(), net2.named_parameters()):
        if float(p2.grad.abs().max()) < 1e-4 * gmax:
            # a conv bias followed by train-mode BN has an exactly-zero true gradient (BN removes
            # the mean): both sides hold rounding noise only
            assert float(p1.grad.abs().max()) < 1e-3 * gmax, n1
            continue
        worst.append((float(cos(p1.grad.flatten().float(), p2.grad.flatten().float(), dim=0)), n1))
    worst.sort()
    print("lowest gradient cosines:", [(round(c, 5), n) for c, n in worst[:5]])
    assert worst[0][0] > 0.99, worst[:5]


@pytest.mark.parametrize("channels", [16, 64, 128])
def test_fused_basic_block_equals_the_per_layer_block(channels):
    """sparse_block._BlockFn (SparseBasicBlock as one autograd node: residual add + ReLU inside the BatchNorm sweeps,
    d(identity) out of bn_bwd_dx_residual, conv1's data gradient accumulating it through D3dConvEpilogue.addend)
    against the per-layer path (the same kernels with PyTorch's add / clamp / gradient accumulation in between),
    f32: the forward bit for bit, running statistics bit for bit, every gradient to rounding (the backward's ReLU mask
    is recomputed from the pre-activation in one, read from the output in the other), two identical runs bit-equal."""
    import copy

    from det3d_amd import sparse_block, spconv
    from det3d_amd.backbones import SparseBasicBlock

    torch.manual_seed(channels)
    dev = torch.device("cuda", 0)
    rng = np.random.default_rng(1)
    cells = np.unique(rng.integers(0, [2, 12, 40, 40], size=(9000, 4)), axis=0).astype(np.int32)
    idx = torch.from_numpy(cells).to(dev)
    blk = SparseBasicBlock(channels, channels, indice_key="res").to(dev).train()
    with torch.no_grad():
        for bn in (blk.bn1, blk.bn2):
            bn.weight.uniform_(0.5, 1.5)
            bn.bias.normal_(0, 0.3)
    ref = copy.deepcopy(blk)
    twin = copy.deepcopy(blk)
    feats0 = torch.randn((len(cells), channels), device=dev)
    w = torch.randn((len(cells), channels), device=dev)

    def run(m, fused):
        prev = sparse_block.FUSE_BLOCK
        sparse_block.FUSE_BLOCK = fused
        try:
            f = feats0.clone().requires_grad_(True)
            x = spconv.SparseConvTensor(f, idx, [12, 40, 40], 2)
            spconv.SparseSequential(m).plan(x)
            spconv.ARENA.reset(dev)
            y = m(x).features
            (y * w).sum().backward()
            return y.detach(), f.grad, {k: p.grad.clone() for k, p in m.named_parameters()}
        finally:
            sparse_block.FUSE_BLOCK = prev

    ya, ga, pa = run(blk, True)
    yb, gb, pb = run(ref, False)
    yc, gc, pc = run(twin, True)
    assert torch.equal(ya, yc) and torch.equal(ga, gc) and all(torch.equal(pa[k], pc[k]) for k in pa)
    assert torch.equal(ya, yb)
    for k in ("running_mean", "running_var"):
        assert torch.equal(getattr(blk.bn1, k), getattr(ref.bn1, k)) and torch.equal(getattr(blk.bn2, k), getattr(ref.bn2, k))
    rel = lambda a, b: float((a - b).norm()) / (float(b.norm()) + 1e-30)
    assert rel(ga, gb) < 1e-5, rel(ga, gb)
    gmax = max(float(v.abs().max()) for v in pb.values())
    for k in pa:
        if k.endswith("conv1.bias") or k.endswith("conv2.bias"):
            # exactly zero here, summation noise there: the batch mean of the BatchNorm behind it removes the bias
            assert float(pa[k].abs().max()) == 0.0 and float(pb[k].abs().max()) < 1e-3 * gmax, k
        else:
            assert rel(pa[k], pb[k]) < 1e-5, (k, rel(pa[k], pb[k]))


def _nusc_example(cfg, batch, n_points, seed, dev):
    """voxelized synthetic nuScenes-like cloud + device-assigned targets for every task group of `cfg`."""
    from det3d_amd import synthetic
    from det3d_amd.point_cloud_ops import grid_size_of, voxelize_batch

    rng = np.random.default_rng(seed)
    vg = cfg.voxel_generator
    lo, hi = np.array(vg.range[:3]), np.array(vg.range[3:])
    pts = np.concatenate([rng.uniform(lo, hi, (batch * n_points, 3)), rng.uniform(0, 1, (batch * n_points, 1)),
                          rng.integers(0, 10, (batch * n_points, 1)) * 0.05], 1).astype(np.float32)
    pts[:, 2] = rng.normal(-1.5, 0.6, len(pts)).clip(lo[2] + 0.01, hi[2] - 0.01)  # a ground-ish slab
    off = torch.arange(batch + 1, dtype=torch.int32, device=dev) * n_points
    out = voxelize_batch(torch.from_numpy(pts).to(dev), off, vg.voxel_size, vg.range, vg.max_points_in_voxel,
                         vg.max_voxel_num, want_voxels=False, mean_channels=5).sync()
    grid = grid_size_of(vg.voxel_size, vg.range)
    f = cfg.assigner.out_size_factor
    fmap = [1, int(grid[1]) // f, int(grid[0]) // f]
    vec = bool(cfg.box_coder.encode_angle_vector)
    ex = dict(coordinates=out["coordinates"], num_points=out["num_points"], num_voxels=out["num_voxels"],
              voxel_mean=out["mean"], shape=np.stack([grid] * batch), metadata=[dict(token=i) for i in range(batch)],
              anchors=[], labels=[], reg_targets=[], reg_weights=[])
    gens_all = {g.class_name: g for g in cfg.target_assigner.anchor_generators}
    for task in cfg.tasks:
        gens = [gens_all[c].to_dict() if hasattr(gens_all[c], "to_dict") else dict(gens_all[c]) for c in task.class_names]
        anchors, k, r, hi_t, lo_t = synthetic.task_anchors(fmap, gens)
        nd = anchors.shape[1]
        gmax = 6
        gt = np.zeros((batch, gmax, nd), np.float32)
        cls = np.zeros((batch, gmax), np.int32)
        cnt = rng.integers(0, gmax + 1, batch).astype(np.int32)
        for b in range(batch):
            for j in range(cnt[b]):
                c = int(rng.integers(0, k))
                cls[b, j] = c + 1
                gt[b, j, :2] = rng.uniform(-40, 40, 2)
                gt[b, j, 2] = gens[c]["anchor_ranges"][2]
                gt[b, j, 3:6] = np.array(gens[c]["sizes"]) * rng.uniform(0.9, 1.1, 3)
                gt[b, j, 6:8] = rng.normal(0, 2, 2)
                gt[b, j, 8] = rng.uniform(-3.1, 3.1)
        a = torch.from_numpy(anchors).to(dev)
        lab, tgt, w = synthetic.assign_targets_task_device(a, k, r, hi_t, lo_t, torch.from_numpy(gt).to(dev),
                                                           torch.from_numpy(cls).to(dev), torch.from_numpy(cnt).to(dev),
                                                           encode_angle_vector=vec, limit_gt_angle=True)
        ex["anchors"].append(a[None].expand(batch, -1, -1).contiguous())
        ex["labels"].append(lab)
        ex["reg_targets"].append(tgt)
        ex["reg_weights"].append(w)
    return ex


def test_nuscenes_cbgs_six_group_model_trains_and_predicts():
    """BASELINE.json configs[3] (nusc CBGS: VFEv3 -> SpMiddleResNetFHD -> RPN -> 6 task groups, 9-dim boxes with
    the angle-vector coder) from det3d_amd/configs/nusc_cbgs.py: device voxelizer, device AssignTarget per task
    group, one bf16 training step through TrainStep, then predict() with the rotated NMS."""
    from det3d_amd import Config, build_detector
    from det3d_amd.trainer import TrainStep

    here = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cfg = Config.fromfile(os.path.join(here, "det3d_amd", "configs", "nusc_cbgs.py"))
    dev = torch.device("cuda")
    torch.manual_seed(0)
    model = build_detector(cfg.model, cfg.train_cfg, cfg.test_cfg).to(dev).train()
    ex = _nusc_example(cfg, batch=2, n_points=40000, seed=0, dev=dev)
    assert [a.shape[1] for a in ex["anchors"]] == [128 * 128 * 2 * t.num_class for t in cfg.tasks]
    assert sum(int((l > 0).sum()) for l in ex["labels"]) > 0
    step = TrainStep(model, total_steps=10, lr_max=cfg.lr_config.lr_max, autocast_dtype=torch.bfloat16)
    l0 = float(step(ex))
    l1 = float(step(ex))
    assert np.isfinite(l0) and np.isfinite(l1) and l1 < l0 * 1.5
    assert all(torch.isfinite(p.grad).all() for p in model.parameters() if p.grad is not None)
    assert len(step.last["losses"]["loss"]) == 6
    model.eval()
    with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16):
        dets = model(ex, return_loss=False)
    assert len(dets) == 2
    for d in dets:
        n = d["box3d_lidar"].shape[0]
        assert d["box3d_lidar"].shape == (n, 9) and d["scores"].shape == (n,) and n <= 6 * cfg.test_cfg.nms.nms_post_max_size
        if n:
            assert int(d["label_preds"].max()) < 10 and float(d["scores"].min()) >= cfg.test_cfg.score_threshold


# ------------------------------------------------ whole detectors vs the REFERENCE's own detector classes
def _golden_case(name):
    import sys

    gdir = os.path.join(ROOT, "tests", "golden")
    if gdir not in sys.path:
        sys.path.insert(0, gdir)
    import detector_specs as S

    g = torch.load(os.path.join(gdir, "detectors.pt"), weights_only=False)[name]
    cfg = S.load_cfg(name)
    coder = cfg.model.bbox_head.box_coder
    ex, fmap, nv = S.make_example(name, cfg.model.bbox_head.tasks, coder.n_dim, coder.code_size)
    S.check_input_sums(ex, g["input_sums"])
    return S, g, cfg, ex


def _bias_before_bn(name):
    """SparseBasicBlock's convs carry a bias and feed a train-mode BatchNorm (scn.py:60-66): the batch mean removes
    the bias, its true gradient is exactly zero and every implementation holds its own rounding noise there"""
    import re

    return re.search(r"middle_conv\.\d+\.conv[12]\.bias$", name) is not None


def _to_gpu(ex):
    out = {}
    for k, v in ex.items():
        if torch.is_tensor(v):
            out[k] = v.cuda()
        elif isinstance(v, list) and v and torch.is_tensor(v[0]):
            out[k] = [t.cuda() for t in v]
        else:
            out[k] = v
    return out


@pytest.mark.parametrize("name", ["kitti_second", "nusc_cbgs", "lyft_cbgs", "nusc_pp"])
def test_detector_matches_reference_detector_golden_fp32(name):
    """BASELINE.json configs[1..4] on cropped ranges: the HIP product (fp32 mode) against tests/golden/detectors.pt,
    i.e. the reference's OWN VoxelNet / PointPillars + scn.py (SpMiddleFHD, SpMiddleResNetFHD with its residual
    blocks and biased convs) + RPN + MultiGroupHead run in the build container.  north_star tolerance: raw box
    regressions within 1e-3; per-task losses 2e-4; gradient norms 1 % and whole small gradient tensors 2 %."""
    from det3d_amd import build_detector
    from oracle import model_ref as M

    S, g, cfg, ex = _golden_case(name)
    m = build_detector(cfg.model, cfg.train_cfg, cfg.test_cfg)
    assert S.param_fingerprint(m) == [(k, tuple(s)) for k, s in g["fingerprint"]]
    M.fill_deterministic(m)
    m = m.cuda().eval()
    exg = _to_gpu(ex)
    with torch.no_grad():
        data = dict(features=exg["voxels"], num_voxels=exg["num_points"], coors=exg["coordinates"],
                    batch_size=S.BATCH, input_shape=exg["shape"][0])
        preds = m.bbox_head(m.extract_feat(data))
        dets = m.bbox_head.predict(exg, preds, m.test_cfg)
    worst = 0.0
    for p, gp in zip(preds, g["preds"]):
        for k in gp:
            worst = max(worst, float((p[k][:, ::2, 1::2].float().cpu() - gp[k]).abs().max()))
    print("%s: max |pred - reference| = %.2e (max |ref box| %.2f)" % (
        name, worst, max(float(gp["box_preds"].abs().max()) for gp in g["preds"])))
    assert worst < 1e-3
    # detections.  fill_deterministic gives O(10) size regressions, i.e. exp()-decoded boxes of 1e4 m and more whose
    # rotated-NMS decisions are arbitrary on both sides; the strict detection comparison lives in
    # tests/test_nms_gpu.py::test_head_predict_matches_reference_golden (sane weights).  Here: the same number of
    # detections within a few, and 90 % of the reference's sane boxes (< 100 m everywhere) have a GPU twin within
    # 1e-3 (relative above 1) with the same score to 1e-4.
    for d, gd in zip(dets, g["dets"]):
        a, b = d["box3d_lidar"].cpu(), gd["box3d_lidar"]
        sa, sb = d["scores"].cpu(), gd["scores"]
        assert abs(a.shape[0] - b.shape[0]) <= max(3, b.shape[0] // 20), (a.shape, b.shape)
        keep_a, keep_b = (a.abs() < 100).all(1), (b.abs() < 100).all(1)
        a, b, sa, sb = a[keep_a], b[keep_b], sa[keep_a], sb[keep_b]
        if len(a) and len(b):
            nn = torch.cdist(b[:, :2], a[:, :2]).argmin(1)
            err = ((b - a[nn]).abs() / b.abs().clamp(min=1.0)).max(1)[0]
            ok = (err < 1e-3) & ((sb - sa[nn]).abs() < 1e-4)
            assert float(ok.float().mean()) >= 0.9, (int(ok.sum()), len(b))
    m.train()
    # the strided / transposed convolutions of the CBGS and PointPillars necks are MIOpen's: ask for its deterministic
    # solvers so that the gradient figures below do not move from box to box with the solver choice
    with torch.backends.cudnn.flags(enabled=True, benchmark=False, deterministic=True):
        losses = m(exg, return_loss=True)["loss"]
        for a, b in zip(losses, g["loss"]):
            assert float(a.detach()) == pytest.approx(float(b), rel=2e-4)
        params = dict(m.named_parameters())
        grads = dict(zip(params, torch.autograd.grad(sum(losses), list(params.values()), allow_unused=True)))
    # ---- the conditioning yardstick: float64 run of the CPU oracle on the same case (tests/golden/make_grad_f64.py).
    # dev(X) = deviation of X's gradients from the float64 ones (norms | elements of small tensors | direction);
    # the float32 ORACLE's own deviation is in the file.  Split arithmetic rounds products to 2^-17, float32 to 2^-24:
    # a path whose error sources are 2^7 coarser may sit up to 2^7 further from float64 than the float32 oracle does,
    # and no comparison against a float32 golden can be tighter than the golden's own distance from float64.
    from make_grad_f64 import deviation

    g64 = torch.load(os.path.join(ROOT, "tests", "golden", "detector_grads_f64.pt"), weights_only=False)[name]
    ref64 = {k: g64["grads"].get(k) for k in g64["grad_norms"]}
    have = {k: v for k, v in ref64.items() if v is not None}
    dev_hip = deviation({k: grads[k].detach().cpu() for k in have}, have, _bias_before_bn)
    dev_o32 = g64["oracle_f32_deviation"]
    norm_hip = max(abs(float(grads[k].float().norm()) - n) / n for k, n in g64["grad_norms"].items()
                   if n > 1e-4 * max(g64["grad_norms"].values()) and not _bias_before_bn(k) and grads[k] is not None)
    print("%s: vs float64 -- HIP norm %.2e (all tensors %.2e) element %.2e 1-cos %.2e | f32 oracle norm %.2e element %.2e "
          "1-cos %.2e | ratios %.1f %.1f %.1f" % (
              name, dev_hip["norm"], norm_hip, dev_hip["element"], dev_hip["one_minus_cos"], dev_o32["norm"],
              dev_o32["element"], dev_o32["one_minus_cos"], norm_hip / dev_o32["norm"],
              dev_hip["element"] / dev_o32["element"], dev_hip["one_minus_cos"] / dev_o32["one_minus_cos"]))
    # measured (profiles/round4_tests_figures.txt): HIP / f32-oracle ratios of the norm deviation 43 (KITTI SECOND: the
    # oracle's own 1.6e-4 is the smallest), 10, 4.6, 8.6; of the element deviation 26, 5.2, 4.7, 1.8; of 1 - cos 502,
    # 54, 56, 11 -- against the allowance of 2^7 (2^14 for 1 - cos)
    # Round 5: the allowance (2^7 / 2^14 x the oracle's own deviation) is replaced by what tests/test_precision_gpu.py
    # DEMONSTRATES on the same cases with three legs: the exact-f32 HIP path (same model, same host code, every conv
    # product one f32 fma) sits 0.8-2.6x the float32 oracle's distance from float64 -- so the distance of the split path
    # is its product format, not the kernels' summation order -- and the split path sits at the values below (identical
    # on every box and in rounds 4 and 5: every reduction is in a fixed order).  Bounds: 2x the measured values
    # (norm of the gradient tensors | elements of the small ones | 1 - cos).
    measured = {"kitti_second": (6.95e-3, 5.15e-2, 2.78e-4), "nusc_cbgs": (9.49e-3, 6.11e-2, 5.03e-4),
                "lyft_cbgs": (1.68e-2, 6.79e-2, 9.70e-4), "nusc_pp": (7.59e-3, 8.95e-2, 2.82e-4)}[name]
    assert norm_hip <= 2 * measured[0] and dev_hip["element"] <= 2 * measured[1]
    assert dev_hip["one_minus_cos"] <= 2 * measured[2]
    worst_t, worst_n, worst_c = 0.0, 0.0, 0.0
    gmax = max(g["grad_norms"].values())
    for k, gn in g["grad_norms"].items():
        got = grads[k]
        n = float(got.float().norm()) if got is not None else 0.0
        if gn < 1e-4 or _bias_before_bn(k):
            # conv bias in front of a train-mode BN: exactly-zero true gradient, both sides hold rounding noise
            assert n < 1e-2 * gmax, (k, n, gmax)
            continue
        worst_n = max(worst_n, abs(n - gn) / gn)
        if k in g["grads"]:
            ref = g["grads"][k]
            worst_t = max(worst_t, float((got.float().cpu() - ref).abs().max()) / float(ref.abs().max()))
            worst_c = max(worst_c, 1.0 - float(torch.nn.functional.cosine_similarity(
                got.float().cpu().flatten(), ref.flatten(), dim=0)))
    print("%s: worst gradient-norm rel err %.2e, worst small-tensor rel err %.2e, worst 1-cos %.2e" % (
        name, worst_n, worst_t, worst_c))
    # split arithmetic (2^-16 per product) against the exact-f32 oracle through up to 27 train-mode BN + ReLU layers
    # (see test_forward_loss_and_grads_fp32).  Measured: gradient norms within 0.5 % (KITTI) .. 2.0 % (lyft_cbgs:
    # 1.99e-2 / 2.00e-2 on different boxes -- that neck's strided and transposed convs are MIOpen's, whose solver
    # choice moves the last digits), elements within 10 % of the tensor's max, direction within 2e-3.  Bounds: 1.5x the
    # largest measured value.
    assert worst_n < 3e-2 and worst_t < 1e-1 and worst_c < 2e-3


@pytest.mark.parametrize("name", ["kitti_second", "nusc_cbgs", "lyft_cbgs", "nusc_pp"])
def test_every_detection_matches_the_reference_detector_with_a_sane_head(name):
    """tests/golden/detectors_sane.pt: the reference's own detectors with trained-like head magnitudes
    (oracle.model_ref.tame_head: boxes within metres of their anchors, scores on both sides of the threshold) and the
    reference's OWN predict().  Every reference detection must have its GPU twin: same count per frame, same label,
    box within 1e-3 (relative above 1 m / rad), score within 1e-4 -- no 90 % allowance."""
    from det3d_amd import build_detector
    from oracle import model_ref as M

    S, g, cfg, ex = _golden_case(name)
    gs = torch.load(os.path.join(ROOT, "tests", "golden", "detectors_sane.pt"), weights_only=False)[name]
    m = build_detector(cfg.model, cfg.train_cfg, cfg.test_cfg)
    M.fill_deterministic(m)
    M.tame_head(m, **S.SANE_HEAD[name])
    m = m.cuda().eval()
    exg = _to_gpu(ex)
    with torch.no_grad():
        data = dict(features=exg["voxels"], num_voxels=exg["num_points"], coors=exg["coordinates"],
                    batch_size=S.BATCH, input_shape=exg["shape"][0])
        preds = m.bbox_head(m.extract_feat(data))
        dets = m.bbox_head.predict(exg, preds, m.test_cfg)
    worst = max(float((p[k][:, ::2, 1::2].float().cpu() - gp[k]).abs().max()) for p, gp in zip(preds, gs["preds"]) for k in gp)
    assert worst < 1e-3, worst
    total = 0
    for f, (d, gd) in enumerate(zip(dets, gs["dets"])):
        a, b = d["box3d_lidar"].cpu(), gd["box3d_lidar"]
        sa, sb = d["scores"].cpu(), gd["scores"]
        la, lb = d["label_preds"].cpu(), gd["label_preds"]
        assert float(b.abs().max()) < 200  # the golden really is sane
        assert a.shape == b.shape, (name, f, a.shape, b.shape)
        # twin = the GPU detection of the same label closest in (x, y, z, w, l, h)
        dist = torch.cdist(b[:, :6], a[:, :6]) + 1e3 * (lb[:, None] != la[None, :]).float()
        nn = dist.argmin(1)
        assert len(set(nn.tolist())) == len(nn), (name, f, "two reference detections share one GPU twin")
        err = ((b - a[nn]).abs() / b.abs().clamp(min=1.0)).max(1)[0]
        assert bool((lb == la[nn]).all()), (name, f)
        assert float(err.max()) < 1e-3, (name, f, float(err.max()))
        assert float((sb - sa[nn]).abs().max()) < 1e-4, (name, f)
        total += len(b)
    print("%s: %d of %d reference detections matched (sane head), max |pred - reference| = %.2e" % (name, total, total, worst))


@pytest.mark.parametrize("name", ["nusc_cbgs", "lyft_cbgs"])
def test_resnet_fhd_backbone_vs_oracle_on_a_larger_cloud(name):
    """SpMiddleResNetFHD alone, 3x the golden's voxel count: HIP fp32 vs oracle/model_ref.SpMiddleResNetFHDRef
    (pinned to the reference's scn.py by the detector golden) -- dense output, input gradient, all parameter
    gradients."""
    from det3d_amd.backbones import SpMiddleResNetFHD
    from oracle import clib
    from oracle import model_ref as M

    S, g, cfg, _ = _golden_case(name)
    case = dict(S.CASES[name], points=18000)
    cin = cfg.model.backbone.num_input_features
    vox, coors, nums = [], [], []
    for b in range(2):
        v, cc, n = clib.points_to_voxel(S.synth_points(case, seed=7 + b), case["voxel"], case["range"], case["T"], True, 60000)
        vox.append(v); nums.append(n)
        coors.append(np.pad(cc, ((0, 0), (1, 0)), mode="constant", constant_values=b))
    v, n = torch.from_numpy(np.concatenate(vox)), torch.from_numpy(np.concatenate(nums))
    coors = torch.from_numpy(np.concatenate(coors))
    feats = v[:, :, :cin].sum(1) / n.float().view(-1, 1)
    grid = [int(s) for s in S.grid_of(case)]
    torch.manual_seed(0)
    ref = M.SpMiddleResNetFHDRef(cin).train()
    M.fill_deterministic(ref)
    prod = SpMiddleResNetFHD(num_input_features=cin)
    prod.load_state_dict(ref.state_dict())
    prod = prod.cuda().train()
    f_ref = feats.clone().requires_grad_(True)
    out_ref = ref(f_ref, coors.numpy(), 2, grid)
    gout = torch.randn(out_ref.shape, generator=torch.Generator().manual_seed(1))
    out_ref.backward(gout)
    f_gpu = feats.cuda().requires_grad_(True)
    out = prod(f_gpu, coors.cuda(), 2, grid)
    out.backward(gout.cuda())
    scale = float(out_ref.abs().max())
    assert float((out.cpu() - out_ref).abs().max()) < 1e-3 * max(scale, 1.0)
    # Gradients are compared in relative L2 and direction, not element-wise: the f32 path multiplies in split
    # arithmetic (2^-16 relative per product, not 2^-24), so a pre-activation within ~1e-5 of zero can land on the
    # other side of a ReLU than in the oracle.  The forward effect is nil (checked to 1e-3 above); the backward mask is
    # discontinuous there, so single sites differ by O(1) while the tensors agree to a fraction of a percent.
    l2 = lambda a, b: float((a - b).norm()) / (float(b.norm()) + 1e-30)
    gi = l2(f_gpu.grad.cpu(), f_ref.grad)
    assert gi < 3e-2, gi
    worst, worst_cos = 0.0, 0.0
    for (k, p), (_, q) in zip(prod.named_parameters(), ref.named_parameters()):
        if float(q.grad.abs().max()) < 1e-5 or _bias_before_bn(k):
            continue
        worst = max(worst, l2(p.grad.cpu(), q.grad))
        worst_cos = max(worst_cos, 1.0 - float(torch.nn.functional.cosine_similarity(p.grad.cpu().flatten(), q.grad.flatten(), dim=0)))
    print("%s ResNetFHD %d voxels: input-grad rel L2 %.2e, worst param-grad rel L2 %.2e, worst 1-cos %.2e" % (
        name, len(feats), gi, worst, worst_cos))
    # (measured: 9e-3 / 2e-2 / 2e-4 -- 27 train-mode BN + ReLU layers with random O(1) weights amplify the last bits)
    assert worst < 5e-2 and worst_cos < 1e-3


def test_nodes_that_hold_packed_weights_refuse_a_backward_behind_the_in_place_repack():
    """ADVICE round 4 (sparse_block.py:118, repack.py:78): Repacker.run() rewrites the cached packed operands IN PLACE.
    A graph built before it holds those very buffers, so its backward would differentiate with the NEW weights -- the
    nodes take repack.stamp() in their forward and raise in their backward instead.  Also: the fused block keeps its
    activations in save_for_backward (an in-place change of the input between forward and backward is an autograd error,
    not a silent wrong gradient), and a second backward over a retained graph equals the first to rounding."""
    from det3d_amd import repack, spconv
    from det3d_amd.backbones import SparseBasicBlock

    torch.manual_seed(7)
    dev = torch.device("cuda", 0)
    rng = np.random.default_rng(2)
    cells = np.unique(rng.integers(0, [2, 12, 40, 40], size=(6000, 4)), axis=0).astype(np.int32)
    idx = torch.from_numpy(cells).to(dev)
    blk = SparseBasicBlock(64, 64, indice_key="res").to(dev).train()
    conv = spconv.SubMConv3d(64, 64, 3, bias=False, indice_key="res").to(dev)
    w = torch.randn((len(cells), 64), device=dev)

    def forward(m, leaf=True):
        f = torch.randn((len(cells), 64), device=dev, requires_grad=True)
        src = f if leaf else f * 1.0
        x = spconv.SparseConvTensor(src, idx, [12, 40, 40], 2)
        spconv.SparseSequential(m).plan(x)
        spconv.ARENA.reset(dev)
        return f, src, (m(x).features * w).sum()

    # retained graph: two backward runs, same gradients
    f, _, loss = forward(blk)
    loss.backward(retain_graph=True)
    g1, p1 = f.grad.clone(), [p.grad.clone() for p in blk.parameters()]
    f.grad = None
    blk.zero_grad()
    loss.backward()
    # (the second run takes bn1's backward sums from the stand-alone sweep instead of conv2's epilogue: equal to rounding)
    rel = lambda a, b: float((a - b).norm()) / (float(b.norm()) + 1e-30)
    assert rel(f.grad, g1) < 1e-5 and all(rel(p.grad, q) < 1e-5 for p, q in zip(blk.parameters(), p1))

    # in-place change of the (non-leaf) input between forward and backward: autograd notices
    f, src, loss = forward(blk, leaf=False)
    with torch.no_grad():
        src.mul_(2.0)
    with pytest.raises(RuntimeError, match="modified by an inplace operation"):
        loss.backward()

    # a repack between forward and backward
    for m in (blk, conv):
        f, _, loss = forward(m)
        rp = repack.Repacker(list(m.parameters()))
        for p in m.parameters():
            p.data.mul_(1.01)  # as the fused Adam kernel does: through the storage, no autograd version bump
        assert rp.run() >= 1
        with pytest.raises(RuntimeError, match="Repacker.run"):
            loss.backward()
        f, _, loss = forward(m)  # a fresh graph is fine
        loss.backward()
        assert torch.isfinite(f.grad).all()
