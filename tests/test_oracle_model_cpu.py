"""oracle/model_ref.py (the CPU SECOND restatement used as end-to-end checker and CPU
baseline) pinned against goldens from the reference's own RPN / MultiGroupHead, and
state-dict compatible with the product model."""
import os

import pytest
import torch

from conftest import GOLDEN, ROOT

from oracle import model_ref as M


def _load(name):
    return torch.load(os.path.join(GOLDEN, name), weights_only=False)


@pytest.mark.parametrize("name", ["second", "pillars", "nusc_pp", "cbgs"])
def test_rpn_ref_vs_reference_golden(name):
    g = _load("rpn.pt")[name]
    m = M.RPNRef(**g["cfg"])
    m.load_state_dict(g["state"])
    m.eval()
    assert torch.allclose(m(g["x"]), g["out_eval"], atol=1e-5, rtol=1e-5)
    m.train()
    assert torch.allclose(m(g["x"]), g["out_train"], atol=1e-4, rtol=1e-4)


def _second_from_head_golden(g):
    from det3d_amd.builder import GroundBox3dCoderTorch

    c = g["coder"]
    coder = GroundBox3dCoderTorch(c["linear_dim"], c["vec"], n_dim=c["n_dim"])
    cfg = dict(model=dict(
        reader=dict(type="VoxelFeatureExtractorV3", num_input_features=4),
        backbone=dict(type="SpMiddleFHD", num_input_features=4),
        neck=dict(layer_nums=[1], ds_layer_strides=[1], ds_num_filters=[g["in_ch"]], us_layer_strides=[1],
                  us_num_filters=[g["in_ch"]], num_input_features=128),
        bbox_head=dict(tasks=g["tasks"], in_channels=g["in_ch"], loss_aux=g["loss_aux"], box_coder=coder,
                       encode_rad_error_by_sin=not c["vec"], direction_offset=g["direction_offset"],
                       loss_bbox=dict(loss_weight=2.0), loss_cls=dict(loss_weight=1.0),
                       loss_norm=dict(pos_cls_weight=1.0, neg_cls_weight=2.0))), test_cfg=g["test_cfg"])
    m = M.SecondRef(cfg)
    m.bbox_head.load_state_dict(g["state"])
    return m


@pytest.mark.parametrize("name", ["kitti", "nusc", "lyft"])
def test_head_loss_and_predict_vs_reference_golden(name):
    g = _load("head.pt")[name]
    m = _second_from_head_golden(g)
    preds = [t(g["x"]) for t in m.bbox_head.tasks]
    for p, gp in zip(preds, g["preds"]):
        for k in gp:
            assert torch.allclose(p[k], gp[k], atol=1e-5)
    ex = dict(anchors=g["anchors"], labels=g["labels"], reg_targets=g["reg_targets"])
    loss = m.loss(ex, preds)["loss"]
    for a, b in zip(loss, g["loss"]):
        assert torch.allclose(a, b, atol=1e-4, rtol=1e-5)
    dets = m.predict(ex, preds)
    for d, gd in zip(dets, g["dets"]):
        assert d["box3d_lidar"].shape == gd["box3d_lidar"].shape
        assert torch.allclose(d["box3d_lidar"], gd["box3d_lidar"], atol=1e-5)
        assert torch.allclose(d["scores"], gd["scores"], atol=1e-6)
        assert torch.equal(d["label_preds"], gd["label_preds"])


def test_state_dict_compatible_with_product_model():
    from det3d_amd import Config, build_detector

    cfg = Config.fromfile(os.path.join(ROOT, "det3d_amd", "configs", "kitti_car_second.py"))
    prod = build_detector(cfg.model, cfg.train_cfg, cfg.test_cfg)
    ref = M.SecondRef(cfg)
    missing, unexpected = ref.load_state_dict(prod.state_dict(), strict=True)
    assert not missing and not unexpected


# ---------------------------------------------------------------- whole detectors vs the reference's classes
def _detector_case(name):
    sys_path = os.path.join(ROOT, "tests", "golden")
    import sys
    if sys_path not in sys.path:
        sys.path.insert(0, sys_path)
    import detector_specs as S

    g = _load("detectors.pt")[name]
    cfg = S.load_cfg(name)
    coder = cfg.model.bbox_head.box_coder
    ex, fmap, nv = S.make_example(name, cfg.model.bbox_head.tasks, coder.n_dim, coder.code_size)
    S.check_input_sums(ex, g["input_sums"])
    return S, g, cfg, ex


def _compare_with_detector_golden(g, preds, dets, losses, named_grads, tol_pred=1e-4, tol_loss=1e-4, tol_grad=2e-3):
    for p, gp in zip(preds, g["preds"]):
        for k in gp:
            assert float((p[k][:, ::2, 1::2].float().cpu() - gp[k]).abs().max()) < tol_pred, k
    for d, gd in zip(dets, g["dets"]):
        assert d["box3d_lidar"].shape == gd["box3d_lidar"].shape
        assert torch.equal(d["label_preds"].cpu(), gd["label_preds"])
        assert float((d["box3d_lidar"].cpu() - gd["box3d_lidar"]).abs().max()) < 10 * tol_pred
        assert float((d["scores"].cpu() - gd["scores"]).abs().max()) < tol_pred
    for a, b in zip(losses, g["loss"]):
        assert float(a.detach()) == pytest.approx(float(b), rel=tol_loss)
    worst = 0.0
    for k, gn in g["grad_norms"].items():
        got = named_grads[k]
        n = float(got.float().norm()) if got is not None else 0.0
        # a conv bias in front of a train-mode BatchNorm has an exactly-zero true gradient: rounding noise
        if gn < 1e-4:
            assert n < 1e-3, k
            continue
        assert n == pytest.approx(gn, rel=tol_grad), k
        if k in g["grads"]:
            ref = g["grads"][k]
            worst = max(worst, float((got.float().cpu() - ref).abs().max()) / float(ref.abs().max()))
    assert worst < 5 * tol_grad, worst


@pytest.mark.parametrize("name", ["kitti_second", "nusc_cbgs", "lyft_cbgs", "nusc_pp"])
def test_oracle_detector_vs_reference_detector_golden(name):
    """oracle/model_ref.DetectorRef (SpMiddleFHDRef / SpMiddleResNetFHDRef / PillarFeatureNetRef ...) against the
    REFERENCE's own VoxelNet / PointPillars (scn.py executed over oracle/spconv_shim.py), same weights by
    fill_deterministic: state-dict layout, raw predictions, detections, per-task losses, every gradient."""
    S, g, cfg, ex = _detector_case(name)
    m = M.DetectorRef(cfg)
    assert S.param_fingerprint(m) == [(k, tuple(s)) for k, s in g["fingerprint"]]
    M.fill_deterministic(m)
    m.eval()
    with torch.no_grad():
        x = m.features(ex)
        preds = [t(x) for t in m.bbox_head.tasks]
        dets = m.predict(ex, preds)
    m.train()
    losses = m(ex)["loss"]
    params = dict(m.named_parameters())
    grads = torch.autograd.grad(sum(losses), list(params.values()), allow_unused=True)
    _compare_with_detector_golden(g, preds, dets, losses, dict(zip(params, grads)))


@pytest.mark.parametrize("name", ["kitti_second", "nusc_cbgs", "lyft_cbgs", "nusc_pp"])
def test_oracle_detector_vs_reference_detector_golden_with_a_sane_head(name):
    """detectors_sane.pt (the reference's detectors after model_ref.tame_head: trained-like head magnitudes, boxes within
    metres of their anchors): the oracle reproduces the reference's raw predictions and EVERY detection of its predict()."""
    S, g, cfg, ex = _detector_case(name)
    gs = _load("detectors_sane.pt")[name]
    m = M.DetectorRef(cfg)
    M.fill_deterministic(m)
    M.tame_head(m, **S.SANE_HEAD[name])
    m.eval()
    with torch.no_grad():
        x = m.features(ex)
        preds = [t(x) for t in m.bbox_head.tasks]
        dets = m.predict(ex, preds)
    for p, gp in zip(preds, gs["preds"]):
        for k in gp:
            assert float((p[k][:, ::2, 1::2] - gp[k]).abs().max()) < 1e-4, k
    n = 0
    for d, gd in zip(dets, gs["dets"]):
        assert float(gd["box3d_lidar"].abs().max()) < 200 and gd["scores"].shape[0] > 0
        assert d["box3d_lidar"].shape == gd["box3d_lidar"].shape
        assert torch.equal(d["label_preds"], gd["label_preds"])
        assert float((d["box3d_lidar"] - gd["box3d_lidar"]).abs().max()) < 1e-3
        assert float((d["scores"] - gd["scores"]).abs().max()) < 1e-4
        n += gd["scores"].shape[0]
    assert n >= 40


def test_product_point_pillars_cpu_path_vs_reference_detector_golden():
    """BASELINE configs[0]/[2] plumbing: the product's PyTorch-op PointPillars path on the CPU against the
    reference's PointPillars detector (nuScenes 6-group head)."""
    from det3d_amd import build_detector

    S, g, cfg, ex = _detector_case("nusc_pp")
    m = build_detector(cfg.model, cfg.train_cfg, cfg.test_cfg)
    assert S.param_fingerprint(m) == [(k, tuple(s)) for k, s in g["fingerprint"]]
    M.fill_deterministic(m)
    m.eval()
    with torch.no_grad():
        data = dict(features=ex["voxels"], num_voxels=ex["num_points"], coors=ex["coordinates"],
                    batch_size=S.BATCH, input_shape=ex["shape"][0])
        preds = m.bbox_head(m.extract_feat(data))
    for p, gp in zip(preds, g["preds"]):
        for k in gp:
            assert float((p[k][:, ::2, 1::2] - gp[k]).abs().max()) < 1e-4, k
    m.train()
    losses = m(ex, return_loss=True)["loss"]
    for a, b in zip(losses, g["loss"]):
        assert float(a.detach()) == pytest.approx(float(b), rel=1e-4)
