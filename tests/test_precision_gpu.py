"""What the split arithmetic of the f32 mode costs, DEMONSTRATED instead of allowed for.

The product multiplies f32 operands as three bf16 MFMAs on hi / lo pairs (~2^-16 relative per product).
`det3d_amd.precision.exact_f32()` runs the same model and host code with every convolution product as one exact f32
multiply-add (sparse: libd3d_hip's VALU kernels behind d3d_set_tuning knob 10; dense: MIOpen f32).  Three legs --
HIP split, HIP exact f32, CPU oracle float32 (+ float64 for gradients) -- on the same inputs:

  * forward: split vs exact raw predictions within 1e-4 of the prediction scale on all four detector goldens;
  * gradients: each leg's distance from the float64 oracle (tests/golden/detector_grads_f64.pt).  The exact-f32 HIP leg
    must sit where the float32 ORACLE sits (same product format, another summation order), the split leg further out by
    the factor its coarser products explain -- the measured factors are the bounds of tests/test_e2e_gpu.py now;
  * training: 300 optimizer steps of KITTI SECOND from the same seed in all three legs (the CPU leg is the committed
    tests/golden/train_curve_f32.pt): loss curves and held-out predictions of split vs exact no further apart than the two
    exact-f32 legs (HIP vs CPU) are from each other, times a stated factor.
"""
import contextlib
import os
import sys

import numpy as np
import pytest
import torch

from conftest import GOLDEN, ROOT

pytestmark = pytest.mark.gpu

sys.path.insert(0, GOLDEN)


def _mode(exact):
    from det3d_amd import precision

    return precision.exact_f32() if exact else contextlib.nullcontext()


def _detector_legs(name):
    """predictions (eval), losses and gradients (train) of one detector golden in split and in exact-f32 arithmetic"""
    from det3d_amd import build_detector
    from oracle import model_ref as M
    from test_e2e_gpu import _golden_case, _to_gpu

    S, g, cfg, ex = _golden_case(name)
    exg = _to_gpu(ex)
    legs = {}
    for exact in (False, True):
        with _mode(exact):
            m = build_detector(cfg.model, cfg.train_cfg, cfg.test_cfg)
            M.fill_deterministic(m)
            m = m.cuda().eval()
            with torch.no_grad():
                data = dict(features=exg["voxels"], num_voxels=exg["num_points"], coors=exg["coordinates"],
                            batch_size=S.BATCH, input_shape=exg["shape"][0])
                preds = [{k: v.float().cpu() for k, v in p.items()} for p in m.bbox_head(m.extract_feat(data))]
            m.train()
            with torch.backends.cudnn.flags(enabled=True, benchmark=False, deterministic=True):
                losses = m(exg, return_loss=True)["loss"]
                params = dict(m.named_parameters())
                grads = dict(zip(params, torch.autograd.grad(sum(losses), list(params.values()), allow_unused=True)))
            legs[exact] = (preds, [float(l) for l in losses], {k: v.detach().cpu() for k, v in grads.items() if v is not None})
    return g, legs


# measured on MI355X (profiles/round5_precision_figures.txt), per case: the bounds below are <= 2x these
#   exact-f32 HIP / f32 oracle, distance from float64:  norm, element, 1 - cos
#   split HIP / exact-f32 HIP, distance from float64:   norm, element, 1 - cos
#   kitti_second   f32 oracle 1.6e-4 2.0e-3 5.6e-7 | HIP exact-f32 2.5e-4 2.0e-3 6.9e-7 | HIP split 7.0e-3 5.2e-2 2.8e-4
#   nusc_cbgs                 1.1e-3 1.1e-2 8.9e-6 |               1.7e-3 1.0e-2 1.0e-5 |           9.5e-3 6.1e-2 5.0e-4
#   lyft_cbgs                 3.5e-3 1.4e-2 1.7e-5 |      4.2..8.8e-3 3.1e-2 0.8..1.7e-4 |           1.7e-2 6.8e-2 9.7e-4
#   nusc_pp                   4.8e-4 2.1e-2 1.4e-5 |               3.9e-4 2.1e-2 1.2e-5 |           7.6e-3 9.0e-2 2.8e-4
# (lyft's exact leg moves from run to run: its strided / transposed neck convolutions are MIOpen's f32 kernels)
# => the exact-f32 HIP leg is in the float32 oracle's class: norms and elements within 0.8-2.6x of the oracle's own distance
#    from float64, 1 - cos within 0.9-9.7x; the split leg sits 4-28x (norms), 2-26x (elements), 6-400x (1 - cos) beyond it.
EXACT_VS_ORACLE_BOUND = 5.0       # norms, elements (2x the largest measured ratio)
EXACT_VS_ORACLE_COS_BOUND = 20.0  # 1 - cos
# worst per-tensor relative L2 and 1 - cos between the split and the exact-f32 gradients: 2x the measured values
# (2.4e-2 / 2.8e-4, 3.2e-2 / 4.9e-4, 5.2e-2 / 1.3e-3, 2.4e-2 / 2.8e-4)
SPLIT_VS_EXACT_BOUND = {"kitti_second": (4.8e-2, 5.6e-4), "nusc_cbgs": (6.4e-2, 1.0e-3), "lyft_cbgs": (1.05e-1, 2.7e-3),
                        "nusc_pp": (4.8e-2, 5.6e-4)}


@pytest.mark.parametrize("name", ["kitti_second", "nusc_cbgs", "lyft_cbgs", "nusc_pp"])
def test_split_against_exact_f32_on_the_detector_goldens(name):
    from make_grad_f64 import deviation
    from test_e2e_gpu import _bias_before_bn

    g, legs = _detector_legs(name)
    (p_s, l_s, g_s), (p_e, l_e, g_e) = legs[False], legs[True]
    # ---- forward: same kernels' worth of work, only the product format differs
    scale = max(float(gp["box_preds"].abs().max()) for gp in g["preds"])
    fwd = max(float((a[k] - b[k]).abs().max()) for a, b in zip(p_s, p_e) for k in a)
    ref = max(float((a[k][:, ::2, 1::2] - gp[k]).abs().max()) for a, gp in zip(p_e, g["preds"]) for k in gp)
    print("%s: forward split vs exact-f32 %.2e, exact-f32 vs reference golden %.2e (prediction scale %.2f)" % (name, fwd, ref, scale))
    # measured: split vs exact 1.6e-5 .. 2.9e-5 of the prediction scale, exact vs the reference golden 2e-6
    assert fwd <= 6e-5 * max(scale, 1.0)
    assert ref <= 1e-5 * max(scale, 1.0)  # exact f32 on the GPU == the reference's f32 to summation order
    for a, b, c in zip(l_s, l_e, g["loss"]):
        assert a == pytest.approx(b, rel=2e-4) and b == pytest.approx(float(c), rel=5e-5)
    # ---- gradients: every leg against the float64 oracle
    g64 = torch.load(os.path.join(GOLDEN, "detector_grads_f64.pt"), weights_only=False)[name]
    have = {k: v for k, v in ((k, g64["grads"].get(k)) for k in g64["grad_norms"]) if v is not None}
    dev_s = deviation({k: g_s[k] for k in have}, have, _bias_before_bn)
    dev_e = deviation({k: g_e[k] for k in have}, have, _bias_before_bn)
    dev_o = g64["oracle_f32_deviation"]
    l2 = cos = 0.0
    gmax = max(float(v.norm()) for v in g_e.values())
    for k, e in g_e.items():
        if float(e.norm()) < 1e-4 * gmax or _bias_before_bn(k):
            continue
        l2 = max(l2, float((g_s[k] - e).norm()) / float(e.norm()))
        cos = max(cos, 1.0 - float(torch.nn.functional.cosine_similarity(g_s[k].flatten(), e.flatten(), dim=0)))
    print("%s: distance from float64 (norm | element | 1-cos): f32 oracle %.2e %.2e %.2e | HIP exact-f32 %.2e %.2e %.2e | "
          "HIP split %.2e %.2e %.2e | split vs exact: worst tensor rel L2 %.2e, worst 1-cos %.2e" % (
              name, dev_o["norm"], dev_o["element"], dev_o["one_minus_cos"], dev_e["norm"], dev_e["element"],
              dev_e["one_minus_cos"], dev_s["norm"], dev_s["element"], dev_s["one_minus_cos"], l2, cos))
    # the exact-f32 HIP leg is a float32 evaluation like the oracle: same class of distance from float64
    assert dev_e["norm"] <= EXACT_VS_ORACLE_BOUND * dev_o["norm"] + 1e-6
    assert dev_e["element"] <= EXACT_VS_ORACLE_BOUND * dev_o["element"] + 1e-6
    assert dev_e["one_minus_cos"] <= EXACT_VS_ORACLE_COS_BOUND * dev_o["one_minus_cos"] + 1e-9
    # split vs exact on the same GPU, tensor by tensor (the ReLU gates that flip at 2^-16 move single elements by O(1);
    # whole tensors stay put): bounds = 2x the measured values per case
    bl2, bcos = SPLIT_VS_EXACT_BOUND[name]
    assert l2 <= bl2 and cos <= bcos


_LEGS = {}  # (exact, steps, eps, seed) -> (losses, held-out predictions): the two 300-step tests share their common legs


def _train_leg(exact, steps, eps=0.0, unfused=False, seed=None):
    key = (bool(exact), int(steps), float(eps), seed)
    if not unfused and key in _LEGS:
        return _LEGS[key]
    out = _train_leg_run(exact, steps, eps, unfused, seed)
    if not unfused:
        _LEGS[key] = out
    return out


def _train_leg_run(exact, steps, eps=0.0, unfused=False, seed=None):
    """one training run of tests/golden/train_run_spec.py on the GPU.  exact: precision.exact_f32(); eps: the initial
    weights multiplied by (1 + eps) -- the sensitivity probe; unfused: the per-layer sparse path instead of the fused stack;
    seed: another default initialisation (torch.manual_seed(seed) on the CPU generator) instead of the spec's seed 0"""
    import train_run_spec as T
    from det3d_amd import build_detector, sparse_block, sparse_stack, synthetic

    cfg = T.config()
    ds = T.dataset()
    dev = torch.device("cuda")
    saved = (sparse_stack.FUSE_STACK, sparse_block.FUSE_BLOCK)
    try:
        if unfused:
            sparse_stack.FUSE_STACK = sparse_block.FUSE_BLOCK = False
        with _mode(exact):
            model = build_detector(cfg.model, cfg.train_cfg, cfg.test_cfg)
            if seed is None:
                model.load_state_dict(T.initial_state(cfg))
            else:
                torch.manual_seed(seed)
                model.load_state_dict(build_detector(cfg.model, cfg.train_cfg, cfg.test_cfg).state_dict())
            if eps:
                with torch.no_grad():
                    for p in model.parameters():
                        p.mul_(1.0 + eps)
            model = model.to(dev).train()
            step = T.train_step_for(model, cfg)
            batches = [synthetic.make_example(ds.device_batch(i, T.BATCH, dev), synthetic.KITTI_CAR) for i in range(T.N_BATCHES)]
            losses = []
            for i in range(steps):
                losses.append(step(batches[i % T.N_BATCHES]))
            losses = [float(l) for l in torch.stack(losses).cpu()]
            model.eval()
            with torch.no_grad():
                ex = synthetic.make_example(ds.device_batch(T.HELD_OUT, T.BATCH, dev), synthetic.KITTI_CAR)
                data = dict(features=ex["voxel_mean"], num_voxels=ex["num_points"], coors=ex["coordinates"],
                            batch_size=T.BATCH, input_shape=ex["shape"][0])
                preds = T.subsample(model.bbox_head(model.extract_feat(data))[0])
    finally:
        sparse_stack.FUSE_STACK, sparse_block.FUSE_BLOCK = saved
    return np.array(losses), preds


def test_300_training_steps_split_vs_exact_f32_vs_cpu_oracle():
    """Same seed, same batches, same schedule (tests/golden/train_run_spec.py: KITTI SECOND, 2 x 6000 points, the
    config's one-cycle Adam over 300 steps, loss 244 -> 0.2-0.6) in five legs: HIP split, HIP exact f32, each of them
    again from initial weights multiplied by (1 + 1e-6), and the committed CPU-oracle curve.

    What the legs show (profiles/round5_train_legs.json, tools/train_legs.py): the run is a CHAOTIC map.  A 1e-6
    perturbation of the initial weights moves the loss of step 10 by 3 % and the mean loss of the last 20 steps by a factor of
    two (split: 0.56 vs 0.24; exact f32: 0.30 vs 0.30 vs the CPU oracle's 0.35), and the raw predictions on a held-out
    frame differ by half their scale between ANY two legs, the two exact-f32 ones (HIP vs CPU) included -- so "final
    predictions within 1e-3" is not a property any pair of float32 implementations has here, and the comparison has to be
    statistical: the split legs must (a) follow the oracle to the precision of the product format while the trajectories
    are still together (the first steps), (b) train as far as the exact ones (final loss inside the band the exact legs
    span, widened by 2), (c) differ from the exact legs no more than the exact legs differ among themselves or a split
    leg from its own perturbed twin.  Independently of chaos, (d) the fused sparse stack is the per-layer path BIT for
    BIT over 60 optimizer steps."""
    import train_run_spec as T

    path = os.path.join(GOLDEN, "train_curve_f32.pt")
    if not os.path.exists(path):
        pytest.skip("tests/golden/train_curve_f32.pt not generated (python tests/golden/make_train_curve.py)")
    cpu = torch.load(path, weights_only=False)
    steps = cpu["steps"]
    assert steps == T.STEPS
    legs = {"cpu_oracle": (np.array(cpu["losses"]), cpu["held_out_preds"]),
            "split": _train_leg(False, steps), "split_eps": _train_leg(False, steps, eps=1e-6),
            "exact": _train_leg(True, steps), "exact_eps": _train_leg(True, steps, eps=1e-6)}
    last = {k: float(v[0][-20:].mean()) for k, v in legs.items()}
    for k, (l, _) in legs.items():
        assert np.isfinite(l).all() and last[k] < 0.01 * l[0], (k, last[k])  # every leg trains: 244 -> below 2.4
    rel = lambda a, b: np.abs(a - b) / np.maximum(np.abs(b), 1e-3)
    dp = lambda a, b: max(float((a[k] - b[k]).abs().max()) for k in a)
    scale = max(float(v.abs().max()) for v in legs["exact"][1].values())
    pairs = [("split", "exact"), ("split", "cpu_oracle"), ("exact", "cpu_oracle"), ("split", "split_eps"), ("exact", "exact_eps")]
    early = {p: float(rel(legs[p[0]][0][:3], legs[p[1]][0][:3]).max()) for p in pairs}
    med = {p: float(np.median(rel(legs[p[0]][0], legs[p[1]][0]))) for p in pairs}
    pred = {p: dp(legs[p[0]][1], legs[p[1]][1]) for p in pairs}
    print("300 steps, loss %.2f -> mean of the last 20: %s" % (legs["split"][0][0], {k: round(v, 4) for k, v in last.items()}))
    print("relative loss difference, first 3 steps max: %s" % {"%s vs %s" % p: "%.2e" % v for p, v in early.items()})
    print("relative loss difference, median over the run: %s" % {"%s vs %s" % p: "%.3f" % v for p, v in med.items()})
    print("held-out raw predictions after 300 steps, max abs difference (scale %.2f): %s" % (
        scale, {"%s vs %s" % p: "%.2f" % v for p, v in pred.items()}))
    # (a) while the trajectories are together.  Step 0 sees identical weights (loss equal to 1e-6), step 1 the weights after
    #     ONE Adam update from each leg's gradients (measured: split 3e-4, exact 6e-5 from the oracle), step 2 after two
    #     (split 5e-3, exact 3e-4: Adam's first updates are sign-like, a gradient element that is 1 % off near zero moves its
    #     weight by the full learning rate) -- bounds 2-3x the measured values
    first2 = {p: float(rel(legs[p[0]][0][:2], legs[p[1]][0][:2]).max()) for p in pairs}
    assert first2[("split", "cpu_oracle")] <= 1e-3 and first2[("exact", "cpu_oracle")] <= 2e-4, first2
    assert early[("split", "exact")] <= 1e-2 and early[("split", "cpu_oracle")] <= 1e-2 and early[("exact", "cpu_oracle")] <= 1e-3
    # (b) where the runs end: the exact-f32 legs span [lo, hi]; the split legs end inside [lo / 2, 2 hi]
    lo = min(last["exact"], last["exact_eps"], last["cpu_oracle"])
    hi = max(last["exact"], last["exact_eps"], last["cpu_oracle"])
    assert lo / 2 <= last["split"] <= 2 * hi and lo / 2 <= last["split_eps"] <= 2 * hi, (last, lo, hi)
    # (c) split differs from exact no more than twice what a leg differs from its own 1e-6-perturbed twin / the two exact
    #     implementations differ from each other
    yard = max(med[("split", "split_eps")], med[("exact", "exact_eps")], med[("exact", "cpu_oracle")])
    assert med[("split", "exact")] <= 2.0 * yard + 1e-2, (med, yard)
    yard_p = max(pred[("split", "split_eps")], pred[("exact", "exact_eps")], pred[("exact", "cpu_oracle")])
    assert pred[("split", "exact")] <= 2.0 * yard_p + 1e-3 * scale, (pred, yard_p)
    # (d) fused sparse stack == per-layer path, bit for bit, over 60 optimizer steps
    l_unfused, _ = _train_leg(False, 60, unfused=True)
    assert np.array_equal(l_unfused, legs["split"][0][:60])


def test_300_training_steps_five_initialisations_per_mode():
    """VERDICT round 5 (weak 1): the comparison above rests on one seed pair per mode, and the run is chaotic -- one pair
    says little.  Here: FIVE initialisations per mode -- the spec's seed-0 weights, their (1 + 1e-6) twin (both shared
    with the test above: the legs are cached) and torch.manual_seed(1..3) -- each trained for 300 steps in split
    arithmetic and in exact f32 from the same weights, batches and schedule.  Asserted on the two DISTRIBUTIONS of the
    final loss (mean of the last 20 steps): every leg trains (loss falls by > 100 x), the medians of the two modes agree
    within the exact-f32 legs' own max / min spread (at least a factor 1.5), and no split leg ends outside [min / 2,
    2 max] of the exact legs.  Which mode ends lower flips from initialisation to initialisation (printed)."""
    import train_run_spec as T

    inits = [dict(), dict(eps=1e-6), dict(seed=1), dict(seed=2), dict(seed=3)]
    split = [_train_leg(False, T.STEPS, **kw)[0] for kw in inits]
    exact = [_train_leg(True, T.STEPS, **kw)[0] for kw in inits]
    fs = np.array([l[-20:].mean() for l in split])
    fe = np.array([l[-20:].mean() for l in exact])
    print("300 steps x 5 initialisations, final loss (mean of the last 20): split %s | exact f32 %s" % (np.round(fs, 4).tolist(), np.round(fe, 4).tolist()))
    print("medians: split %.4f exact %.4f; exact spread max / min %.2f; split lower than exact on %d of %d" % (
        np.median(fs), np.median(fe), fe.max() / fe.min(), int((fs < fe).sum()), len(inits)))
    for l in split + exact:
        assert np.isfinite(l).all() and l[-20:].mean() < 0.01 * l[0]
    spread = max(fe.max() / fe.min(), 1.5)
    assert 1.0 / spread <= np.median(fs) / np.median(fe) <= spread, (fs, fe)
    assert fe.min() / 2 <= fs.min() and fs.max() <= 2 * fe.max(), (fs, fe)
    first = max(abs(a[0] - b[0]) / abs(b[0]) for a, b in zip(split, exact))
    assert first <= 1e-4, first  # step 0 sees identical weights: the forward agrees to the product format
