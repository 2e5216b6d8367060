"""Conditioning of the detector gradients: the CPU oracle (oracle/model_ref.DetectorRef, pinned to the reference's own
VoxelNet / PointPillars by tests/test_oracle_model_cpu.py) run in float64 and in float32 on the four golden cases.

    python tests/golden/make_grad_f64.py        (CPU, ~2 minutes; needs neither /root/reference nor a GPU)

Why: tests/test_e2e_gpu.py compares the HIP path's gradients with float32 goldens through 14-27 train-mode
BatchNorm + ReLU layers.  Round 3 chose those bounds as "1.5x the largest value seen".  Here the float64 run is the
yardstick: the file records, per case, every parameter's float64 gradient norm, the small gradient tensors in float64,
and how far the float32 ORACLE itself lands from float64 (norms, elements, direction).  The GPU test then bounds the HIP
path's deviation from float64 by a stated multiple of the float32 oracle's own -- a conditioning bound instead of a
calibration.  Output: tests/golden/detector_grads_f64.pt
"""
import os
import sys

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)


def to_dtype(ex, dt):
    out = {}
    for k, v in ex.items():
        if torch.is_tensor(v):
            out[k] = v.to(dt) if v.is_floating_point() else v
        elif isinstance(v, (list, tuple)) and v and torch.is_tensor(v[0]):
            out[k] = [t.to(dt) if t.is_floating_point() else t for t in v]
        else:
            out[k] = v
    return out


def deviation(grads, ref64, skip):
    """(worst relative gradient-norm error, worst element error / tensor max on small tensors, worst 1 - cos)"""
    wn = wt = wc = 0.0
    gmax = max(float(g.norm()) for g in ref64.values() if g is not None)
    for k, r in ref64.items():
        if r is None or float(r.norm()) < 1e-4 * max(gmax, 1e-30) or skip(k):
            continue
        g = grads[k].double()
        wn = max(wn, abs(float(g.norm()) - float(r.norm())) / float(r.norm()))
        if r.numel() <= 7000:
            wt = max(wt, float((g - r).abs().max()) / float(r.abs().max()))
        wc = max(wc, 1.0 - float(torch.nn.functional.cosine_similarity(g.flatten(), r.flatten(), dim=0)))
    return dict(norm=wn, element=wt, one_minus_cos=wc)


def main():
    import re

    import detector_specs as S
    from oracle import model_ref as M

    skip = lambda k: re.search(r"middle_conv\.\d+\.conv[12]\.bias$", k) is not None  # zero true gradient (bias -> BN)
    out = {}
    for name in S.CASES:
        cfg = S.load_cfg(name)
        coder = cfg.model.bbox_head.box_coder
        ex, fmap, nv = S.make_example(name, cfg.model.bbox_head.tasks, coder.n_dim, coder.code_size)
        res = {}
        for dt in (torch.float64, torch.float32):
            m = M.DetectorRef(cfg)
            M.fill_deterministic(m)  # float32 values: both precisions start from the same numbers
            m = m.to(dt).train()
            losses = m(to_dtype(ex, dt))["loss"]
            params = dict(m.named_parameters())
            grads = torch.autograd.grad(sum(losses), list(params.values()), allow_unused=True)
            res[dt] = (dict(zip(params, grads)), [float(l) for l in losses])
        g64, l64 = res[torch.float64]
        g32, l32 = res[torch.float32]
        dev = deviation(g32, g64, skip)
        out[name] = dict(loss=l64, loss_f32=l32,
                         grad_norms={k: (float(g.norm()) if g is not None else 0.0) for k, g in g64.items()},
                         grads={k: g.clone() for k, g in g64.items() if g is not None and g.numel() <= 7000},
                         oracle_f32_deviation=dev)
        print(name, "loss64", [round(v, 5) for v in l64], "f32-oracle vs f64:", {k: "%.2e" % v for k, v in dev.items()},
              flush=True)
    torch.save(out, os.path.join(HERE, "detector_grads_f64.pt"))


if __name__ == "__main__":
    main()
