"""300-step KITTI-SECOND training curves in several arithmetics / code paths from the SAME seed, batches and schedule
(tests/golden/train_run_spec.py) -> one JSON with every loss curve, so that "is the gap between two legs chaos or a
systematic effect?" can be answered with data:

    python tools/train_legs.py out.json [leg ...]

legs:  split            the product's f32 mode (split bf16x3 products, fused nodes)
       exact            precision.exact_f32(): exact-f32 products (sparse VALU kernels, MIOpen f32 RPN)
       split_eps        split, initial weights multiplied by (1 + 1e-6): the sensitivity of the trajectory itself
       exact_eps        exact, same perturbation
       split_unfused    split arithmetic through the per-layer sparse path (D3D_FUSE_STACK / D3D_FUSE_BLOCK off)
       split_norepack   split, weight packs rebuilt layer by layer instead of by d3d_pack_all behind the optimizer
       split_torchadam  split, torch.optim.Adam instead of d3d_adam_step
       bf16             autocast (for scale)
The CPU-oracle curve of tests/golden/train_curve_f32.pt is added as `cpu_oracle`."""
import contextlib
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))


def leg(name, steps):
    import train_run_spec as T
    from det3d_amd import build_detector, precision, repack, sparse_block, sparse_stack, synthetic
    from det3d_amd.trainer import TrainStep

    cfg = T.config()
    ds = T.dataset()
    dev = torch.device("cuda")
    ctx = precision.exact_f32() if name.startswith("exact") else contextlib.nullcontext()
    saved = (sparse_stack.FUSE_STACK, sparse_block.FUSE_BLOCK, repack.ENABLED, os.environ.get("D3D_FUSED_ADAM"))
    try:
        if name == "split_unfused":
            sparse_stack.FUSE_STACK = sparse_block.FUSE_BLOCK = False
        if name == "split_norepack":
            repack.ENABLED = False
        if name == "split_torchadam":
            os.environ["D3D_FUSED_ADAM"] = "0"
        with ctx:
            model = build_detector(cfg.model, cfg.train_cfg, cfg.test_cfg)
            model.load_state_dict(T.initial_state(cfg))
            if name.endswith("_eps"):
                with torch.no_grad():
                    for p in model.parameters():
                        p.mul_(1.0 + 1e-6)
            model = model.to(dev).train()
            step = TrainStep(model, total_steps=T.STEPS, lr_max=cfg.lr_config.lr_max, wd=cfg.optimizer.VALUE.wd,
                             moms=cfg.lr_config.moms, div_factor=cfg.lr_config.div_factor, pct_start=cfg.lr_config.pct_start,
                             grad_clip=cfg.optimizer_config.grad_clip.max_norm,
                             autocast_dtype=torch.bfloat16 if name == "bf16" else None)
            batches = [synthetic.make_example(ds.device_batch(i, T.BATCH, dev), synthetic.KITTI_CAR) for i in range(T.N_BATCHES)]
            losses, gnorms = [], []
            for i in range(steps):
                losses.append(step(batches[i % T.N_BATCHES]))
                gnorms.append(step.last["grad_norm"])
            losses = [float(v) for v in torch.stack(losses).cpu()]
            gnorms = [float(v) for v in torch.stack([torch.as_tensor(g, device=dev).float() for g in gnorms]).cpu()]
    finally:
        sparse_stack.FUSE_STACK, sparse_block.FUSE_BLOCK, repack.ENABLED = saved[:3]
        if saved[3] is None:
            os.environ.pop("D3D_FUSED_ADAM", None)
        else:
            os.environ["D3D_FUSED_ADAM"] = saved[3]
    return dict(losses=losses, grad_norms=gnorms)


def main():
    import train_run_spec as T

    out = sys.argv[1]
    legs = sys.argv[2:] or ["split", "exact", "split_eps", "exact_eps", "split_unfused", "split_norepack", "split_torchadam"]
    res = {}
    path = os.path.join(ROOT, "tests", "golden", "train_curve_f32.pt")
    if os.path.exists(path):
        cpu = torch.load(path, weights_only=False)
        res["cpu_oracle"] = dict(losses=cpu["losses"], grad_norms=cpu["grad_norms"])
    for name in legs:
        res[name] = leg(name, T.STEPS)
        l = res[name]["losses"]
        print("%-16s loss %.3f -> step 10 %.4f, 50 %.4f, 100 %.4f, 200 %.4f, mean of last 20 %.4f" % (
            name, l[0], l[10], l[50], l[100], l[200], sum(l[-20:]) / 20), flush=True)
        json.dump(res, open(out, "w"))
    if "cpu_oracle" in res:
        l = res["cpu_oracle"]["losses"]
        print("%-16s loss %.3f -> step 10 %.4f, 50 %.4f, 100 %.4f, 200 %.4f, mean of last 20 %.4f" % (
            "cpu_oracle", l[0], l[10], l[50], l[100], l[200], sum(l[-20:]) / 20))


if __name__ == "__main__":
    main()
