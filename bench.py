#!/usr/bin/env python
"""Benchmark of the Det3D hot path on MI355X: training frames/sec of KITTI-car SECOND
(BASELINE.json configs[1]: kitti_car_vfev3_spmiddlefhd_rpn1_mghead, batch 4 per GPU, bf16).

    python bench.py --gpus 1 --steps 20 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

One step = one pass of the whole hot path over one batch of synthetic KITTI-range clouds that
are already resident in HBM: device voxelizer + fused mean VFE -> anchor target assignment (HIP) -> SpMiddleFHD (HIP rulebooks,
sparse conv fwd) -> RPN / head (MIOpen via PyTorch-ROCm, bf16 autocast) -> losses -> backward
(HIP dgrad / wgrad) -> gradient mean over ranks (RCCL) -> clip -> true-WD Adam.  Rank 0 prints
ONE JSON line; `roofline` describes the hand-written sparse-conv gather kernel
(spconv_fwd_mfma_kernel, timed with HIP events on its launch stream inside the timed region),
`cpu_baseline` is the CPU oracle port (reference CPU voxelizer + PyTorch-CPU model) timed on
this box's host cores on a bounded sample.
"""
import argparse
import json
import os
import sys
import time

# PyTorch-ROCm only hands channels-last tensors to MIOpen as NHWC when this is set before the
# first convolution; otherwise it converts to NCHW and MIOpen transposes again internally
if "--no-channels-last" not in sys.argv:
    os.environ.setdefault("PYTORCH_MIOPEN_SUGGEST_NHWC", "1")
    os.environ.setdefault("PYTORCH_MIOPEN_SUGGEST_NHWC_BATCHNORM", "1")

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=4, help="frames per GPU (config samples_per_gpu)")
    ap.add_argument("--points", type=int, default=20000, help="points per synthetic frame")
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "fp32"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-launch-log", action="store_true")
    ap.add_argument("--cached-targets", action="store_true",
                    help="use the host-precomputed anchor targets instead of d3d_assign_targets in the step")
    ap.add_argument("--no-gc-freeze", action="store_true",
                    help="do not gc.freeze() after warm-up (shows the cost of full CPython collections)")
    ap.add_argument("--no-channels-last", action="store_true",
                    help="NCHW dense()/RPN activations (default: channels-last, written directly by d3d_dense_scatter)")
    ap.add_argument("--cpu-baseline-only", action="store_true", help=argparse.SUPPRESS)
    return ap.parse_args()


def b_alg(kind, cin, cout, n_out, pairs, kvol, esz):
    """algorithmic bytes of one sparse-conv launch, SURVEY.md 8(d):
    P*Cin*e (gathered rows) + P*8 (pair indices) + Nout*Cout*e (rows written once) + weights."""
    return pairs * cin * esz + pairs * 8 + n_out * cout * esz + kvol * cin * cout * esz


def _cpu_threads():
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    # the reference's CPU path is DataLoader workers + a PyTorch-CPU model; past ~16 threads the
    # small per-offset mm / index_add_ ops of the sparse layers only lose to OpenMP barriers
    return max(1, min(avail, int(os.environ.get("D3D_CPU_THREADS", "16"))))


def cpu_baseline_worker(batch, points):
    """runs in a child process (bench.py --cpu-baseline-only): reference CPU path (oracle port)
    on the host cores, 1 one-frame warm-up + 1 timed step of `batch` frames."""
    from det3d_amd import Config, synthetic
    from oracle import model_ref as M

    cores = _cpu_threads()
    torch.set_num_threads(cores)
    cfg = Config.fromfile(os.path.join(ROOT, "det3d_amd", "configs", "kitti_car_second.py"))
    torch.manual_seed(0)
    m = M.SecondRef(cfg)
    m.train()
    ds = synthetic.KittiCarSynthetic(n_points=points, n_seeds=batch + 1)
    opt = torch.optim.Adam(m.parameters(), lr=3e-4, betas=(0.9, 0.99))

    def one(step, bsz):
        hb = ds.host_batch(step, bsz)
        ex = M.cpu_example(hb, synthetic.KITTI_CAR)
        loss = sum(m(ex)["loss"])
        opt.zero_grad()
        loss.backward()
        torch.nn.utils.clip_grad_norm_(m.parameters(), 35)
        opt.step()

    one(0, 1)  # primitive creation / page faults, untimed
    t0 = time.perf_counter()
    one(1, batch)
    dt = time.perf_counter() - t0
    print(json.dumps(dict(value=round(batch / dt, 4), unit="frames/s", cores=cores, kind="port",
                          sample="1 training step of %d frames x %d pts (oracle CPU voxelizer + "
                                 "PyTorch-CPU fp32 SECOND fwd/bwd/clip/Adam) after a 1-frame warm-up: "
                                 "%.1f s on %d threads" % (batch, points, dt, cores))))


def cpu_baseline(batch, points, budget_s=240):
    import subprocess

    try:
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "--cpu-baseline-only",
                            "--batch", str(batch), "--points", str(points)],
                           capture_output=True, text=True, timeout=budget_s)
        line = [l for l in r.stdout.splitlines() if l.startswith("{")]
        if r.returncode == 0 and line:
            return json.loads(line[-1])
        return {"error": "rc=%d %s" % (r.returncode, r.stderr[-300:])}
    except subprocess.TimeoutExpired:
        return {"error": "cpu baseline exceeded its %d s budget" % budget_s}


def main():
    args = parse()
    if args.cpu_baseline_only:
        return cpu_baseline_worker(args.batch, args.points)
    from det3d_amd import Config, build_detector, synthetic
    from det3d_amd import dist as D
    from det3d_amd import spconv
    from det3d_amd import trainer
    from det3d_amd.trainer import TrainStep

    t_start = time.perf_counter()
    rank, local_rank, world = D.init_dist()
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the det3d_amd hot path has no CPU fallback")
    dev = torch.device("cuda", local_rank % torch.cuda.device_count())
    torch.cuda.set_device(dev)
    # MIOpen find mode for the dense RPN convs (+3 % over immediate mode, < 20 s on a cold
    # perf-db cache); D3D_MIOPEN_FIND=0 falls back to immediate mode
    torch.backends.cudnn.benchmark = bool(int(os.environ.get("D3D_MIOPEN_FIND", "1")))

    cfg = Config.fromfile(os.path.join(ROOT, "det3d_amd", "configs", "kitti_car_second.py"))
    torch.manual_seed(0)
    model = build_detector(cfg.model, cfg.train_cfg, cfg.test_cfg)
    model = model.to(dev).train()
    model.backbone.dense_channels_last = not args.no_channels_last
    D.broadcast_params(model)

    n_seeds = 16
    ds = synthetic.KittiCarSynthetic(n_points=args.points, n_seeds=n_seeds)
    n_batches = max(n_seeds // args.batch, 1)
    # inputs resident in HBM before the timed region: distinct batches per rank, cycled
    batches = [ds.device_batch(i, args.batch, dev, rank, world) for i in range(n_batches)]
    vox_cfg = synthetic.KITTI_CAR

    total_steps = args.steps + args.warmup
    step_fn = TrainStep(model, total_steps=max(total_steps, 100), lr_max=cfg.lr_config.lr_max,
                        wd=cfg.optimizer.VALUE.wd, moms=cfg.lr_config.moms,
                        div_factor=cfg.lr_config.div_factor, pct_start=cfg.lr_config.pct_start,
                        grad_clip=cfg.optimizer_config.grad_clip.max_norm,
                        autocast_dtype=torch.bfloat16 if args.dtype == "bf16" else None)

    records = []  # (kind, cin, cout, n_out, rulebook, ms) of every timed sparse-conv launch

    def drain_launch_log():
        """resolve the HIP events of the previous step (complete: make_example just synced)"""
        if spconv.LAUNCH_LOG:
            for kind, cin, cout, n_out, rb, e0, e1 in spconv.LAUNCH_LOG:
                records.append((kind, cin, cout, n_out, rb, e0.elapsed_time(e1)))
            spconv.LAUNCH_LOG.clear()

    def one_step(i):
        ex = synthetic.make_example(batches[i % n_batches], vox_cfg, fused_mean=True,
                                    device_targets=not args.cached_targets)
        drain_launch_log()
        return step_fn(ex)

    def note(msg):
        if rank == 0:
            print("[bench %.1fs] %s" % (time.perf_counter() - t_start, msg), file=sys.stderr, flush=True)

    for i in range(args.warmup):
        one_step(i)
        torch.cuda.synchronize()
        note("warmup step %d done" % i)
    torch.cuda.synchronize()
    # The step allocates a few thousand short-lived Python containers; CPython's generational collector
    # then walks every tracked object of the process (torch + numpy: ~10^6) in a full collection every few
    # steps, 40-80 ms each, which showed up as a bimodal 8.5 / 12.5 ms step.  Freezing what exists after
    # warm-up (gc stays enabled) makes later collections scan only what the loop itself creates.
    if not args.no_gc_freeze:
        trainer.freeze_gc()
    D.barrier()

    # HIP-event timing of individual launches perturbs the stream (~0.4 ms per event pair on
    # ROCm), so inside the timed region only the dominant kernel (64->64 fwd/dgrad) of the LAST
    # TWO steps is timed; the full per-kernel table comes from one extra untimed step below.
    sample_from = max(args.steps - 2, 0)
    spconv.LAUNCH_FILTER = lambda kind, cin, cout: kind != "wgrad" and cin == 64 and cout == 64
    torch.cuda.synchronize()
    ms0 = torch.cuda.memory_stats(dev)
    t0 = time.perf_counter()
    for i in range(args.steps):
        if i == sample_from and not args.no_launch_log:
            spconv.LAUNCH_LOG = []
        loss = one_step(args.warmup + i)
    torch.cuda.synchronize()
    D.barrier()
    elapsed = time.perf_counter() - t0
    ms1 = torch.cuda.memory_stats(dev)
    # hipMalloc / hipFree inside the timed region would stall the stream: report them
    alloc_delta = {k: int(ms1.get(k, 0) - ms0.get(k, 0)) for k in ("num_device_alloc", "num_device_free", "num_alloc_retries")}
    alloc_delta["reserved_MB"] = int(ms1.get("reserved_bytes.all.current", 0) >> 20)
    note("timed region done: %.3f s; allocator during it: %s" % (elapsed, alloc_delta))
    drain_launch_log()
    log = list(records)
    table_log = []
    if not args.no_launch_log:
        spconv.LAUNCH_FILTER = None
        spconv.LAUNCH_LOG = [] if rank == 0 else None
        del records[:]
        one_step(args.warmup + args.steps)  # untimed, all ranks (the step has a collective):
        torch.cuda.synchronize()            # every sparse-conv launch event-timed on rank 0
        drain_launch_log()
        table_log = list(records)
    spconv.LAUNCH_LOG = None
    spconv.LAUNCH_FILTER = None
    t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
    if world > 1:
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
    elapsed = float(t.item())
    frames = args.steps * args.batch * world
    value = frames / elapsed

    out = {
        "metric": "training frames/sec (KITTI-car SECOND, voxelize->sparse conv->RPN->head->loss->bwd->step)",
        "value": round(value, 2), "unit": "frames/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": round(1e3 * elapsed / max(args.steps, 1), 3),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": args.dtype, "data": "synthetic",
        "config": {"workload": "kitti_car_vfev3_spmiddlefhd_rpn1_mghead: %d frames/GPU x %d pts "
                               "(synthetic KITTI-car-range clouds, ~16k voxels/frame), fwd+bwd+Adam"
                               % (args.batch, args.points),
                   "frames_per_gpu": args.batch, "points_per_frame": args.points,
                   "parallelism": "dp%d" % world, "final_loss": round(float(loss), 4)},
    }

    def group(entries):
        esz = 2 if args.dtype == "bf16" else 4
        groups = {}
        for kind, cin, cout, n_out, rb, ms in entries:
            P = pair_cache.setdefault(id(rb), None)
            if P is None:
                P = pair_cache[id(rb)] = rb.pairs()
            if kind == "wgrad":
                key, byts = "spconv_wgrad %dx%d" % (cin, cout), 0.0
            else:
                mfma = cin % 16 == 0 and cout % 16 == 0
                key = ("spconv_fwd_mfma_kernel<%d,%d>" if mfma else "spconv_fwd_generic<%d,%d>") % (cin, cout)
                byts = b_alg(kind, cin, cout, n_out, P, rb.kvol, esz)
            g = groups.setdefault(key, dict(ms=0.0, bytes=0.0, launches=0))
            g["ms"] += ms
            g["launches"] += 1
            g["bytes"] += byts
        return groups

    pair_cache = {}
    if rank == 0 and log:
        g = group(log)["spconv_fwd_mfma_kernel<64,64>"]
        ach = g["bytes"] / (g["ms"] * 1e-3) / 1e9
        out["roofline"] = {"kernel": "spconv_fwd_mfma_kernel<64,64>", "bound": "hbm",
                           "achieved": round(ach, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                           "frac": round(ach / HBM_PEAK_GBS, 4), "traffic": None,
                           "avg_launch_us": round(1e3 * g["ms"] / g["launches"], 2),
                           "alg_bytes_per_launch": int(g["bytes"] / g["launches"]),
                           "launches_timed": g["launches"],
                           "traffic_source": None,
                           "note": "HIP events on the launch stream, 64->64 fwd+dgrad launches of the "
                                   "last 2 timed steps; B_alg per SURVEY.md 8(d)"}
    if rank == 0 and "roofline" in out:
        # HBM traffic needs PMC passes (rocprofv3 --pmc cannot run inside this process): take the
        # committed measurement of the same workload if there is one
        pmc_path = os.path.join(ROOT, "profiles", "round1_pmc_spconv.json")
        try:
            pmc = json.load(open(pmc_path))
            w = pmc["workload"]
            if (w["frames_per_gpu"], w["points_per_frame"], w["dtype"]) == (args.batch, args.points, args.dtype):
                out["roofline"]["traffic"] = pmc["traffic_bytes_per_launch"]
                out["roofline"]["traffic_source"] = "profiles/round1_pmc_spconv.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, read side x2 per MI355X_MICROARCH.md)"
        except (OSError, KeyError, ValueError):
            pass
    if rank == 0 and table_log:
        table = {}
        groups = group(table_log)
        for k, g in groups.items():
            row = dict(launches_per_step=g["launches"], avg_us=round(1e3 * g["ms"] / g["launches"], 2),
                       ms_per_step=round(g["ms"], 4))
            if g["bytes"]:
                row["achieved_GBs"] = round(g["bytes"] / (g["ms"] * 1e-3) / 1e9, 1)
                row["B_alg_MB_per_launch"] = round(g["bytes"] / g["launches"] / 1e6, 2)
            table[k] = row
        out["kernels"] = table
        out["sparse_conv_ms_per_step"] = round(sum(g["ms"] for g in groups.values()), 3)

    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        note("cpu baseline (child process) ...")
        out["cpu_baseline"] = cpu_baseline(args.batch, args.points)
        if "value" in out["cpu_baseline"]:
            out["config"]["gpu_over_cpu"] = round(value / out["cpu_baseline"]["value"], 1)
        note("cpu baseline done")

    if rank == 0:
        print(json.dumps(out))
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
