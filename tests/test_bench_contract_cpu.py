"""bench.py contract pieces that do not need a GPU: the algorithmic-bytes formula of SURVEY.md 8(d), the flags the
driver passes, the recorded bench line's schema, and that the script refuses to run without a GPU."""
import json
import os
import subprocess
import sys

import pytest
import torch

from conftest import ROOT

sys.path.insert(0, ROOT)


def test_algorithmic_bytes_formula():
    import bench

    # B_alg = P*Cin*e + P*8 + Nout*Cout*e + kvol*Cin*Cout*e
    assert bench.b_alg("fwd", 64, 64, 1000, 13000, 27, 2) == 13000 * 64 * 2 + 13000 * 8 + 1000 * 64 * 2 + 27 * 64 * 64 * 2
    assert bench.b_alg("dgrad", 32, 64, 10, 20, 27, 4) == 20 * 32 * 4 + 20 * 8 + 10 * 64 * 4 + 27 * 32 * 64 * 4
    assert bench.HBM_PEAK_GBS == 8000.0


def test_driver_flags_parse(monkeypatch):
    import bench

    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "8", "--steps", "7", "--warmup", "3"])
    a = bench.parse()
    assert (a.gpus, a.steps, a.warmup, a.batch, a.points, a.dtype) == (8, 7, 3, None, None, "fp32")
    # --batch / --points default to the workload's configured values (the headline: 4 frames x 20000 points per GPU)
    assert (bench.WORKLOADS[a.workload]["batch"], bench.WORKLOADS[a.workload]["points"]) == (4, 20000)
    monkeypatch.setattr(sys, "argv", ["bench.py"])
    a = bench.parse()
    assert a.gpus == 1 and a.steps >= 10 and a.warmup >= 1 and not a.no_gc_freeze


@pytest.mark.parametrize("name", ["round1_bench_v5.json", "round2_bench_final.json", "round3_bench_driver_flags.json",
                                  "round4_bench_driver_flags.json"])
def test_recorded_bench_line_has_the_contract_fields(name):
    d = json.loads(open(os.path.join(ROOT, "profiles", name)).read().strip().splitlines()[-1])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["unit"] == "frames/s" and d["higher_is_better"] is True and d["scaling"] == "weak"
    assert d["vs_baseline"] is None and d["data"] == "synthetic" and "workload" in d["config"]
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3
    assert r["traffic"] is None or r["traffic"] > 0
    c = d["cpu_baseline"]
    assert c["kind"] in ("port", "reference") and c["cores"] >= 1 and c["value"] > 0 and "sample" in c
    assert abs(d["value"] - d["config"]["frames_per_gpu"] * d["n_gpus"] * 1e3 / d["ms_per_step"]) / d["value"] < 1e-2
    if name.startswith("round4"):
        import math

        # round 4: the dense 3x3 layers on the halo-patch / three-taps kernels, distinct clouds for four batches, the
        # first loss next to the final one, >= 5 timed CPU steps at the best thread count
        assert d["dtype"].startswith("f32") and d["precision"]["fp32_within_tolerance"] is True and d["value"] > 540
        c = d["config"]
        assert math.isfinite(c["first_loss"]) and c["final_loss"] < c["first_loss"] and c["steps_before_final_loss"] >= 25
        assert d["rpn_mfma"]["frac"] > 0.37 and d["rpn_mfma"]["gflop_per_step_left_on_miopen"] == 0.0
        best = d["cpu_baseline"]["runs"][str(d["cpu_baseline"]["cores"])]
        assert best["steps"] >= 5
        for w in ("c2-60k", "c4", "c5", "c3"):
            assert d["workloads"][w]["value"] > 0 and d["workloads"][w]["final_loss"] < d["workloads"][w]["first_loss"]
    if name.startswith("round3"):
        import math

        # round 3: the headline IS the precision that meets the 1e-3 contract; bf16 autocast is the second line
        assert d["dtype"].startswith("f32") and d["precision"]["fp32_within_tolerance"] is True
        assert d["precision"]["fp32_max_abs_err"] < 1e-3 < d["precision"]["bf16_max_abs_err"]
        assert d["bf16"]["dtype"] == "bf16" and d["bf16"]["value"] > d["value"] > 450
        assert math.isfinite(d["config"]["final_loss"]) and math.isfinite(d["config"]["final_grad_norm"])
        assert r["kernel"] == "spconv_fwd_mfma_kernel<64,64>" and r["traffic"] > 0 and r["traffic_measured_in_run"] is False
        m = d["rpn_mfma"]
        assert m["bound"] == "mfma" and m["convs_on_library_kernels"] == 7 and m["gflop_per_step_left_on_miopen"] == 0.0
        assert abs(m["frac"] - m["mfma_issued_TFLOPs"] / m["peak_TFLOPs"]) < 1e-3
        # every single-GPU configuration at its configured batch size
        assert {"c2-60k", "c4", "c5", "c3"} <= set(d["workloads"])
        assert "16 frames/GPU" in d["workloads"]["c4"]["workload"] and "6 frames/GPU" in d["workloads"]["c5"]["workload"]
        assert all(math.isfinite(w["final_loss"]) for w in d["workloads"].values())
    if name.startswith("round2"):
        import math

        # round 2: a finite training state is part of the line, the precision question is answered next to it
        assert math.isfinite(d["config"]["final_loss"]) and math.isfinite(d["config"]["final_grad_norm"])
        assert d["fp32"]["dtype"] == "fp32" and d["fp32"]["value"] > 0
        assert d["precision"]["fp32_within_tolerance"] is True
        assert r["kernel"].startswith("spconv_fwd_") and r["traffic_measured_in_run"] is False
        assert {"c2-60k", "c4"} <= set(d["workloads"]) and all(math.isfinite(w["final_loss"]) for w in d["workloads"].values())


@pytest.mark.skipif(torch.cuda.is_available(), reason="only meaningful without a GPU")
def test_bench_refuses_to_run_without_gpu():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "1", "--warmup", "0",
                          "--no-cpu-baseline"], capture_output=True, text=True, timeout=300)
    assert out.returncode != 0 and "no CPU fallback" in (out.stderr + out.stdout)


def test_bench_refuses_a_world_size_that_is_not_gpus():
    """a line whose n_gpus differs from --gpus must not be printed (round 5: `--gpus 8` silently ran one rank): under a
    launcher's environment with another WORLD_SIZE the run is refused before anything else happens"""
    env = dict(os.environ, RANK="0", LOCAL_RANK="0", WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT="29999")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4", "--steps", "1", "--warmup", "0",
                          "--no-cpu-baseline"], env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode != 0 and "--gpus 4 but WORLD_SIZE=2" in (out.stderr + out.stdout)


def test_rank_core_shares_are_disjoint_and_cover():
    import bench

    if not hasattr(os, "sched_getaffinity"):
        pytest.skip("no sched_setaffinity here")
    before = os.sched_getaffinity(0)
    try:
        cores = sorted(before)
        world = max(1, len(cores) // 4)
        if world < 2:
            pytest.skip("needs >= 8 cores for two shares of 4")
        shares = []
        for r in range(world):
            os.sched_setaffinity(0, before)
            shares.append(bench.pin_rank_cores(r, world))
        assert all(s is not None and len(s) == len(cores) // world for s in shares)
        flat = [c for s in shares for c in s]
        assert len(flat) == len(set(flat)) and set(flat) <= set(cores)
        os.sched_setaffinity(0, before)
        assert bench.pin_rank_cores(0, 1) is None and bench.pin_rank_cores(0, len(cores)) is None  # one rank / shares < 4 cores
    finally:
        os.sched_setaffinity(0, before)
