"""profiles/round6_pmc_spconv.json from the per-counter summaries of tools/pmc_spconv.sh: HBM traffic per launch of
the roofline kernel (read side doubled per MI355X_MICROARCH.md: gfx950's FETCH_SIZE tallies 128-B requests at 64 B)
stamped with the sha of the kernel source AND the host files that launch it (tools/stamp.py), so bench.py can refuse a stale file; MFMA-busy fraction of the dense conv
kernels.  usage: pmc_json.py <dir with round6_pmc_*.txt>"""
import json
import os
import re
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import stamp

d = sys.argv[1]
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def rows(name):
    out = []
    p = os.path.join(d, name)
    if not os.path.exists(p):
        return out
    for line in open(p):
        if line.startswith("#"):
            continue
        m = re.match(r"\s*(\d+)\s+([\d.]+)\s+([\d.]+)\s+(.*)", line)
        if m:
            out.append((int(m.group(1)), float(m.group(2)), float(m.group(3)), m.group(4).strip()))
    return out


def pick(rs, pat):
    for launches, val, us, name in rs:
        if re.search(pat, name):
            return launches, val, us, name
    return None


fetch, write, mfma = rows("round6_pmc_fetch_size.txt"), rows("round6_pmc_write_size.txt"), rows("round6_pmc_sq_valu_mfma_busy_cycles.txt")
# the roofline kernel of the default (f32, split arithmetic) step; D3D_PMC_DTYPE=bf16 for the autocast step
DT = os.environ.get("D3D_PMC_DTYPE", "fp32")
# (the trailing `false` of the mfma kernel = not the planned variant of the strided data gradients)
pat = r"spconv_fwd_mfma_kernel<64, 64, \d+, \d+, \d+, false, false>" if DT == "fp32" else \
    r"spconv_fwd_(group_kernel<64, 64, \d+, \d+, \d+>|mfma_kernel<64, 64, \d+, \d+, \d+, true, false>)"
f, w = pick(fetch, pat), pick(write, pat)
out = {"_comment": "separate rocprofv3 --kernel-trace --pmc passes over `bench.py --steps 3 --warmup 3` (tools/pmc_spconv.sh); "
                   "FETCH_SIZE / WRITE_SIZE in KiB per launch, read side x2 (MI355X_MICROARCH.md, gfx950), write side as is",
       "kernel": (re.match(r"(?:void )?(spconv_fwd_\w+)<", f[3]).group(1) + "<64,64>") if f else "spconv_fwd_group_kernel<64,64>",
       "source_sha16": stamp.source_sha16(), "source_files": list(stamp.FILES),
       "workload": {"name": "c2-20k", "frames_per_gpu": 4, "points_per_frame": 20000, "dtype": DT}}
if f and w:
    out.update(fetch_size_kib_per_launch=f[1], write_size_kib_per_launch=w[1], fetch_correction=2.0,
               launches=f[0], avg_us_under_pmc=f[2], traffic_bytes_per_launch=int((2.0 * f[1] + w[1]) * 1024))
wg = pick(fetch, r"spconv_wgrad_tile_kernel<64, 64,")
if wg:
    ww = pick(write, r"spconv_wgrad_tile_kernel<64, 64,")
    out["wgrad_64x64"] = dict(fetch_size_kib_per_launch=wg[1], write_size_kib_per_launch=ww[1] if ww else None, avg_us=wg[2])
if mfma:
    # SQ counters come as one row per instance (32 per dispatch on this part): instances = rows / dispatches, with the
    # dispatch count of the same kernel from the FETCH_SIZE pass; busy fraction = sum over instances / (1024 SIMDs x cycles)
    disp = {r[3]: r[0] for r in fetch}

    def busy_frac(row):
        inst = row[0] / disp[row[3]] if row[3] in disp and disp[row[3]] else 32.0
        return row[1] * inst / (1024.0 * row[2] * 1e-6 * 2.4e9)

    conv = [r for r in mfma if re.search(r"conv2d_halo_kernel|conv2d_wgrad3_kernel|conv2d_nhwc_kernel|conv2d_wgrad_kernel|igemm|grouped_conv|batched_gemm", r[3])]
    tot_t = sum(r[0] * r[2] for r in conv)
    out["rpn_mfma_busy"] = dict(
        kernels={r[3][:60]: round(busy_frac(r), 3) for r in conv[:8]},
        time_weighted_frac=round(sum(busy_frac(r) * r[0] * r[2] for r in conv) / tot_t, 4) if tot_t else None,
        note="SQ_VALU_MFMA_BUSY_CYCLES of the dense conv kernels (csrc/d3d_conv2d.hip; MIOpen for the head), summed over the counter's instances, divided by "
             "duration x 2.4 GHz x 1024 SIMDs: the share of SIMD cycles with the MFMA pipe busy")
    mine = pick(mfma, pat)
    if mine:
        out["fwd_mfma_busy_frac"] = round(busy_frac(mine), 4)
    wgm = pick(mfma, r"spconv_wgrad_tile_kernel<64, 64, 9")
    if wgm:
        out.setdefault("wgrad_64x64", {})["mfma_busy_frac"] = round(busy_frac(wgm), 4)
json.dump(out, open(os.path.join(d, "round6_pmc_spconv.json"), "w"), indent=1)
print("wrote", os.path.join(d, "round6_pmc_spconv.json"))
