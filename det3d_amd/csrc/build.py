"""Build libd3d_hip.so (gfx950) in-tree with hipcc.  Used by __graft_entry__.build().

    python -m det3d_amd.csrc.build [--force]

hipcc cross-compiles without a GPU.  The .so is git-ignored but travels with the
gpurun snapshot, so the GPU box uses exactly what was built here.
"""
import concurrent.futures
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.dirname(HERE)
LIB = os.path.join(PKG, "libd3d_hip.so")
OBJ_DIR = os.path.join(HERE, "build")

SOURCES = [
    "d3d_core.hip",
    "d3d_voxelize.hip",
    "d3d_rulebook.hip",
    "d3d_rbchain.hip",
    "d3d_spconv.hip",
    "d3d_spconv_pairs.hip",
    "d3d_wgrad.hip",
    "d3d_stack.hip",
    "d3d_bn.hip",
    "d3d_conv2d.hip",
    "d3d_head.hip",
    "d3d_optim.hip",
    "d3d_pfn.hip",
    "d3d_dense.hip",
    "d3d_assign.hip",
    "d3d_nms.hip",
    "d3d_decode.hip",
    "d3d_iou3d.hip",
    "d3d_loss.hip",
]

FLAGS = [
    "--offload-arch=gfx950",
    "-O3",
    "-std=c++17",
    "-fPIC",
    "-ffp-contract=off",  # bit-exact f32 cell arithmetic / no surprise FMAs
    "-fhip-fp32-correctly-rounded-divide-sqrt",
    "-Wall",
    "-Wno-unused-function",
]


def _hipcc():
    for c in ("/opt/rocm/bin/hipcc", "hipcc"):
        if os.path.isabs(c) and os.path.exists(c):
            return c
    return "hipcc"


def _newest_header_mtime():
    m = 0.0
    for d in (HERE, os.path.join(os.path.dirname(PKG), "include")):
        for f in os.listdir(d):
            if f.endswith(".h"):
                m = max(m, os.path.getmtime(os.path.join(d, f)))
    return m


def _compile(src, force, hdr_m):
    obj = os.path.join(OBJ_DIR, src.replace(".hip", ".o"))
    sp = os.path.join(HERE, src)
    if (not force and os.path.exists(obj) and os.path.getmtime(obj) > os.path.getmtime(sp)
            and os.path.getmtime(obj) > hdr_m):
        return obj, False
    cmd = [_hipcc()] + FLAGS + ["-c", sp, "-o", obj]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("hipcc failed for %s:\n%s\n%s" % (src, r.stdout, r.stderr))
    if r.stderr.strip():
        sys.stderr.write(r.stderr)
    return obj, True


def build(force=False, verbose=True):
    os.makedirs(OBJ_DIR, exist_ok=True)
    srcs = [s for s in SOURCES if os.path.exists(os.path.join(HERE, s))]
    hdr_m = _newest_header_mtime()
    with concurrent.futures.ThreadPoolExecutor(max_workers=min(8, len(srcs))) as ex:
        results = list(ex.map(lambda s: _compile(s, force, hdr_m), srcs))
    objs = [o for o, _ in results]
    changed = any(c for _, c in results)
    if changed or force or not os.path.exists(LIB):
        cmd = [_hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("link failed:\n%s\n%s" % (r.stdout, r.stderr))
        if verbose:
            print("built", LIB)
    elif verbose:
        print("up to date:", LIB)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
