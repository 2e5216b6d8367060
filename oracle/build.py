"""Build the CPU oracle (TEST INFRASTRUCTURE, never imported by det3d_amd).

  oracle/libd3d_oracle.so   gcc build of this repo's C restatements
  oracle/_ref/d3d_ref*.so   the REFERENCE's own C++ (voxelizer twin, detectron2
                            rotated IoU) compiled from /root/reference when that
                            tree is present (this container); the GPU box only
                            ever uses the prebuilt file.
"""
import os
import subprocess
import sys
import sysconfig

HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.path.join(HERE, "libd3d_oracle.so")
REF_DIR = os.path.join(HERE, "_ref")
REFERENCE_ROOT = "/root/reference"
C_SOURCES = ["voxel_oracle.c", "nms_oracle.c", "iou3d_oracle.c"]


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build_oracle(force=False):
    srcs = [os.path.join(HERE, s) for s in C_SOURCES]
    if force or _stale(LIB, srcs):
        cmd = ["gcc", "-O2", "-std=c11", "-ffp-contract=off", "-fPIC", "-shared", "-o", LIB] + srcs + ["-lm"]
        subprocess.run(cmd, check=True)
    return LIB


def build_oracle_sanitized(force=False):
    """AddressSanitizer + UBSan build of the same C restatements (oracle/libd3d_oracle_asan.so): the checker the
    GPU parity tests trust is itself run under sanitizers by tests/test_oracle_sanitizers_cpu.py."""
    out = os.path.join(HERE, "libd3d_oracle_asan.so")
    srcs = [os.path.join(HERE, s) for s in C_SOURCES]
    if force or _stale(out, srcs):
        cmd = ["gcc", "-O1", "-g", "-std=c11", "-ffp-contract=off", "-fPIC", "-shared", "-fsanitize=address,undefined",
               "-fno-sanitize-recover=undefined", "-fno-omit-frame-pointer", "-o", out] + srcs + ["-lm"]
        subprocess.run(cmd, check=True)
    return out


def ref_module_path():
    suffix = sysconfig.get_config_var("EXT_SUFFIX") or ".so"
    return os.path.join(REF_DIR, "d3d_ref" + suffix)


def build_ref(force=False):
    """Compile the reference's own C++ into oracle/_ref (needs /root/reference)."""
    out = ref_module_path()
    if not os.path.isdir(REFERENCE_ROOT):
        return out if os.path.exists(out) else None
    import pybind11

    os.makedirs(REF_DIR, exist_ok=True)
    src = os.path.join(HERE, "ref_wrap.cc")
    deps = [src,
            os.path.join(REFERENCE_ROOT, "det3d/ops/point_cloud/point_cloud_ops.h"),
            os.path.join(REFERENCE_ROOT, "det3d/ops/box2d_iou_rotated/src/box_iou_rotated_utils.h")]
    if force or _stale(out, deps):
        cmd = ["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-w",
               "-I", REFERENCE_ROOT, "-I", pybind11.get_include(),
               "-I", sysconfig.get_paths()["include"], src, "-o", out]
        subprocess.run(cmd, check=True)
    return out


def ref_iou3d_module_path():
    suffix = sysconfig.get_config_var("EXT_SUFFIX") or ".so"
    return os.path.join(REF_DIR, "d3d_ref_iou3d" + suffix)


def build_ref_iou3d(force=False):
    """Host build of the reference's iou3d DEVICE functions (iou3d_kernel.cu:13-304).  The part of the
    file before its <<<>>> launchers is written to a temporary directory and #included by
    ref_wrap_iou3d.cc; nothing of it enters the repo."""
    out = ref_iou3d_module_path()
    cu = os.path.join(REFERENCE_ROOT, "det3d/ops/iou3d/src/iou3d_kernel.cu")
    if not os.path.exists(cu):
        return out if os.path.exists(out) else None
    import tempfile

    import pybind11

    os.makedirs(REF_DIR, exist_ok=True)
    src = os.path.join(HERE, "ref_wrap_iou3d.cc")
    if force or _stale(out, [src, cu]):
        text = open(cu).read()
        cut = text.index("void boxesoverlapLauncher")
        with tempfile.TemporaryDirectory() as tmp:
            with open(os.path.join(tmp, "iou3d_dev.inc"), "w") as f:
                f.write(text[:cut])
            cmd = ["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-w", "-ffp-contract=off",
                   "-DIOU3D_DEV_INC=\"iou3d_dev.inc\"", "-I", tmp, "-I", pybind11.get_include(),
                   "-I", sysconfig.get_paths()["include"], src, "-o", out]
            subprocess.run(cmd, check=True)
    return out


def build(force=False):
    return build_oracle(force), build_ref(force), build_ref_iou3d(force)


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
