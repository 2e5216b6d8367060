"""sha of everything that decides how the roofline kernel is launched: the kernel source, its internal header, and the
host files that choose its tile shapes / arguments.  profiles/round6_pmc_spconv.json and round6_rocprof_spconv.json carry
it; bench.py refuses a committed measurement whose stamp differs from the tree it runs from."""
import hashlib
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FILES = ("det3d_amd/csrc/d3d_spconv.hip", "det3d_amd/csrc/d3d_spconv_shared.h", "det3d_amd/csrc/d3d_spconv_internal.h", "det3d_amd/csrc/d3d_fold.h",
         "det3d_amd/spconv.py", "det3d_amd/sparse_stack.py", "det3d_amd/sparse_block.py")


def source_sha16(root=ROOT):
    h = hashlib.sha256()
    for f in FILES:
        with open(os.path.join(root, f), "rb") as fh:
            h.update(f.encode() + b"\0" + fh.read())
    return h.hexdigest()[:16]


if __name__ == "__main__":
    print(source_sha16())
