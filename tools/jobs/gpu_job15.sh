#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r2j15; mkdir -p $O
export TMPDIR=/tmp
echo "== launch overhead"; mkdir -p tools/bin; [ -x tools/bin/launch_bench ] || hipcc --offload-arch=gfx950 -O2 tools/launch_bench.hip -o tools/bin/launch_bench; ./tools/bin/launch_bench; python tools/ctypes_bench.py 2>&1 | tail -6
echo "== graph diag"; timeout 900 python tools/diag_graph.py > $O/diag_graph.txt 2>&1; grep -c "step" $O/diag_graph.txt; grep "dtype\|<<<" $O/diag_graph.txt | cut -c1-200 | head -60
