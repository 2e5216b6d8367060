#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r3j16
mkdir -p $O
VARIANTS=0,8 timeout 300 python tools/bench_conv2d.py --shape 4 200 176 128 128 3 1 1 > $O/bench_conv2d.txt 2>&1
grep shape $O/bench_conv2d.txt
D3D_CONV2D_VARIANT=8 timeout 300 python -m pytest tests/test_conv2d_gpu.py -q -m gpu --tb=short -x 2>&1 | tail -3
