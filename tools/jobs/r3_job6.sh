#!/bin/bash
# round 3, job 6: conv variants, pack / reduce kernels, tests
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r3j6
mkdir -p $O
timeout 600 python -m pytest tests/test_conv2d_gpu.py tests/test_spconv_gpu.py tests/test_e2e_gpu.py -q -m gpu --tb=short -x 2>&1 | tail -30 > $O/tests.log
tail -6 $O/tests.log
VARIANTS=0,1,2,3 timeout 300 python tools/bench_conv2d.py --shape 4 200 176 128 128 3 1 1 > $O/bench_conv2d.txt 2>&1
grep shape $O/bench_conv2d.txt
D3D_CONV2D_VARIANT=3 timeout 300 python -m pytest tests/test_conv2d_gpu.py -q -m gpu --tb=short -x 2>&1 | tail -3
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > $O/bench.json 2> $O/bench.err
python tools/show_bench.py $O/bench.json | grep -E "value|host|fp32"
