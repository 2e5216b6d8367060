#!/bin/bash
# round 3, job 8: wgrad staging without divisions; the bench exactly as the driver runs it
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r3j8
mkdir -p $O
timeout 600 python -m pytest tests/test_conv2d_gpu.py tests/test_e2e_gpu.py -q -m gpu --tb=short -x 2>&1 | tail -8 > $O/tests.log
tail -3 $O/tests.log
timeout 200 python tools/bench_conv2d.py --shape 4 200 176 128 128 3 1 1 > $O/bench_conv2d.txt 2>&1
grep shape $O/bench_conv2d.txt
T0=$(date +%s)
timeout 1500 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err
echo "bench wall $(( $(date +%s) - T0 )) s"
python tools/show_bench.py $O/bench.json workloads | cut -c1-700
