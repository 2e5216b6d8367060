#!/bin/bash
# round 3, job 3: dense stack in the model -- conv tests, e2e goldens, determinism, full bench
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r3j3
mkdir -p $O
timeout 900 python -m pytest tests/test_conv2d_gpu.py tests/test_determinism_gpu.py tests/test_e2e_gpu.py tests/test_graph_tail_gpu.py -q -m gpu --tb=short 2>&1 | tail -60 > $O/tests.log
tail -25 $O/tests.log
timeout 200 python tools/bench_conv2d.py --shape 4 200 176 128 128 3 1 1 > $O/bench_conv2d.txt 2>&1
grep shape $O/bench_conv2d.txt
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > $O/bench.json 2> $O/bench.err
python tools/show_bench.py $O/bench.json | head -24
