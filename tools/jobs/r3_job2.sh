#!/bin/bash
# round 3, job 2: dense conv kernels -- parity and micro-benchmark; fold without release fences
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r3j2
mkdir -p $O
timeout 600 python -m pytest tests/test_conv2d_gpu.py tests/test_determinism_gpu.py -q -m gpu --tb=short -x 2>&1 | tail -40 > $O/tests.log
tail -15 $O/tests.log
timeout 300 python tools/bench_conv2d.py > $O/bench_conv2d.txt 2>&1
cat $O/bench_conv2d.txt | grep shape
timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-extras > $O/bench.json 2> $O/bench.err
python tools/show_bench.py $O/bench.json | head -8
