#!/bin/bash
# round 3, job 5: split arithmetic in the f32 sparse kernels -- parity, goldens, the fp32 line
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r3j5
mkdir -p $O
timeout 900 python -m pytest tests/test_spconv_gpu.py tests/test_determinism_gpu.py tests/test_e2e_gpu.py -q -m gpu --tb=short 2>&1 | tail -40 > $O/tests.log
tail -12 $O/tests.log
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > $O/bench.json 2> $O/bench.err
python tools/show_bench.py $O/bench.json | grep -E "value|fp32|precision"
bash tools/profile_step.sh $O/kernel_stats_fp32.txt --dtype fp32 > $O/prof_fp32.log 2>&1
head -30 $O/kernel_stats_fp32.txt | cut -c1-150
