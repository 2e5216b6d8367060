#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r2j5; mkdir -p $O
export TMPDIR=/tmp
echo "== tune"; timeout 600 python tools/tune_fwd.py $O/tune_fwd.txt 2>&1 | tail -60
echo "== diag predict"; timeout 300 python tools/diag_predict.py 2>&1 | tail -12
echo "== tests"; timeout 1800 python -m pytest tests/test_spconv_gpu.py tests/test_e2e_gpu.py tests/test_pipeline_gpu.py tests/test_dist_gpu.py -q -m gpu --tb=short 2>&1 | tail -60 > $O/t_all.log; tail -8 $O/t_all.log
echo "== bench default"; timeout 900 python bench.py --steps 40 --warmup 8 --no-cpu-baseline --no-extras > $O/bench_default.json 2> $O/bench_default.err; python tools/show_bench.py $O/bench_default.json
echo "== bench no remap"; D3D_XCD_REMAP=0 timeout 900 python bench.py --steps 40 --warmup 8 --no-cpu-baseline --no-extras > $O/bench_noremap.json 2> /dev/null; python tools/show_bench.py $O/bench_noremap.json | head -1
