#!/bin/bash
# round-2 GPU job 4: fwd tiling sweep with the empty-stage skip, fused predict path, test fixes
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r2j4; mkdir -p $O
export TMPDIR=/tmp
echo "== tune"; timeout 600 python tools/tune_fwd.py $O/tune_fwd.txt 2>&1 | tail -45
echo "== nms/decode tests"; timeout 900 python -m pytest tests/test_nms_gpu.py -q -m gpu --tb=short 2>&1 | tail -40 > $O/t_nms.log; tail -6 $O/t_nms.log
echo "== spconv tests"; timeout 900 python -m pytest tests/test_spconv_gpu.py -q -m gpu --tb=short -x 2>&1 | tail -30 > $O/t_spconv.log; tail -3 $O/t_spconv.log
echo "== e2e + dist"; timeout 1500 python -m pytest tests/test_e2e_gpu.py tests/test_dist_gpu.py -q -m gpu --tb=short 2>&1 | tail -60 > $O/t_e2e.log; tail -8 $O/t_e2e.log
echo "== bench default"; timeout 900 python bench.py --steps 40 --warmup 8 --no-cpu-baseline > $O/bench_default.json 2> $O/bench_default.err; python tools/show_bench.py $O/bench_default.json; tail -3 $O/bench_default.err
