#!/bin/bash
# round-2 GPU job 3: truly pipelined fwd / wgrad kernels (counted vmcnt, LDS-only barriers, buffer-load masking)
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r2j3; mkdir -p $O
export TMPDIR=/tmp
echo "== spconv tests"; timeout 900 python -m pytest tests/test_spconv_gpu.py -q -m gpu --tb=short -x 2>&1 | tail -40 > $O/t_spconv.log; tail -4 $O/t_spconv.log
echo "== e2e"; timeout 1500 python -m pytest tests/test_e2e_gpu.py -q -m gpu --tb=short -s 2>&1 | tail -100 > $O/t_e2e.log; tail -8 $O/t_e2e.log
echo "== dist"; timeout 900 python -m pytest tests/test_dist_gpu.py -q -m gpu --tb=short 2>&1 | tail -60 > $O/t_dist.log; tail -8 $O/t_dist.log
echo "== bench default"; timeout 900 python bench.py --steps 40 --warmup 8 --no-cpu-baseline > $O/bench_default.json 2> $O/bench_default.err; python tools/show_bench.py $O/bench_default.json
for v in 1 2 3 4; do echo "== fwd variant $v"; D3D_FWD_VARIANT=$v timeout 600 python bench.py --steps 40 --warmup 8 --no-cpu-baseline --no-extras > $O/bench_fv$v.json 2> /dev/null; python tools/show_bench.py $O/bench_fv$v.json; done
echo "== wgrad 512 blocks"; D3D_WGRAD_BLOCKS=512 timeout 600 python bench.py --steps 40 --warmup 8 --no-cpu-baseline --no-extras > $O/bench_wg512.json 2> /dev/null; python tools/show_bench.py $O/bench_wg512.json
echo "== profile"; timeout 900 bash tools/profile_step.sh $O/kernel_stats.txt; head -40 $O/kernel_stats.txt
echo "== workloads all"; timeout 1500 python bench.py --steps 40 --warmup 8 --workload all --no-cpu-baseline > $O/bench_all.json 2> $O/bench_all.err; python tools/show_bench.py $O/bench_all.json workloads; tail -3 $O/bench_all.err
