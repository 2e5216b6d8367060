#!/bin/bash
# round-2 GPU job 2: wgrad v2 (x32 MFMA, f32 variant), epilogue replica sweep, new bench.py
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r2j2; mkdir -p $O
export TMPDIR=/tmp
echo "== spconv tests"; timeout 900 python -m pytest tests/test_spconv_gpu.py -q -m gpu --tb=short 2>&1 | tail -60 > $O/t_spconv.log; tail -4 $O/t_spconv.log
echo "== e2e"; timeout 1500 python -m pytest tests/test_e2e_gpu.py -q -m gpu --tb=short -s 2>&1 | tail -150 > $O/t_e2e.log; tail -15 $O/t_e2e.log
echo "== bench default (new bench.py)"; timeout 900 python bench.py --steps 40 --warmup 8 > $O/bench_default.json 2> $O/bench_default.err; tail -c 1500 $O/bench_default.json; tail -5 $O/bench_default.err
for c in 16 32 64; do echo "== epi copies $c"; D3D_EPI_COPIES=$c timeout 600 python bench.py --steps 40 --warmup 8 --no-cpu-baseline --no-extras > $O/bench_copies$c.json 2> /dev/null; head -c 330 $O/bench_copies$c.json; echo; done
echo "== profile"; timeout 900 bash tools/profile_step.sh $O/kernel_stats.txt --no-extras; head -45 $O/kernel_stats.txt
echo "== workloads all"; timeout 1500 python bench.py --steps 40 --warmup 8 --workload all --no-cpu-baseline > $O/bench_all.json 2> $O/bench_all.err; tail -c 2500 $O/bench_all.json; tail -8 $O/bench_all.err
