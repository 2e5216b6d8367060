#!/bin/bash
# round 3 evidence: PMC passes of the roofline kernel, kernel traces of both precisions, the bench as the driver runs it
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r3ev
mkdir -p $O
bash tools/pmc_spconv.sh $O > $O/pmc.log 2>&1
tail -3 $O/pmc.log
bash tools/profile_step.sh $O/round3_kernel_stats_fp32.txt --dtype fp32 > $O/prof_fp32.log 2>&1
bash tools/profile_step.sh $O/round3_kernel_stats_bf16.txt --dtype bf16 > $O/prof_bf16.log 2>&1
head -12 $O/round3_kernel_stats_fp32.txt | cut -c1-140
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/round3_bench_driver_flags.json 2> $O/bench.err
python tools/show_bench.py $O/round3_bench_driver_flags.json w | cut -c1-400
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
