#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r3j23
mkdir -p $O
timeout 300 python bench.py --gpus 1 --steps 40 --warmup 5 --no-cpu-baseline --no-extras --augment > $O/aug.json 2> $O/aug.err; python tools/show_bench.py $O/aug.json | head -3
timeout 300 python bench.py --gpus 1 --steps 40 --warmup 5 --no-cpu-baseline --no-extras > $O/noaug.json 2> $O/noaug.err; python tools/show_bench.py $O/noaug.json | head -3
bash tools/profile_step.sh $O/kernel_stats_c4.txt --workload c4 --burn-in 0 > $O/prof_c4.log 2>&1
head -45 $O/kernel_stats_c4.txt | cut -c1-150
bash tools/profile_step.sh $O/kernel_stats_c3.txt --workload c3 --burn-in 0 > $O/prof_c3.log 2>&1
head -30 $O/kernel_stats_c3.txt | cut -c1-150
