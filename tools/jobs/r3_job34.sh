#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r3j34; mkdir -p $O
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/round3_bench_driver_flags.json 2> $O/bench.err
python tools/show_bench.py $O/round3_bench_driver_flags.json w | grep -E "^value|host|bf16 |^   c" | cut -c1-220
