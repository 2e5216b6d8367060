#!/bin/bash
# round 3, job 1: full GPU suite (no -x), graph-tail diagnosis, driver-flag bench after the deterministic reductions
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r3j1
mkdir -p $O
timeout 900 python -m pytest tests -q -m gpu --tb=short 2>&1 | tail -60 > $O/tests.log
tail -5 $O/tests.log
timeout 300 python tools/diag_graph.py > $O/diag_graph.txt 2>&1
grep -c "<<<" $O/diag_graph.txt
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err
tail -c 1500 $O/bench.json
