#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r2j9; mkdir -p $O
export TMPDIR=/tmp
echo "== full gpu suite"; timeout 1500 python -m pytest tests -q -m gpu --tb=short 2>&1 | tail -25
run() { # name, env..., -- args
  name=$1; shift
  echo "== bench $name"
  env "$@" timeout 900 python bench.py --steps 100 --warmup 5 --no-cpu-baseline --no-extras $EXTRA > $O/bench_$name.json 2> $O/bench_$name.err
  python tools/show_bench.py $O/bench_$name.json | head -2; grep -i "warn\|error" $O/bench_$name.err | head -3
}
EXTRA="" run pf1_g1 D3D_GRAPH_TAIL=1
EXTRA="" run pf1_g0 D3D_GRAPH_TAIL=0
EXTRA="--no-prefetch" run pf0_g1 D3D_GRAPH_TAIL=1
EXTRA="--no-prefetch" run pf0_g0 D3D_GRAPH_TAIL=0
echo "== profile default"; timeout 900 bash tools/profile_step.sh $O/kernel_stats.txt; head -34 $O/kernel_stats.txt | cut -c1-150; grep -n "# gaps" -A12 $O/kernel_stats.txt | cut -c1-150
