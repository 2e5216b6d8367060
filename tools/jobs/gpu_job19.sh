#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r2j19; mkdir -p $O
export TMPDIR=/tmp
echo "== prefetch test"; timeout 600 python -m pytest tests/test_graph_tail_gpu.py -q -m gpu --tb=short -k prefetched 2>&1 | tail -2
for i in 1 2 3; do for f in 1 0; do
  D3D_MIOPEN_FIND=$f timeout 600 python bench.py --steps 100 --warmup 5 --no-cpu-baseline --no-extras --no-launch-log > $O/bench_f${f}_$i.json 2> /dev/null
  echo "find=$f run $i: $(python tools/show_bench.py $O/bench_f${f}_$i.json | head -2 | tr '\n' ' ' | cut -c1-150)"
done; done
