#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r2j17; mkdir -p $O
export TMPDIR=/tmp
echo "== find on, graph on (capture in immediate mode)"; DIAG_STEPS=60 timeout 300 python tools/diag_nan.py 2>&1 | grep "^step\|non-finite" | head -8
for g in 1 0; do
  echo "== bench graph=$g"
  D3D_GRAPH_TAIL=$g timeout 900 python bench.py --steps 200 --warmup 5 --no-cpu-baseline --no-extras > $O/bench_g$g.json 2> $O/bench_g$g.err
  python tools/show_bench.py $O/bench_g$g.json | head -2; python -c "
import json,sys; d=json.loads(open('$O/bench_g$g.json').read().strip().splitlines()[-1]); print('final loss', d['config']['final_loss'], 'gnorm', d['config'].get('final_grad_norm'))"; tail -2 $O/bench_g$g.err | cut -c1-200
done
echo "== graph tests"; timeout 900 python -m pytest tests/test_graph_tail_gpu.py -q -m gpu --tb=short 2>&1 | tail -6
