#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r2j12; mkdir -p $O
export TMPDIR=/tmp
echo "== conv tests"; timeout 900 python -m pytest tests/test_spconv_gpu.py -q -m gpu --tb=short -x 2>&1 | tail -5
for v in 0 40 23; do
  echo "== bench fwd variant $v"
  D3D_FWD_VARIANT=$v timeout 900 python bench.py --steps 200 --warmup 5 --no-cpu-baseline --no-extras > $O/bench_v$v.json 2> $O/bench_v$v.err
  python tools/show_bench.py $O/bench_v$v.json | head -2
done
