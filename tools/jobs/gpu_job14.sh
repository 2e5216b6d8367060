#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r2j14; mkdir -p $O
export TMPDIR=/tmp
echo "== tests"; timeout 1500 python -m pytest tests/test_graph_tail_gpu.py tests/test_e2e_gpu.py tests/test_spconv_gpu.py -q -m gpu --tb=short > $O/tests.log 2>&1; tail -4 $O/tests.log; grep -n "^E  \|^FAILED\|Error" $O/tests.log | head -30
for st in 1 0; do
  echo "== bench stack=$st"
  D3D_FUSE_STACK=$st timeout 900 python bench.py --steps 200 --warmup 5 --no-cpu-baseline --no-extras > $O/bench_s$st.json 2> $O/bench_s$st.err
  python tools/show_bench.py $O/bench_s$st.json | head -2; grep -i "error\|Traceback" -A5 $O/bench_s$st.err | head -20
done
