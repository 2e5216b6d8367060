#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r2j8; mkdir -p $O
export TMPDIR=/tmp
echo "== tests"; timeout 1200 python -m pytest tests/test_graph_tail_gpu.py tests/test_nms_gpu.py tests/test_spconv_gpu.py tests/test_dist_gpu.py -q -m gpu --tb=short -x -k "graph or nms or bn or syncbn or predict" 2>&1 | tail -25
echo "== bn microbench"; timeout 600 python tools/bench_bn.py $O/bench_bn.json 2>&1 | tail -80
for g in 1 0; do for b in 0 1; do
  echo "== bench graph=$g bn2d=$b"
  D3D_GRAPH_TAIL=$g D3D_FUSE_BN2D=$b timeout 900 python bench.py --steps 100 --warmup 5 --no-cpu-baseline --no-extras > $O/bench_g${g}_b${b}.json 2> $O/bench_g${g}_b${b}.err
  python tools/show_bench.py $O/bench_g${g}_b${b}.json | head -1; grep -i "warn\|error\|burn" $O/bench_g${g}_b${b}.err | head -5
done; done
echo "== profile graph=1 bn2d=1"; D3D_FUSE_BN2D=1 timeout 900 bash tools/profile_step.sh $O/kernel_stats_g1_b1.txt; head -30 $O/kernel_stats_g1_b1.txt | cut -c1-150
