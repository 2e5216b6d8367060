#!/bin/bash
# round-2 GPU job 1: new kernels (tile wgrad, conv-epilogue BN sums, padded stem) -- tests, A/B bench, profile
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r2j1; mkdir -p $O
export TMPDIR=/tmp
echo "== new tests"; timeout 900 python -m pytest tests/test_spconv_gpu.py -q -m gpu -x 2>&1 | tail -15 > $O/t_spconv.log; tail -5 $O/t_spconv.log
echo "== e2e + rest"; timeout 1500 python -m pytest tests -q -m gpu --deselect tests/test_spconv_gpu.py 2>&1 | tail -40 > $O/t_rest.log; tail -12 $O/t_rest.log
echo "== bench default"; timeout 600 python bench.py --steps 30 --warmup 8 > $O/bench_bf16.json 2> $O/bench_bf16.err; tail -c 600 $O/bench_bf16.json
echo "== bench fp32"; timeout 600 python bench.py --steps 20 --warmup 5 --dtype fp32 --no-cpu-baseline > $O/bench_fp32.json 2> $O/bench_fp32.err; head -c 400 $O/bench_fp32.json
echo "== bench legacy wgrad"; D3D_WGRAD_LEGACY=1 timeout 600 python bench.py --steps 30 --warmup 8 --no-cpu-baseline > $O/bench_legacy_wgrad.json 2> /dev/null; head -c 300 $O/bench_legacy_wgrad.json
echo "== bench no epilogue fusion"; D3D_FUSE_BN_STATS=0 timeout 600 python bench.py --steps 30 --warmup 8 --no-cpu-baseline > $O/bench_nofuse.json 2> /dev/null; head -c 300 $O/bench_nofuse.json
echo "== bench wgrad 512 blocks"; D3D_WGRAD_BLOCKS=512 timeout 600 python bench.py --steps 30 --warmup 8 --no-cpu-baseline > $O/bench_wg512.json 2> /dev/null; head -c 300 $O/bench_wg512.json
echo "== bench bn2d fused"; D3D_FUSE_BN2D=1 timeout 600 python bench.py --steps 30 --warmup 8 --no-cpu-baseline > $O/bench_bn2d.json 2> /dev/null; head -c 300 $O/bench_bn2d.json
echo "== profile"; timeout 900 bash tools/profile_step.sh $O/kernel_stats.txt; head -60 $O/kernel_stats.txt
