#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r2j7; mkdir -p $O
export TMPDIR=/tmp
echo "== bn tests"; timeout 900 python -m pytest tests/test_spconv_gpu.py -q -m gpu --tb=short -k "bn" 2>&1 | tail -8
echo "== bench bn2d off"; timeout 900 python bench.py --steps 100 --warmup 5 --no-cpu-baseline --no-extras > $O/bench_bn2d0.json 2> /dev/null; python tools/show_bench.py $O/bench_bn2d0.json | head -1
echo "== bench bn2d on"; D3D_FUSE_BN2D=1 timeout 900 python bench.py --steps 100 --warmup 5 --no-cpu-baseline --no-extras > $O/bench_bn2d1.json 2> /dev/null; python tools/show_bench.py $O/bench_bn2d1.json | head -1
echo "== bench bn2d on, 128 big blocks"; D3D_FUSE_BN2D=1 D3D_BN_BIG_BLOCKS=128 timeout 900 python bench.py --steps 100 --warmup 5 --no-cpu-baseline --no-extras > $O/bench_bn2d1b.json 2> /dev/null; python tools/show_bench.py $O/bench_bn2d1b.json | head -1
echo "== bench bn2d on, 512 big blocks"; D3D_FUSE_BN2D=1 D3D_BN_BIG_BLOCKS=512 timeout 900 python bench.py --steps 100 --warmup 5 --no-cpu-baseline --no-extras > $O/bench_bn2d1c.json 2> /dev/null; python tools/show_bench.py $O/bench_bn2d1c.json | head -1
echo "== profile bn2d on"; D3D_FUSE_BN2D=1 timeout 900 bash tools/profile_step.sh $O/kernel_stats_bn2d.txt; head -24 $O/kernel_stats_bn2d.txt | cut -c1-150
echo "== bench driver flags again (burn-in)"; timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-extras > $O/bench_burn.json 2> $O/bench_burn.err; python tools/show_bench.py $O/bench_burn.json | head -1; grep burn $O/bench_burn.err
