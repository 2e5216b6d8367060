#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_graph_tail_gpu.py tests/test_dist_gpu.py -q -m gpu --tb=short -x 2>&1 | tail -4
for a in "" "--no-prefetch-thread" "" "--no-prefetch-thread"; do
timeout 300 python bench.py --gpus 1 --steps 60 --warmup 5 --no-cpu-baseline --no-extras $a 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('fp32 $a', d['value'], d['ms_per_step'], d['config']['host_ms_per_step'], d.get('final_loss'))"
done
for a in "" "--no-prefetch-thread"; do
timeout 300 python bench.py --gpus 1 --steps 60 --warmup 5 --no-cpu-baseline --no-extras --dtype bf16 $a 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bf16 $a', d['value'], d['ms_per_step'], d['config']['host_ms_per_step'], d.get('final_loss'))"
done
