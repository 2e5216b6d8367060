#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r2j13; mkdir -p $O
export TMPDIR=/tmp
timeout 600 python -m cProfile -o /tmp/prof.out bench.py --steps 150 --warmup 5 --no-cpu-baseline --no-extras --no-launch-log --burn-in 0.3 > $O/bench.json 2> $O/bench.err
python tools/show_bench.py $O/bench.json | head -2
python - <<'PY' > gpurun_out/r2j13/host_profile.txt
import pstats
p = pstats.Stats('/tmp/prof.out')
p.sort_stats('tottime').print_stats(45)
p.sort_stats('cumulative').print_stats(70)
PY
head -120 $O/host_profile.txt | cut -c1-170
