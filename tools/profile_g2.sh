#!/bin/bash
# rocprofv3 kernel stats for the G2 MSM bench line (2^20 points)
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/profg2
mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 bench.py --group g2 --steps 5 --warmup 2 --no-cpu-baseline --no-secondary > $OUT.stats.log 2>&1
find $OUT -name "*kernel_stats.csv" | head -2
