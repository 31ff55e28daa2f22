#!/bin/bash
# rocprofv3 evidence for tools/bench_pairing.py (2^16 pairs): kernel stats, then HBM traffic counters in their own passes.
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/profp
mkdir -p $OUT
ARGS="tools/bench_pairing.py 16 2"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $ARGS > $OUT.stats.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/fetch -- python3 $ARGS > $OUT.fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/write -- python3 $ARGS > $OUT.write.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/sq -- python3 $ARGS > $OUT.sq.log 2>&1
find $OUT -name "*.csv" | head -20
tail -1 $OUT.stats.log
