#!/bin/bash
# kernel trace of a few pipelined calls: tools/trace_pipe.sh <tag> <group> <log_n> <pipeline conf> [pipe_scan flags]
set -e
TAG=$1; G=$2; LN=$3; CONF=$4; shift 4 || true
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/trace_$TAG
mkdir -p $OUT
rocprofv3 --kernel-trace --output-format csv -d $OUT/kt -- python3 tools/pipe_scan.py $G $LN "$CONF" "$@" > $OUT/scan.jsonl 2> $OUT/err.log
F=$(find $OUT/kt -name "*kernel_trace.csv" | head -1)
python3 tools/trace_overlap.py $F > $OUT/timeline.txt
rm -rf $OUT/kt
cat $OUT/scan.jsonl; cat $OUT/timeline.txt
