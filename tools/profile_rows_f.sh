#!/bin/bash
# rocprofv3 evidence for the rows-(f) kernels (k_norm_*, k_deserialize_g1/g2, k_validate): kernel-trace stats, then FETCH_SIZE / WRITE_SIZE / SQ
# counters in their own passes (MI355X_MICROARCH.md: never combined with the hip/hsa trace domains).  VERDICT r05 missing #4.
#   tools/profile_rows_f.sh <tag> <g1|g2> <log_n>      then: python tools/summarize_profile.py gpurun_out/prof_<tag> <tag> <group> <log_n>
set -e
TAG=$1; G=$2; LN=$3
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
ARGS="tools/bench_rows_f.py $G $LN --oneline --once --only=normalize_batch,deserialize_batch_validate,check_batch"
echo "$ARGS" > $OUT/command.txt
python3 -c "import bench; print(bench.source_hash())" > $OUT/source.sha256
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $ARGS > $OUT/bench_line_under_rocprof.json 2> $OUT/stats.log
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/fetch -- python3 $ARGS > $OUT/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/write -- python3 $ARGS > $OUT/write.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/sq -- python3 $ARGS > $OUT/sq.log 2>&1
find $OUT -name "*.csv" | wc -l
