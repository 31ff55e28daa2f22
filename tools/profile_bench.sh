#!/bin/bash
# rocprofv3 evidence for one bench.py workload: kernel-trace stats, then PMC counters in their own passes
# (MI355X_MICROARCH.md: FETCH_SIZE and WRITE_SIZE do not fit one pass; never combined with the hip/hsa trace domains).
# Run on the GPU box via gpurun:   tools/profile_bench.sh <tag> [bench.py arguments...]
# then, back in the container:     python tools/summarize_profile.py gpurun_out/prof_<tag> <tag> <group> <log_n> [precomputed]
set -e
TAG=${1:-r03_g1_2p20}; shift || true
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
ARGS="bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-secondary $*"
echo "$ARGS" > $OUT/command.txt
python3 -c "import bench; print(bench.source_hash())" > $OUT/source.sha256   # the tree the counters were measured on
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $ARGS > $OUT/bench_line_under_rocprof.json 2> $OUT/stats.log
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/fetch -- python3 $ARGS > $OUT/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/write -- python3 $ARGS > $OUT/write.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/sq -- python3 $ARGS > $OUT/sq.log 2>&1
find $OUT -name "*.csv" | wc -l
