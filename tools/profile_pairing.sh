#!/bin/bash
# rocprofv3 evidence for the pairing row (BASELINE config #5): kernel stats and PMC passes of tools/bench_pairing.py 16.
# Run on the GPU box via gpurun:  tools/profile_pairing.sh <tag>;  then  python tools/summarize_profile.py gpurun_out/prof_<tag> <tag> pairing 16
set -e
TAG=${1:-r03_pairing_2p16}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
ARGS="tools/bench_pairing.py 16 2"
echo "$ARGS" > $OUT/command.txt
python3 -c "import bench; print(bench.source_hash())" > $OUT/source.sha256   # the tree the counters were measured on
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $ARGS > $OUT/bench_line_under_rocprof.json 2> $OUT/stats.log
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/fetch -- python3 $ARGS > $OUT/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/write -- python3 $ARGS > $OUT/write.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/sq -- python3 $ARGS > $OUT/sq.log 2>&1
# what a lone wave of k_miller_accumulate waits for (own passes: a counter the part does not know must not lose the others)
rocprofv3 --pmc SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT --kernel-trace --output-format csv -d $OUT/stall -- python3 $ARGS > $OUT/stall.log 2>&1 || true
rocprofv3 --pmc SQ_INST_CYCLES_VMEM SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_WAVES SQ_INSTS_VALU --kernel-trace --output-format csv -d $OUT/stall2 -- python3 $ARGS > $OUT/stall2.log 2>&1 || true
find $OUT -name "*.csv" | wc -l
