#!/bin/bash
# does the window-group pipeline become reliable when the process gives the runtime more hardware queues?  GPU_MAX_HW_QUEUES is read by the
# HIP runtime when it initialises, so it is set for the whole process here.  Same box, same library.
for q in 4 8; do
  for acc2 in 0 1; do
    echo "== GPU_MAX_HW_QUEUES=$q ACC2=$acc2 pipe_scan"
    GPU_MAX_HW_QUEUES=$q ARKBLST_AMD_PIPELINE_ACC2=$acc2 python tools/pipe_scan.py g1 20,24 "off;1,3;1,2" 2>/dev/null || exit 1
    for p in "" "1,3"; do
      echo "== GPU_MAX_HW_QUEUES=$q ACC2=$acc2 bench.py pipeline='$p'"
      GPU_MAX_HW_QUEUES=$q ARKBLST_AMD_PIPELINE_ACC2=$acc2 ARKBLST_AMD_PIPELINE=$p python bench.py --steps 20 --warmup 5 --no-secondary --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
l=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(l['ms_per_step'],3), l['bit_exact'])" || exit 1
    done
  done
done
