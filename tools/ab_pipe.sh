#!/bin/bash
# same-box A/B of library builds under tools/pipe_scan.py: tools/ab_pipe.sh "<alt1.so> <alt2.so> .." <pipe_scan args...>
ALTS=$1; shift
MAIN=ark-blst_amd/lib/libarkblst_amd.so
cp $MAIN /tmp/main.so
for v in main $ALTS; do
  if [ $v = main ]; then cp /tmp/main.so $MAIN; else cp $v $MAIN; fi
  echo "== $v"
  python tools/pipe_scan.py "$@" 2>/dev/null
done
cp /tmp/main.so $MAIN
