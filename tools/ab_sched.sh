#!/bin/bash
# same-box A/B of library builds on the size sweep's phase columns (product library swapped in place)
LIB=ark-blst_amd/lib/libarkblst_amd.so
cp $LIB /tmp/keep.so
for v in "$@"; do
  cp $v $LIB
  echo "== $v"
  python tools/sweep_sizes.py g1 16 24 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    r=json.loads(l); print(r['log_n'],'c',r['c'],'ms',r['ms'],'sort',r['sort'],'sched',r['sched'],'acc',r['acc'],'red',r['reduce'],r['ok'])"
done
cp /tmp/keep.so $LIB
