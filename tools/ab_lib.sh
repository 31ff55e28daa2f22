#!/bin/bash
# same-box A/B of two builds of the library: interleaved runs of bench.py (devices differ by ~10 %, so never compare
# numbers taken on different boxes).  usage: tools/ab_lib.sh <alt.so> [bench args]
ALT=$1; shift
MAIN=ark-blst_amd/lib/libarkblst_amd.so
cp $MAIN /tmp/main.so
for i in 1 2 3; do
  for v in main alt; do
    if [ $v = alt ]; then cp $ALT $MAIN; else cp /tmp/main.so $MAIN; fi
    python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-secondary "$@" 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); p=d['phases_ms']; print('$v', round(d['value']/1e6,1), 'Mpts/s', round(d['ms_per_step'],3), 'ms', d['bit_exact'], 'acc', round(p['accumulate_ms'],3), 'red', round(p['reduce_ms'],3), 'host', round(p['host_fold_ms'],3))"
  done
done
cp /tmp/main.so $MAIN
