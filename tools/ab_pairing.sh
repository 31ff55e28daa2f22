#!/bin/bash
# same-box A/B of two library builds on tools/bench_pairing.py: usage tools/ab_pairing.sh <main.so> <alt.so> [logn]
MAIN=$1; ALT=$2; LOGN=${3:-16}
LIB=ark-blst_amd/lib/libarkblst_amd.so
for i in 1 2; do
  for v in main alt; do
    if [ $v = alt ]; then cp $ALT $LIB; else cp $MAIN $LIB; fi
    python tools/bench_pairing.py $LOGN 3 | python -c "import sys,json; d=json.loads(sys.stdin.read()); p=d['phases_ms']; print('$v', round(d['value']/1e6,2), 'Mpairs/s', round(d['ms'],2), 'ms miller', round(p['miller_loops'],2), 'lines', round(p['k_miller_lines2'],3), 'acc', round(p['k_miller_accumulate'],3), 'tree', round(p['fp12_tree'],2), 'host', round(p['host_tail_and_final_exp'],2))"
  done
done
cp $MAIN $LIB
