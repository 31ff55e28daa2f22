#!/bin/bash
# same-box comparison of several library builds on tools/bench_pairing.py: usage tools/ab_pairing3.sh <logn> <lib1.so> <lib2.so> ...
LOGN=$1; shift
LIB=ark-blst_amd/lib/libarkblst_amd.so
cp $LIB /tmp/keep.so
for i in 1 2; do
  for v in "$@"; do
    cp $v $LIB
    python tools/bench_pairing.py $LOGN 3 | python -c "import sys,json; d=json.loads(sys.stdin.read()); p=d['phases_ms']; print('$v', round(d['value']/1e6,2), 'Mpairs/s', round(d['ms'],2), 'ms miller', round(p['miller_loops'],2), 'lines', round(p['k_miller_lines2'],3), 'acc', round(p['k_miller_accumulate'],3), 'tree', round(p['fp12_tree'],2), 'host', round(p['host_tail_and_final_exp'],2))"
  done
done
cp /tmp/keep.so $LIB
