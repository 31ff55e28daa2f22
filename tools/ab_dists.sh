#!/bin/bash
# usage: ab_dists.sh name=path ...   (interleaved, skewed dists at 2^20)
MAIN=ark-blst_amd/lib/libarkblst_amd.so
cp $MAIN /tmp/main.so
for i in 1 2; do
  for v in main=/tmp/main.so "$@"; do
    name=${v%%=*}; path=${v#*=}
    cp $path $MAIN
    for d in zero_one all_ones all_equal r1cs_mix; do
      python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-secondary --dist $d 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$name $d', round(d['ms_per_step'],3), d['bit_exact'], 'acc', round(d['phases_ms']['accumulate_ms'],3))"
    done
  done
done
cp /tmp/main.so $MAIN
