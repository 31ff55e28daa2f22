#!/bin/bash
# same-box A/B over several sizes: usage tools/ab_sizes.sh <alt.so> "<logn list>" [bench args]
ALT=$1; SIZES=$2; shift; shift
MAIN=ark-blst_amd/lib/libarkblst_amd.so
cp $MAIN /tmp/main.so
for ln in $SIZES; do
  for v in main alt main alt; do
    if [ $v = alt ]; then cp $ALT $MAIN; else cp /tmp/main.so $MAIN; fi
    python bench.py --log-n $ln --steps 10 --warmup 3 --no-cpu-baseline --no-secondary "$@" 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); p=d['phases_ms']; print('$v', 2**$ln, round(d['value']/1e6,1), 'Mpts/s', round(d['ms_per_step'],3), 'ms', d['bit_exact'], 'c', d['config']['window_bits'], 'acc', round(p['accumulate_ms'],3), 'red', round(p['reduce_ms'],3), 'host', round(p['host_fold_ms'],3))"
  done
done
cp /tmp/main.so $MAIN
