#!/bin/bash
# same-box A/B of several builds of the library: interleaved runs of bench.py (devices of the pool differ by ~8 %, so numbers
# taken on different boxes are never compared).  usage: tools/ab_multi.sh "<name=path.so> ..." [bench args]
#   the in-tree library is always run as "main"; builds come from  make OBJDIR=.. LIBDIR=../../ab_libs/<name> EXTRA=-D...
VARS=$1; shift
MAIN=ark-blst_amd/lib/libarkblst_amd.so
cp $MAIN /tmp/main.so
for i in 1 2 3; do
  for v in main=/tmp/main.so $VARS; do
    name=${v%%=*}; path=${v#*=}
    cp $path $MAIN
    python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-secondary "$@" 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); p=d['phases_ms']; print('$name', round(d['value']/1e6,1), 'Mpts/s', round(d['ms_per_step'],3), 'ms', d['bit_exact'], 'sort', round(p['digits_ms']+p['scatter_ms'],3), 'acc', round(p['accumulate_ms'],3), 'red', round(p['reduce_ms'],3), 'comb', round(p['combine_ms'],3), 'host', round(p['host_fold_ms'],3))"
  done
done
cp /tmp/main.so $MAIN
