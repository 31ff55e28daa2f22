run() { python bench.py --group g2 --steps 8 --warmup 2 --no-cpu-baseline --no-secondary "$@" 2>&1 | tail -1 | python -c "import sys,json,os; d=json.loads(sys.stdin.read()); p=d['phases_ms']; print(os.environ.get('TAG'), round(d['value']/1e6,1), 'Mpts/s', round(d['ms_per_step'],3), 'ms', d['bit_exact'], 'acc', round(p['accumulate_ms'],3), 'red', round(p['reduce_ms'],3), 'host', round(p['host_fold_ms'],3))"; }
for rep in 1 2; do
TAG=coop run
TAG=one_lane MI_G2_REDUCE_ONE_LANE=1 run
done
TAG=coop18 run --log-n 18
TAG=one18 MI_G2_REDUCE_ONE_LANE=1 run --log-n 18
