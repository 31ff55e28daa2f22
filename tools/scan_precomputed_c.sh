for c in 13 14 15 16 17 18 20; do
python bench.py --steps 10 --warmup 3 --no-secondary --no-cpu-baseline --precomputed --window-bits $c 2>/dev/null | python -c "
import json,sys
l=json.loads(sys.stdin.read().strip().splitlines()[-1]); print($c, round(l['ms_per_step'],3), l['bit_exact'], {k:round(v,3) for k,v in l['phases_ms'].items()})" || exit 1
done
