#!/bin/bash
# One GPU-box call that reproduces every number quoted in DESIGN.md section 8 (takes ~4 minutes):
#   /usr/local/graft/bin/gpurun --timeout 1100 -- 'bash tools/repro_all.sh'
set -e
mkdir -p gpurun_out/repro
python -m pytest tests -m gpu -x -q 2>&1 | tail -1
python -c "import __graft_entry__ as g; g.smoke()"
python bench.py > gpurun_out/repro/g1_2p20.json
python bench.py --group g2 --no-secondary > gpurun_out/repro/g2_2p20.json
python bench.py --log-n 16 --no-secondary > gpurun_out/repro/g1_2p16.json
python bench.py --log-n 24 --steps 3 --warmup 1 --no-secondary > gpurun_out/repro/g1_2p24.json
python tools/bench_pairing.py 16 5 > gpurun_out/repro/pairing_2p16.json
python tools/bench_normalize.py > gpurun_out/repro/normalize.txt 2>&1 || true
python tools/bench_deserialize.py 20 > gpurun_out/repro/deserialize.txt 2>&1 || true
python - <<'PY'
import json
for f in ("g1_2p16", "g1_2p20", "g1_2p24", "g2_2p20"):
    d = json.loads(open(f"gpurun_out/repro/{f}.json").read().strip().splitlines()[-1])
    print(f, "%.3g points/s" % d["value"], "%.2f ms" % d["ms_per_step"], "bit_exact", d["bit_exact"], "c", d["config"]["window_bits"],
          "cpu_baseline %.3g" % d["cpu_baseline"]["value"] if "cpu_baseline" in d else "")
d = json.loads(open("gpurun_out/repro/pairing_2p16.json").read().strip().splitlines()[-1])
print("pairing_2p16", "%.3g pairs/s" % d["value"], "%.2f ms" % d["ms"], "exact", d["bit_exact_1024_pairs_vs_c_oracle"], "cpu_baseline %.3g" % d["cpu_baseline"]["value"])
PY
tail -2 gpurun_out/repro/normalize.txt; tail -3 gpurun_out/repro/deserialize.txt
