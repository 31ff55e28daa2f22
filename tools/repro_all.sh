#!/bin/bash
# One GPU-box call that reproduces every number quoted in DESIGN.md section 5 (takes ~4 minutes):
#   /usr/local/graft/bin/gpurun --timeout 1100 -- 'bash tools/repro_all.sh'
# The default bench run covers every single-GPU BASELINE config: the headline in the JSON line, the other legs (g1_2p24, g2_2p20, pairing_2p16,
# normalize / deserialize for both groups, call shapes, in-process multi-device) in the sidecar file; the size sweeps follow.
set -e
mkdir -p gpurun_out/repro
python -m pytest tests -m gpu -x -q 2>&1 | tail -1
python -c "import __graft_entry__ as g; g.smoke()"
python bench.py --secondary-out gpurun_out/repro/bench_secondary.json > gpurun_out/repro/bench_default.json
python tools/sweep_sizes.py g1 14 24 --validated > gpurun_out/repro/sweep_g1.jsonl 2>/dev/null
python tools/sweep_sizes.py g2 14 21 --validated > gpurun_out/repro/sweep_g2.jsonl 2>/dev/null
python tools/bench_normalize.py > gpurun_out/repro/normalize.txt 2>&1 || true
python tools/bench_deserialize.py 20 > gpurun_out/repro/deserialize.txt 2>&1 || true
python - <<'PY'
import json
d = json.loads([l for l in open("gpurun_out/repro/bench_default.json") if l.startswith("{")][-1])
print("g1_2p20", "%.3g points/s" % d["value"], "%.2f ms" % d["ms_per_step"], "bit_exact", d["bit_exact"], "valu frac %.2f" % d["roofline"]["frac"],
      "cpu_baseline %.3g" % d["cpu_baseline"]["value"])
for k, v in json.load(open("gpurun_out/repro/bench_secondary.json"))["secondary"].items():
    if "error" in v:
        print(k, v["error"]); continue
    if "ms_per_step" in v and "value" in v:
        print(k, "%.3g %s" % (v["value"], v.get("unit", "")), "%.2f ms" % v["ms_per_step"], "bit_exact", v.get("bit_exact", v.get("same_result")))
    elif k == "pairing_2p16":
        print(k, "%.3g pairs/s" % v["value"], "%.2f ms" % v["ms"], "cancels", v["product_cancels_to_one"], "exact", v["bit_exact"],
              "cpu_baseline %.3g" % v["cpu_baseline"]["value"])
    else:
        print(k, {a: b for a, b in v.items() if not isinstance(b, (dict, str))})
for f in ("sweep_g1", "sweep_g2"):
    for l in open(f"gpurun_out/repro/{f}.jsonl"):
        r = json.loads(l)
        print(r["group"], "2^%d" % r["log_n"], "c", r["c"], "%.3f ms" % r["ms"], "%.3g points/s" % r["points_per_s"], "ok", r["ok"])
PY
tail -2 gpurun_out/repro/normalize.txt; tail -3 gpurun_out/repro/deserialize.txt
