#!/bin/bash
# round-5 GPU session A: full GPU suite, default bench line + sidecar, size sweeps (plain and validated plans)
mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q > gpurun_out/r5i_tests.log 2>&1; tail -3 gpurun_out/r5i_tests.log
python bench.py --steps 20 --warmup 5 --secondary-out gpurun_out/r5i_bench_secondary.json > gpurun_out/r5i_bench.log 2> gpurun_out/r5i_bench.err; tail -c 300 gpurun_out/r5i_bench.err; wc -c gpurun_out/r5i_bench.log
python tools/sweep_sizes.py g1 10 25 --validated > gpurun_out/r05_sweep_g1_2p10_2p25_validated.jsonl 2>/dev/null
python tools/sweep_sizes.py g1 10 25 > gpurun_out/r05_sweep_g1_2p10_2p25_plain.jsonl 2>/dev/null
python tools/sweep_sizes.py g2 10 22 --validated > gpurun_out/r05_sweep_g2_2p10_2p22_validated.jsonl 2>/dev/null
python tools/sweep_sizes.py g2 10 22 > gpurun_out/r05_sweep_g2_2p10_2p22_plain.jsonl 2>/dev/null
wc -l gpurun_out/r05_sweep_*.jsonl
