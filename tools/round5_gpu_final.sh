#!/bin/bash
# round-5 final GPU session: full suite, smoke, default bench (line + sidecar), rocprof evidence stamped with the final source hash
mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q > gpurun_out/r5z_tests.log 2>&1; tail -3 gpurun_out/r5z_tests.log
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r5z_smoke.log 2>&1; tail -1 gpurun_out/r5z_smoke.log
python bench.py --steps 20 --warmup 5 --secondary-out gpurun_out/r5z_bench_secondary.json > gpurun_out/r5z_bench.log 2> gpurun_out/r5z_bench.err; wc -c gpurun_out/r5z_bench.log
python bench.py --steps 5 --warmup 2 --log-n 24 --no-secondary > gpurun_out/r5z_bench_2p24.log 2>> gpurun_out/r5z_bench.err
bash tools/round5_gpu_profiles.sh > gpurun_out/r5z_profiles.log 2>&1; tail -2 gpurun_out/r5z_profiles.log
