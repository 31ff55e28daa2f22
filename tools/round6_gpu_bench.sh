#!/bin/bash
# round-6 final GPU session, second half (after the profiles of the same sources are committed): full suite, smoke, the bench lines
set -e
mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q > gpurun_out/r6z_tests.log 2>&1 || { tail -30 gpurun_out/r6z_tests.log; exit 1; }
tail -3 gpurun_out/r6z_tests.log
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r6z_smoke.log 2>&1 || { tail -20 gpurun_out/r6z_smoke.log; exit 1; }
tail -1 gpurun_out/r6z_smoke.log
python bench.py --steps 20 --warmup 5 --secondary-out gpurun_out/r6z_bench_secondary.json > gpurun_out/r6z_bench.log 2> gpurun_out/r6z_bench.err
wc -c gpurun_out/r6z_bench.log
python bench.py --steps 5 --warmup 2 --log-n 24 --no-secondary > gpurun_out/r6z_bench_2p24.log 2>> gpurun_out/r6z_bench.err
python bench.py > gpurun_out/r6z_bench_default.log 2>> gpurun_out/r6z_bench.err
wc -c gpurun_out/r6z_bench_default.log
