#!/bin/bash
# round-6 rocprofv3 evidence: kernel stats + PMC passes for the headline (G1 2^20), the north-star size (G1 2^24), G2 2^20 and the pairing
set -e
bash tools/profile_bench.sh r06_g1_2p20 > gpurun_out/prof_r06_g1_2p20.log 2>&1
bash tools/profile_bench.sh r06_g1_2p24 --log-n 24 > gpurun_out/prof_r06_g1_2p24.log 2>&1
bash tools/profile_bench.sh r06_g2_2p20 --group g2 > gpurun_out/prof_r06_g2_2p20.log 2>&1
bash tools/profile_pairing.sh r06_pairing_2p16 > gpurun_out/prof_r06_pairing.log 2>&1
ls gpurun_out | grep prof_r06 | head
