#!/bin/bash
# back in the container: rocprofv3 output of tools/round6_gpu_{rows_f,profiles}.sh -> profiles/r06_*_{kernel_stats.csv,pmc_summary.json}
set -e
cd "$(dirname "$0")/.."
python tools/summarize_profile.py gpurun_out/prof_r06_g1_2p20 r06_g1_2p20 g1 20 | tail -1
python tools/summarize_profile.py gpurun_out/prof_r06_g1_2p24 r06_g1_2p24 g1 24 | tail -1
python tools/summarize_profile.py gpurun_out/prof_r06_g2_2p20 r06_g2_2p20 g2 20 | tail -1
python tools/summarize_profile.py gpurun_out/prof_r06_pairing_2p16 r06_pairing_2p16 pairing 16 | tail -1
python tools/summarize_profile.py gpurun_out/prof_r06_rows_f_g1_2p20 r06_rows_f_g1_2p20 g1 20 | tail -1
python tools/summarize_profile.py gpurun_out/prof_r06_rows_f_g2_2p18 r06_rows_f_g2_2p18 g2 18 | tail -1
python tools/summarize_profile.py gpurun_out/prof_r06_rows_f_g2_2p20 r06_rows_f_g2_2p20 g2 20 | tail -1
