#!/bin/bash
# round-6 rocprofv3 evidence for the rows-(f) kernels (normalize, deserialize + validate, check_batch)
set -e
bash tools/profile_rows_f.sh r06_rows_f_g1_2p20 g1 20 > gpurun_out/prof_r06_rows_f_g1_2p20.log 2>&1
bash tools/profile_rows_f.sh r06_rows_f_g2_2p18 g2 18 > gpurun_out/prof_r06_rows_f_g2_2p18.log 2>&1
bash tools/profile_rows_f.sh r06_rows_f_g2_2p20 g2 20 > gpurun_out/prof_r06_rows_f_g2_2p20.log 2>&1
ls gpurun_out | grep prof_r06_rows | head
