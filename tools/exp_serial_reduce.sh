set -e
mkdir -p gpurun_out
for cfg in "default" "MI_TEST_SERIAL_MIN_BUCKETS=1 MI_TEST_SERIAL_L=8" "MI_TEST_SERIAL_MIN_BUCKETS=1 MI_TEST_SERIAL_L=4" "MI_TEST_SERIAL_MIN_BUCKETS=1 MI_TEST_SERIAL_L=16" "default"; do
  echo "== $cfg" >> gpurun_out/r5d_serial_exp.txt
  if [ "$cfg" = default ]; then python tools/sweep_sizes.py g1 16 20 --test-hooks >> gpurun_out/r5d_serial_exp.txt 2>&1; else env $cfg python tools/sweep_sizes.py g1 16 20 --test-hooks >> gpurun_out/r5d_serial_exp.txt 2>&1; fi
done
python tools/sweep_sizes.py g1 10 23 --scan-c > gpurun_out/r5d_scan_c_g1_plain.jsonl 2> gpurun_out/r5d_scan.err
