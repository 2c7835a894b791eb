#!/bin/bash
# how much room does the one fixed margin of the full-size velocity distribution leave?  The two ensemble tests, three times.
mkdir -p gpurun_out/r05m
for i in 1 2 3; do
  timeout 1200 python -m pytest tests/test_gpu_fullsize.py -m gpu -q -s -k "s4_sheet_500k_1000 or s3_one_frame" > gpurun_out/r05m/run$i.txt 2>&1; echo "rc=$?" >> gpurun_out/r05m/run$i.txt
  grep "^S[34] substep\|rc=\|passed\|failed" gpurun_out/r05m/run$i.txt | cut -c1-400
done
