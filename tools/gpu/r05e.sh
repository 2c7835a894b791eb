#!/bin/bash
# stress ahead (two launches per cloth substep): parity first, then A/B against MPMHIP_STRESS_AHEAD=0
mkdir -p gpurun_out/r05e
timeout 1500 python -m pytest tests/test_gpu_ref_golden.py tests/test_gpu_parity.py tests/test_gpu_golden.py tests/test_gpu_edges.py tests/test_gpu_api.py tests/test_gpu_sort.py tests/test_gpu_branch_flips.py tests/test_gpu_fd.py -m gpu -x -q --durations=5 > gpurun_out/r05e/parity.txt 2>&1; echo "rc=$?" >> gpurun_out/r05e/parity.txt
tail -6 gpurun_out/r05e/parity.txt
python tools/gpu/ab5.py --libs default,default@MPMHIP_STRESS_AHEAD=0 --scenes sheet-500k,garment-120k-aniso,garment-120k-iso,demo-250 --reps 2 --advance 2000 --out gpurun_out/r05e/ab_stress_ahead.json > gpurun_out/r05e/ab_stress_ahead.txt 2>&1
cat gpurun_out/r05e/ab_stress_ahead.txt
timeout 900 python -m pytest tests/test_gpu_fullsize.py -m gpu -x -q -s -k "gamma0 or s4_sheet_500k_20 or s3_garment or s2_" > gpurun_out/r05e/fullsize.txt 2>&1; echo "rc=$?" >> gpurun_out/r05e/fullsize.txt
grep -v "^Particles\|^Total" gpurun_out/r05e/fullsize.txt | tail -14 | cut -c1-250
timeout 600 python -m pytest tests/test_dist.py -m gpu -x -q -k "four_and_eight" > gpurun_out/r05e/dist48.txt 2>&1; tail -3 gpurun_out/r05e/dist48.txt
