#!/bin/bash
mkdir -p gpurun_out/r05j
timeout 900 python -m pytest tests/test_gpu_stress_ahead.py tests/test_gpu_parity.py -m gpu -x -q > gpurun_out/r05j/sa.txt 2>&1; echo "rc=$?" >> gpurun_out/r05j/sa.txt; tail -6 gpurun_out/r05j/sa.txt
python tools/gpu/ab5.py --libs "default@MPMHIP_STRESS_AHEAD_MAX=100000,default@MPMHIP_STRESS_AHEAD=0" --scenes sheet-500k,garment-120k-aniso --reps 2 --advance 2000 --out gpurun_out/r05j/ab.json > gpurun_out/r05j/ab.txt 2>&1; cut -c1-300 gpurun_out/r05j/ab.txt
