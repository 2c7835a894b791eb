#!/bin/bash
mkdir -p gpurun_out/r05i
timeout 1200 python -m pytest tests/test_gpu_stress_ahead.py tests/test_c_abi_demo.py tests/test_gpu_soak.py tests/test_bench_contract.py -m gpu -x -q -s --durations=8 > gpurun_out/r05i/tail.txt 2>&1; echo "rc=$?" >> gpurun_out/r05i/tail.txt
grep -v "^Particles\|^Total" gpurun_out/r05i/tail.txt | tail -40 | cut -c1-250
bash tools/gpu/r05h.sh
