#!/bin/bash
mkdir -p gpurun_out/r05g
timeout 900 python -m pytest tests/test_gpu_stress_ahead.py -m gpu -q > gpurun_out/r05g/sa.txt 2>&1; echo "rc=$?" >> gpurun_out/r05g/sa.txt; tail -5 gpurun_out/r05g/sa.txt
timeout 2400 python -m pytest tests -m gpu -x -q --durations=12 > gpurun_out/r05g/pytest.txt 2>&1; echo "rc=$?" >> gpurun_out/r05g/pytest.txt; tail -22 gpurun_out/r05g/pytest.txt
