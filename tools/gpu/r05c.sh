#!/bin/bash
mkdir -p gpurun_out/r05c
bash tools/gpu/host_cores_probe.sh > gpurun_out/r05c/host_cores.txt 2>&1
( time python bench.py --scene cube-8k --steps 40 --warmup 10 --no-cpu-baseline ) > gpurun_out/r05c/live_bench.txt 2>&1
timeout 2400 python -m pytest tests/test_gpu_fullsize.py -m gpu -x -q -s --durations=10 > gpurun_out/r05c/fullsize.txt 2>&1; echo "rc=$?" >> gpurun_out/r05c/fullsize.txt
cat gpurun_out/r05c/host_cores.txt; tail -5 gpurun_out/r05c/live_bench.txt; grep -v "^Particles\|^Total" gpurun_out/r05c/fullsize.txt | tail -60
