#!/bin/bash
mkdir -p gpurun_out/r05n
timeout 900 python -m pytest tests/test_gpu_fd.py -m gpu -x -q > gpurun_out/r05n/fd.txt 2>&1; echo "rc=$?" >> gpurun_out/r05n/fd.txt; tail -8 gpurun_out/r05n/fd.txt | cut -c1-300
python tools/gpu/fd_bench.py > gpurun_out/r05n/fd_bench_120k.txt 2>&1; cp gpurun_out/fd_bench.json gpurun_out/r05n/fd_bench_120k.json
python tools/gpu/fd_bench.py 100 100 64 2 400 > gpurun_out/r05n/fd_bench_30k.txt 2>&1; cp gpurun_out/fd_bench.json gpurun_out/r05n/fd_bench_30k.json
grep -h "^sequential\|^concurrent\|^batched" gpurun_out/r05n/fd_bench_*.txt | cut -c1-120
python tools/gpu/ab5.py --libs "default,default@MPMHIP_BATCH_SINGLE=1" --scenes sheet-500k,garment-120k-aniso --reps 2 --advance 2000 --out gpurun_out/r05n/ab.json > gpurun_out/r05n/ab.txt 2>&1; cut -c1-260 gpurun_out/r05n/ab.txt
MPMHIP_BATCH_SINGLE=1 timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_ref_golden.py tests/test_gpu_edges.py -m gpu -x -q > gpurun_out/r05n/parity_single.txt 2>&1; tail -3 gpurun_out/r05n/parity_single.txt
