#!/bin/bash
mkdir -p gpurun_out/r05d
python tools/gpu/ab5.py --libs default,nod3,g2pocc6,occ6,twopass --scenes sheet-500k,garment-120k-aniso --reps 2 --advance 2000 --out gpurun_out/r05d/ab_g2p_d3.json > gpurun_out/r05d/ab_g2p_d3.txt 2>&1
( time MPMHIP_DIST_BACKEND=gloo OMP_NUM_THREADS=1 python bench.py --gpus 2 --scene cube-8k --steps 20 --warmup 5 --advance 0 --no-cpu-baseline --weak-n 48 --weak-grid 64 ) > gpurun_out/r05d/bench2_marks.txt 2>&1
( time python bench.py --scene cube-8k --steps 40 --warmup 10 --no-cpu-baseline ) > gpurun_out/r05d/bench1_marks.txt 2>&1
timeout 1200 python -m pytest tests/test_gpu_fullsize.py -m gpu -x -q -s --durations=10 -k "s4_sheet_500k_1000 or s3_one_frame" > gpurun_out/r05d/fullsize.txt 2>&1; echo "rc=$?" >> gpurun_out/r05d/fullsize.txt
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_ref_golden.py tests/test_gpu_golden.py tests/test_dist.py -m gpu -x -q --durations=8 > gpurun_out/r05d/parity.txt 2>&1; echo "rc=$?" >> gpurun_out/r05d/parity.txt
cat gpurun_out/r05d/ab_g2p_d3.txt; grep "bench +\|real" gpurun_out/r05d/bench2_marks.txt gpurun_out/r05d/bench1_marks.txt; tail -8 gpurun_out/r05d/fullsize.txt | cut -c1-200; tail -14 gpurun_out/r05d/parity.txt
