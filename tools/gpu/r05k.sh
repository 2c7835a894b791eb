#!/bin/bash
# re-measure what the stress-ahead default (now off) touched: garment profile, FD step, garment soak; + two-rank bench line
R=$GRAFT_REPO_ROOT; cd $R; O=gpurun_out/r05k; mkdir -p $O
bash tools/gpu/profile_scene.sh garment-120k-aniso r05 > $O/prof_garment.log 2>&1
cd $R
python tools/gpu/fd_bench.py > $O/fd_bench_120k.txt 2>&1; cp gpurun_out/fd_bench.json $O/fd_bench_120k.json
python tools/gpu/fd_bench.py 100 100 64 2 400 > $O/fd_bench_30k.txt 2>&1; cp gpurun_out/fd_bench.json $O/fd_bench_30k.json
python tools/gpu/soak.py 20000 garment-120k-aniso > $O/soak_garment.txt 2>&1
MPMHIP_DIST_BACKEND=gloo OMP_NUM_THREADS=1 python bench.py --gpus 2 --steps 200 --warmup 20 --no-cpu-baseline > $O/bench_2rank_gloo.json 2> $O/bench_2rank_gloo.err
grep -h "sequential\|concurrent" $O/fd_bench_*.txt | cut -c1-160; tail -1 $O/soak_garment.txt | cut -c1-300; tail -c 400 gpurun_out/prof_r05_garment-120k-aniso/bench_fast.json
