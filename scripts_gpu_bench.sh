#!/bin/bash
# helper run on the GPU box through gpurun: bench (fast + baseline) and a rocprofv3 kernel trace
set -x
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
cd $R
python bench.py --steps 200 --warmup 40 > gpurun_out/bench_fast.json 2> gpurun_out/bench_fast.err
tail -3 gpurun_out/bench_fast.err
cat gpurun_out/bench_fast.json
python bench.py --mode baseline --steps 40 --warmup 5 --no-cpu-baseline > gpurun_out/bench_baseline.json 2> gpurun_out/bench_baseline.err
cat gpurun_out/bench_baseline.json
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_fast -o fast -- python $R/bench.py --steps 100 --warmup 40 --no-cpu-baseline --no-kernels > $R/gpurun_out/prof_fast.log 2>&1
tail -2 $R/gpurun_out/prof_fast.log
ls -R $R/gpurun_out/prof_fast | head -20
