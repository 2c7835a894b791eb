#!/bin/bash
# run on the GPU box through gpurun: bench (fast + baseline), rocprofv3 kernel trace + stats, PMC passes
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/prof
cd $R
python bench.py --steps 400 --warmup 40 > gpurun_out/bench_fast.json 2> gpurun_out/bench_fast.err
cat gpurun_out/bench_fast.json | tail -1
python bench.py --mode baseline --steps 40 --warmup 5 --no-cpu-baseline > gpurun_out/bench_baseline.json 2> gpurun_out/bench_baseline.err
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof -o fast -- python $R/bench.py --steps 200 --warmup 40 --no-cpu-baseline --no-kernels > $R/gpurun_out/prof/fast.log 2>&1
ls $R/gpurun_out/prof
for set in "FETCH_SIZE" "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum" "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_ANY"; do
  n=$(echo $set | cut -d' ' -f1)
  rocprofv3 --kernel-trace --pmc $set -d $R/gpurun_out/prof/pmc_$n -o pmc --output-format csv -- python $R/bench.py --steps 60 --warmup 10 --no-cpu-baseline --no-kernels > $R/gpurun_out/prof/pmc_$n.log 2>&1
done
