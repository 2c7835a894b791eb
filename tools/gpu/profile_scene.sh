#!/bin/bash
# rocprofv3 evidence for one scene (run on the GPU box through gpurun):  bash tools/gpu/profile_scene.sh <scene> <tag>
#   1. bench line (fused-loop kernel events, phases, draped value, CPU baseline only for the headline scene)
#   2. rocprofv3 --kernel-trace --stats of the timed loop
#   3. PMC passes, each in its own run with --kernel-trace only: FETCH_SIZE | WRITE_SIZE + L2 hit/miss | SQ instruction mix |
#      SQ busy / wait
# outputs under gpurun_out/prof_<tag>_<scene>/ ; tools/summarize_prof.py condenses them into profiles/
SC=$1; TAG=$2; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/prof_${TAG}_${SC}; mkdir -p $O; cd $R
CPU="--no-cpu-baseline"; [ "$SC" = "sheet-500k" ] && CPU=""
python bench.py --scene $SC --steps 400 --warmup 40 $CPU > $O/bench_fast.json 2> $O/bench_fast.err
tail -c 600 $O/bench_fast.json; echo
cd /tmp && export TMPDIR=/tmp
ARGS="--scene $SC --steps 200 --warmup 40 --no-cpu-baseline --no-kernels --advance 0"
rocprofv3 --kernel-trace --stats --output-format csv -d $O -o fast -- python $R/bench.py $ARGS > $O/fast.log 2>&1
i=0
for set in "FETCH_SIZE" "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum" "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU" \
           "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $set -d $O/pmc_$i -o pmc --output-format csv -- python $R/bench.py --scene $SC --steps 40 --warmup 8 --no-cpu-baseline --no-kernels --advance 0 > $O/pmc_$i.log 2>&1
done
ls $O | head -20
