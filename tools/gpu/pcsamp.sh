#!/bin/bash
# PC sampling (rocprofv3 beta) of the timed loop of one scene:  bash tools/gpu/pcsamp.sh <scene> <tag> [advance]
# stochastic sampling first (gfx950 hardware sampling: every sample carries the issue / stall reason of its wavefront),
# host-trap sampling as the fallback.  Output under gpurun_out/pcs_<tag>_<scene>/ ; tools/summarize_pcs.py condenses it.
SC=$1; TAG=$2; ADV=${3:-0}; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/pcs_${TAG}_${SC}; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
ARGS="--scene $SC --steps 300 --warmup 40 --no-cpu-baseline --no-kernels --advance 0 --pre-advance $ADV"
timeout 300 rocprofv3 --kernel-trace --pc-sampling-beta-enabled --pc-sampling-method stochastic --pc-sampling-unit cycles \
  --pc-sampling-interval 1048576 --output-format csv -d $O/st -o pcs -- python $R/bench.py $ARGS > $O/st.log 2>&1
echo "stochastic rc=$?"; tail -3 $O/st.log
timeout 300 rocprofv3 --kernel-trace --pc-sampling-beta-enabled --pc-sampling-method host_trap --pc-sampling-unit time \
  --pc-sampling-interval 1 --output-format csv -d $O/ht -o pcs -- python $R/bench.py $ARGS > $O/ht.log 2>&1
echo "host_trap rc=$?"; tail -3 $O/ht.log
find $O -type f | head -30; du -sh $O
