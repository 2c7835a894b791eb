#!/bin/bash
# FETCH_SIZE / WRITE_SIZE against known byte counts (tools/ubench_fetch.hip): two PMC passes, each its own run with --kernel-trace only.
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04_fetch; mkdir -p $O; cd $R
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 tools/ubench_fetch.hip -o $O/ubench_fetch || exit 1
$O/ubench_fetch 2 > $O/plain.txt 2>&1; cat $O/plain.txt
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/pmc_fetch -o pmc --output-format csv -- $O/ubench_fetch 2 > $O/pmc_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/pmc_write -o pmc --output-format csv -- $O/ubench_fetch 2 > $O/pmc_write.log 2>&1
cd $R && python tools/summarize_fetch_calib.py
