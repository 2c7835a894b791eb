#!/bin/bash
# round 3, first verification batch: Givens QR in the cloth path -> reference-sequence report, GPU suite, bench line; profiler probes
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03a; mkdir -p $O; cd $R
python tools/gpu/ref_seq_report.py > $O/ref_seq_report.md 2> $O/ref_seq_report.err; tail -25 $O/ref_seq_report.md
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; tail -5 $O/pytest_gpu.log
python bench.py --steps 200 --warmup 40 > $O/bench.json 2> $O/bench.err; tail -c 1500 $O/bench.json; echo
rocprofv3-avail list --pc-sampling > $O/avail_pcs.txt 2>&1; rocprofv3-avail info --pc-sampling >> $O/avail_pcs.txt 2>&1; tail -20 $O/avail_pcs.txt
cd /tmp && export TMPDIR=/tmp
timeout 200 rocprofv3 --att --kernel-include-regex "k_p2g" --att-target-cu 1 -d $O/att -o att -- python $R/bench.py --scene sheet-500k --steps 3 --warmup 3 --no-cpu-baseline --no-kernels --advance 0 > $O/att.log 2>&1; echo "att rc=$?"; tail -5 $O/att.log; find $O/att -type f | head; du -sh $O/att
