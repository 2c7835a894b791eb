#!/bin/bash
# what the GPU box's host really has (for the oracle ensemble of tests/test_gpu_fullsize.py): visible CPUs, cgroup quota, and how
# 1 / 3 / 5 concurrent 16-thread (and 8-thread) oracle processes scale on the 500k sheet
echo "nproc $(nproc)  cpu.max $(cat /sys/fs/cgroup/cpu.max 2>/dev/null)  affinity $(python -c 'import os; print(len(os.sched_getaffinity(0)))')"
lscpu | grep -E "Model name|Socket|Core|Thread" 
for cfg in "1 16" "3 16" "5 16" "5 8" "5 12"; do
  set -- $cfg; n=$1; t=$2
  s=$(date +%s.%N)
  for i in $(seq 1 $n); do python tests/oracle_worker.py sheet-500k 0 $t /tmp/ow_$i.npz 30 & done
  wait
  e=$(date +%s.%N)
  python - <<P
import numpy as np
print("procs $n threads $t: wall %.1f s, per-process oracle seconds" % ($e - $s), [round(float(np.load(f"/tmp/ow_{i}.npz")["seconds"]),1) for i in range(1,$n+1)])
P
done
