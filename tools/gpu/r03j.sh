#!/bin/bash
# round 3, batch j: whole GPU suite (oracle at 16 threads, S4 1000 substeps in the suite), late state with splats behind the chunks
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03j; mkdir -p $O; cd $R
for cfg in "MPMHIP_SPLAT_FIRST_MAX=100000" "MPMHIP_SPLAT_FIRST_MAX=256" "MPMHIP_SPLAT_FIRST_MAX=0"; do
  env $cfg python bench.py --scene sheet-500k --steps 200 --warmup 20 --no-cpu-baseline --advance 0 --pre-advance 2000 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('$cfg late:', round(d['value']), round(d['ms_per_step']*1e3,1), 'us;', [(k['name'], round(k['ms']*1e3,2)) for k in d['kernels'][:3]])"
done 2>&1 | tee $O/late_order.txt
for cfg in "MPMHIP_SPLAT_FIRST_MAX=256" "MPMHIP_SPLAT_FIRST_MAX=0"; do
  env $cfg python bench.py --scene sheet-500k --steps 400 --warmup 40 --no-cpu-baseline --advance 0 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('$cfg t=0:', round(d['value']), round(d['ms_per_step']*1e3,1), 'us;', [(k['name'], round(k['ms']*1e3,2)) for k in d['kernels'][:3]])"
done 2>&1 | tee -a $O/late_order.txt
timeout 1500 python -m pytest tests -m gpu -q --durations=8 > $O/pytest_gpu.log 2>&1; tail -22 $O/pytest_gpu.log
