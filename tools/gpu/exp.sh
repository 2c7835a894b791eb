#!/bin/bash
# perf experiments: MPMHIP_DBG variants (bit0 skip p2g flush, bit1 skip p2g scatter, bit2 skip collider gather)
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out
for d in ${DBGS:-0 1 2 3}; do
  echo "== MPMHIP_DBG=$d"
  MPMHIP_DBG=$d python bench.py --steps 100 --warmup 40 --no-cpu-baseline ${BENCH_ARGS} 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        o=json.loads(l); print('ms/step',round(o['ms_per_step'],4), {k['name']:round(k['ms']*1000,1) for k in o['kernels']})
"
done
