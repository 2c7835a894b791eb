#!/bin/bash
# quick A/B: fused substep time of the headline scene for a list of environment settings, e.g.
#   VARIANTS="MPMHIP_FUSE_GRID=1 MPMHIP_FUSE_GRID=0" bash tools/gpu/quick.sh
R=$GRAFT_REPO_ROOT; cd $R
for v in ${VARIANTS:-X=0}; do
  echo "== $v"
  env $v python bench.py --steps 200 --warmup 40 --no-cpu-baseline ${BENCH_ARGS} 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        o=json.loads(l); print('us/step',round(o['ms_per_step']*1e3,1), {k['name'][:14]:round(k['ms']*1000,1) for k in o.get('kernels',[])})
"
done
