#!/bin/bash
# A/B over scenes: VARIANTS="ENV=a ENV=b" SCENES="cube-8k block-512k" bash tools/gpu/ab.sh
R=$GRAFT_REPO_ROOT; cd $R
for sc in ${SCENES:-sheet-500k}; do
  for v in ${VARIANTS:-X=0}; do
    env $v python bench.py --scene $sc --steps 200 --warmup 40 --no-cpu-baseline --no-kernels 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        o=json.loads(l); print('$sc', '$v', 'us/step', round(o['ms_per_step']*1e3,1), round(o['value']), 'substeps/s')
"
  done
done
