#!/bin/bash
cd $GRAFT_REPO_ROOT
for cfg in "A=1" "MPMHIP_G2P_TWO_PASS=0"; do for sc in sheet-500k garment-120k-aniso demo-250; do
  env $cfg python bench.py --scene $sc --steps 400 --warmup 40 --no-cpu-baseline --advance 0 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); k={x['phase']:round(x['ms']*1e3,1) for x in d['kernels']}; print('$cfg $sc', round(d['value']), round(d['ms_per_step']*1e3,1), k)"
done; done
