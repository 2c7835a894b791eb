#!/bin/bash
# the driver's exact configuration (--steps 20 --warmup 5), repeated: default vs g2p with the m_flag hop
cd $GRAFT_REPO_ROOT
for rep in 1 2 3 4; do
for cfg in "A=1" "MPMHIP_G2P_MFLAG=1"; do
  env $cfg python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --advance 0 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); k={x['phase']:round(x['ms']*1e3,1) for x in d['kernels']}; print('$cfg', round(d['value']), round(d['ms_per_step']*1e3,1), k)"
done; done
