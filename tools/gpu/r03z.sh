#!/bin/bash
# s_setprio: front of a workgroup (memory chain) over the VALU-bound body (-DMPMHIP_PRIO=1)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03z; rm -f gpurun_out/r03z/*.txt
V=$GRAFT_REPO_ROOT/mpmavatar_amd/lib/variants
for rep in 1 2; do
for v in default cp1 cp3; do
  [ $v = default ] && unset MPMHIP_LIB || export MPMHIP_LIB=$V/libmpmhip_$v.so
  for scene in sheet-500k garment-120k-aniso block-512k; do
    timeout 600 python bench.py --scene $scene --steps 400 --warmup 40 --no-cpu-baseline --advance 2000 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$v $scene', round(d['value']), round(d['value_draped']), [(k['name'], round(k['ms']*1e3,1)) for k in d['kernels'][:3]])" | tee -a gpurun_out/r03z/bench.txt
  done
done
done
