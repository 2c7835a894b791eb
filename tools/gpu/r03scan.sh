#!/bin/bash
# DPP scan depth with the integer tile (ds_add_u64 barely depends on the lane count): 2 / 3 (default) / 4 steps
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03h; rm -f gpurun_out/r03h/*.txt
V=$GRAFT_REPO_ROOT/mpmavatar_amd/lib/variants
for rep in 1 2; do
for v in default st2 st4; do
  [ $v = default ] && unset MPMHIP_LIB || export MPMHIP_LIB=$V/libmpmhip_$v.so
  for scene in sheet-500k garment-120k-aniso demo-250; do
    timeout 600 python bench.py --scene $scene --steps 400 --warmup 40 --no-cpu-baseline --advance 2000 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$v $scene', round(d['value']), round(d['value_draped']), [(k['name'], round(k['ms']*1e3,1)) for k in d['kernels'][:3]])" | tee -a gpurun_out/r03h/bench.txt
  done
done
done
