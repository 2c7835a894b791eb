#!/bin/bash
# compiler scheduling strategies for csrc/fast.hip (-mllvm -amdgpu-sched-strategy=...): headline + garment + block, event-bracketed kernels
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03y; rm -f gpurun_out/r03y/*.txt
V=$GRAFT_REPO_ROOT/mpmavatar_amd/lib/variants
for rep in 1 2; do
for v in default ilp memclause maxocc itilp bias100; do
  [ $v = default ] && unset MPMHIP_LIB || export MPMHIP_LIB=$V/libmpmhip_$v.so
  [ $v != default ] && [ ! -f $MPMHIP_LIB ] && continue
  for scene in sheet-500k garment-120k-aniso block-512k; do
    timeout 600 python bench.py --scene $scene --steps 400 --warmup 40 --no-cpu-baseline --advance 0 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$v $scene', round(d['value']), [(k['name'], round(k['ms']*1e3,1)) for k in d['kernels'][:3]])" | tee -a gpurun_out/r03y/bench.txt
  done
done
done
