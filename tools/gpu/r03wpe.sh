#!/bin/bash
# five wavefronts per SIMD for the traditional p2g instantiations (-DMPMHIP_P2G_WPE=5: 108 -> 96 VGPRs + 6 spilled dwords)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03wpe; rm -f gpurun_out/r03wpe/*.txt
V=$GRAFT_REPO_ROOT/mpmavatar_amd/lib/variants
for rep in 1 2; do
for v in default wpe5; do
  [ $v = default ] && unset MPMHIP_LIB || export MPMHIP_LIB=$V/libmpmhip_$v.so
  for scene in block-512k cube-8k demo-250; do
    timeout 600 python bench.py --scene $scene --steps 400 --warmup 40 --no-cpu-baseline --advance 2000 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$v $scene', round(d['value']), round(d['value_draped']), [(k['name'], round(k['ms']*1e3,1)) for k in d['kernels'][:3]])" | tee -a gpurun_out/r03wpe/bench.txt
  done
done
done
