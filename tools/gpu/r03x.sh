#!/bin/bash
# several chunks of a block per g2p workgroup (k_g2p_multi): MPMHIP_G2P_MERGE = 0 (never) / 1 (policy) / 2 (always), t = 0 and late
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03x; rm -f gpurun_out/r03x/*.txt
MPMHIP_G2P_MERGE=2 timeout 1200 python -m pytest tests/test_gpu_edges.py tests/test_gpu_fuzz.py tests/test_gpu_parity.py -x -q 2>&1 | tail -4 | tee -a gpurun_out/r03x/tests.txt
for rep in 1 2; do
for m in 0 1 2; do
  for scene in block-512k demo-250 cube-8k; do
    MPMHIP_G2P_MERGE=$m timeout 600 python bench.py --scene $scene --steps 400 --warmup 40 --no-cpu-baseline --advance 2000 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('merge=$m $scene', round(d['value']), round(d['value_draped']), [(k['name'], round(k['ms']*1e3,1)) for k in d['kernels'][:3]])" | tee -a gpurun_out/r03x/bench.txt
  done
done
done
