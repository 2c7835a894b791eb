#!/bin/bash
# p2g's chunk tile in packed fixed point (two 32-bit channels per ds_add_u64; default) against the fp64 tile (MPMHIP_P2G_TILE=f64)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03fx; rm -f gpurun_out/r03fx/bench.txt
(python tools/gpu/grid_nodes.py; MPMHIP_P2G_TILE=f64 python tools/gpu/grid_nodes.py) 2>&1 | grep -v "^Particles\|^Total\|amdgpu.ids" | tee gpurun_out/r03fx/grid_nodes.txt
for rep in 1 2; do
for v in fixed f64; do
  for scene in sheet-500k garment-120k-aniso block-512k demo-250 cube-8k; do
    MPMHIP_P2G_TILE=$v timeout 600 python bench.py --scene $scene --steps 400 --warmup 40 --no-cpu-baseline --advance 2000 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$v $scene', round(d['value']), round(d['value_draped']), [(k['name'], round(k['ms']*1e3,1)) for k in d['kernels'][:3]])" | tee -a gpurun_out/r03fx/bench.txt
  done
done
done
timeout 2500 python -m pytest tests -m gpu -q 2>&1 | grep -E "FAILED|passed|failed" | tee gpurun_out/r03fx/tests_all.txt | tail -20
