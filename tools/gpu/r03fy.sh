#!/bin/bash
# hybrid (per-chunk) fixed-point / fp64 tile: bench A/B, demo envelope, mixed trace
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03fy; rm -f gpurun_out/r03fy/*.txt
python tools/gpu/grid_nodes.py 2>&1 | grep "mixed\|cloth\|jelly" | tee gpurun_out/r03fy/grid_nodes.txt
for rep in 1 2; do
for v in fixed f64; do
  for scene in sheet-500k garment-120k-aniso demo-250 cube-8k block-512k; do
    MPMHIP_P2G_TILE=$v timeout 600 python bench.py --scene $scene --steps 400 --warmup 40 --no-cpu-baseline --advance 2000 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$v $scene', round(d['value']), round(d['value_draped']), [(k['name'], round(k['ms']*1e3,1)) for k in d['kernels'][:3]])" | tee -a gpurun_out/r03fy/bench.txt
  done
done
done
python tools/gpu/tile_envelope.py demo-250 1000 2>&1 | grep -v "^Particles\|^Total\|amdgpu.ids" | tee gpurun_out/r03fy/tile_envelope_demo-250.txt
timeout 2500 python -m pytest tests -m gpu -q 2>&1 | grep -E "FAILED|passed|failed" | tee gpurun_out/r03fy/tests_all.txt | tail
