#!/bin/bash
# DPP wave sum for the fixed-point bounds: parity subset + bench
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03j2; rm -f gpurun_out/r03j2/*.txt
timeout 900 python -m pytest tests/test_gpu_ref_golden.py tests/test_gpu_parity.py tests/test_gpu_edges.py -x -q 2>&1 | tail -3
for rep in 1 2; do
  for scene in sheet-500k garment-120k-aniso demo-250 cube-8k block-512k; do
    timeout 600 python bench.py --scene $scene --steps 400 --warmup 40 --no-cpu-baseline --advance 2000 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$scene', round(d['value']), round(d['value_draped']), [(k['name'], round(k['ms']*1e3,1)) for k in d['kernels'][:3]])" | tee -a gpurun_out/r03j2/bench.txt
  done
done
