#!/bin/bash
# round 3, batch l: final evidence -- GPU suite, rocprofv3 + PMC of the headline scene and the garment, bench lines of all scenes
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03l; mkdir -p $O; cd $R
timeout 1500 python -m pytest tests -m gpu -q --durations=6 > $O/pytest_gpu.log 2>&1; tail -14 $O/pytest_gpu.log
bash tools/gpu/profile_scene.sh sheet-500k r03 > $O/prof_sheet.log 2>&1; tail -3 $O/prof_sheet.log
bash tools/gpu/profile_scene.sh garment-120k-aniso r03 > $O/prof_garment.log 2>&1; tail -3 $O/prof_garment.log
for sc in cube-8k block-512k demo-250; do
  python bench.py --scene $sc --steps 400 --warmup 40 --no-cpu-baseline > $R/gpurun_out/r03l/bench_$sc.json 2>/dev/null
  python -c "
import json; d=json.load(open('$R/gpurun_out/r03l/bench_$sc.json')); print('$sc', round(d['value']), round(d['ms_per_step']*1e3,2), d.get('value_draped'), [(k['name'], round(k['ms']*1e3,2)) for k in d['kernels'][:3]])"
done
