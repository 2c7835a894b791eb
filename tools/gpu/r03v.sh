#!/bin/bash
# final evidence of round 3 after the re-sort rewrite: headline profile (bench line, rocprofv3 stats, PMC passes), the driver's
# default bench invocation, bench lines of the other scenes
cd $GRAFT_REPO_ROOT
bash tools/gpu/profile_scene.sh sheet-500k r03f
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03v
python bench.py > gpurun_out/r03v/default_bench.json 2> gpurun_out/r03v/default_bench.err; tail -c 400 gpurun_out/r03v/default_bench.json; echo
for sc in garment-120k-aniso cube-8k block-512k demo-250; do
  python bench.py --scene $sc --steps 400 --warmup 40 --no-cpu-baseline > gpurun_out/r03v/bench_$sc.json 2>/dev/null
  python -c "
import json; d=json.load(open('gpurun_out/r03v/bench_$sc.json')); print('$sc', round(d['value']), round(d.get('value_draped') or 0), d.get('rebins_in_window'))"
done
