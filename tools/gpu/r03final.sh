#!/bin/bash
# evidence for the fixed-point tile (default): headline + garment profiles (bench line, rocprofv3 stats, PMC), the driver's default
# invocation, the other scenes' bench lines, and the 1000-substep parity protocol at full size on all seven scene variants
cd $GRAFT_REPO_ROOT
bash tools/gpu/profile_scene.sh sheet-500k r03g
cd $GRAFT_REPO_ROOT
bash tools/gpu/profile_scene.sh garment-120k-aniso r03g
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03g
python bench.py > gpurun_out/r03g/default_bench.json 2> gpurun_out/r03g/default_bench.err
for sc in cube-8k block-512k demo-250; do
  python bench.py --scene $sc --steps 400 --warmup 40 --no-cpu-baseline > gpurun_out/r03g/bench_$sc.json 2>/dev/null
done
for sc in sheet-500k garment-120k-aniso garment-120k-aniso@gamma0 garment-120k-iso cube-8k block-512k demo-250; do
  timeout 900 python tools/gpu/full_parity.py $sc 1000 2>&1 | tail -3
done
