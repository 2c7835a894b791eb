#!/bin/bash
# final evidence of round 4: full GPU suite, rocprofv3 + PMC for the headline scene and the garment (tag given as $1, default r04g), bench lines of all scenes
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${1:-r04g}_final; mkdir -p $O; cd $R
if [ -z "$SKIP_TESTS" ]; then timeout 2700 python -m pytest tests -m gpu -q > $O/pytest.txt 2>&1; tail -4 $O/pytest.txt; fi
bash tools/gpu/profile_scene.sh sheet-500k ${1:-r04g} > $O/prof_sheet.txt 2>&1; tail -2 $O/prof_sheet.txt
bash tools/gpu/profile_scene.sh garment-120k-aniso ${1:-r04g} > $O/prof_garment.txt 2>&1; tail -2 $O/prof_garment.txt
python bench.py --steps 20 --warmup 5 > $O/bench_driver_args.json 2> $O/bench_driver_args.err
python bench.py > $O/bench_default.json 2> $O/bench_default.err
for sc in cube-8k block-512k demo-250 garment-120k-iso; do
  python bench.py --scene $sc --steps 400 --warmup 40 --no-cpu-baseline > $O/${sc}.json 2> $O/${sc}.err
done
ls $O | wc -l
