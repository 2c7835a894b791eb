#!/bin/bash
# round-4 evidence: the whole GPU suite, then rocprofv3 kernel trace + PMC passes + bench lines for the headline scene and the garment
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04h; mkdir -p $O; cd $R
timeout 2400 python -m pytest tests -m gpu -x -q > $O/pytest.txt 2>&1; tail -6 $O/pytest.txt
bash tools/gpu/profile_scene.sh sheet-500k r04 > $O/prof_sheet.txt 2>&1; tail -3 $O/prof_sheet.txt
bash tools/gpu/profile_scene.sh garment-120k-aniso r04 > $O/prof_garment.txt 2>&1; tail -3 $O/prof_garment.txt
python bench.py --steps 20 --warmup 5 > $O/bench_driver_args.json 2> $O/bench_driver_args.err; head -c 400 $O/bench_driver_args.json; echo
for sc in cube-8k block-512k demo-250 garment-120k-iso; do
  python bench.py --scene $sc --steps 400 --warmup 40 --no-cpu-baseline > $O/${sc}.json 2> $O/${sc}.err
done
ls $O
