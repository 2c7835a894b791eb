#!/bin/bash
# round 5 measurement set at HEAD: rocprofv3 + PMC for the headline scene and the garment, bench lines of the other scenes, full-size
# parity records, shard floors, FD step, soak
R=$GRAFT_REPO_ROOT; cd $R; O=gpurun_out/r05h; mkdir -p $O
bash tools/gpu/profile_scene.sh sheet-500k r05 > $O/prof_sheet.log 2>&1
bash tools/gpu/profile_scene.sh garment-120k-aniso r05 > $O/prof_garment.log 2>&1
cd $R
for sc in cube-8k garment-120k-iso block-512k demo-250; do python bench.py --scene $sc --steps 400 --warmup 40 --no-cpu-baseline > $O/bench_$sc.json 2> $O/bench_$sc.err; done
python bench.py > $O/bench_default.json 2> $O/bench_default.err
python bench.py --steps 20 --warmup 5 > $O/bench_driver.json 2> $O/bench_driver.err
python tools/gpu/shard_floor.py sheet-500k > $O/shard_floor.txt 2>&1
python tools/gpu/fd_bench.py > $O/fd_bench_on.txt 2>&1; cp gpurun_out/fd_bench.json $O/fd_bench_on.json
MPMHIP_STRESS_AHEAD=0 python tools/gpu/fd_bench.py > $O/fd_bench_off.txt 2>&1; cp gpurun_out/fd_bench.json $O/fd_bench_off.json
python tools/gpu/fd_bench.py 100 100 64 2 400 > $O/fd_bench_30k.txt 2>&1
python tools/gpu/soak.py 20000 sheet-500k garment-120k-aniso > $O/soak.txt 2>&1
for sc in cube-8k garment-120k-iso garment-120k-aniso garment-120k-aniso@gamma0 sheet-500k sheet-500k@gamma0 block-512k demo-250; do timeout 900 python tools/gpu/full_parity.py $sc 1000 > $O/full_parity_$sc.txt 2>&1; done
grep -h "substeps/s\|sequential\|concurrent" $O/fd_bench_*.txt | cut -c1-200; cat $O/shard_floor.txt | grep -v "^Particles\|^Total"; tail -3 $O/soak.txt; grep -h "first substep" $O/full_parity_*.txt
