#!/bin/bash
# round-4 start: traffic calibration, and the bench line at the driver's arguments with the windowed timing
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04a; mkdir -p $O; cd $R
bash tools/gpu/r04_fetch_calib.sh > $O/calib.txt 2>&1; tail -30 $O/calib.txt
python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_20_5.json 2> $O/bench_20_5.err; head -c 900 $O/bench_20_5.json; echo
python bench.py --steps 400 --warmup 40 --no-cpu-baseline > $O/bench_400.json 2> $O/bench_400.err; head -c 900 $O/bench_400.json; echo
