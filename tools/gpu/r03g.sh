#!/bin/bash
# round 3, batch g: oracle thread scaling; dist tests with the fused halo; stale-view test; shard floor
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03g; mkdir -p $O; cd $R
python tools/gpu/oracle_threads.py garment-120k-aniso 6 2>/dev/null | tee $O/oracle_threads.txt
python tools/gpu/oracle_threads.py sheet-500k 3 2>/dev/null | tee -a $O/oracle_threads.txt
timeout 1200 python -m pytest tests/test_dist.py tests/test_gpu_api.py tests/test_bench_contract.py -q -x --durations=5 2>&1 | tail -25 | tee $O/pytest_dist.txt
python tools/gpu/shard_floor.py > $O/shard_floor.txt 2>&1; tail -20 $O/shard_floor.txt
