#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03i; mkdir -p $O; cd $R
timeout 600 python -m pytest tests/test_gpu_api.py -q -x -k "reset_state or stale" 2>&1 | tail -60 | tee $O/pytest_api.txt
timeout 600 python -m pytest tests/test_dist.py -q -x -k "migration_over" 2>&1 | tail -40 | tee $O/pytest_mig.txt
