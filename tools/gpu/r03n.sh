#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03n; mkdir -p $O; cd $R
timeout 900 python -m pytest tests/test_bench_contract.py tests/test_gpu_api.py tests/test_dist.py -q -m gpu --durations=4 2>&1 | tail -12 | tee $O/pytest.txt
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3
