#!/bin/bash
# gate run at HEAD: smoke(), the full -m gpu suite in the driver's order, the default bench line
mkdir -p gpurun_out/r05l
git_head=$(cat .git_head 2>/dev/null)
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r05l/smoke.txt 2>&1; echo "smoke rc=$?" >> gpurun_out/r05l/smoke.txt
( time timeout 2400 python -m pytest tests -m gpu -x -q --durations=15 ) > gpurun_out/r05l/pytest.txt 2>&1; echo "pytest rc=$?" >> gpurun_out/r05l/pytest.txt
python bench.py > gpurun_out/r05l/bench.json 2> gpurun_out/r05l/bench.err
tail -3 gpurun_out/r05l/smoke.txt; tail -26 gpurun_out/r05l/pytest.txt; cut -c1-400 gpurun_out/r05l/bench.json
