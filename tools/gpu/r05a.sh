#!/bin/bash
# round 5, first GPU call: the full -m gpu suite at HEAD in the driver's order (-x), smoke(), the shard-floor probe
mkdir -p gpurun_out/r05a
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r05a/smoke.txt 2>&1; echo "smoke rc=$?" >> gpurun_out/r05a/smoke.txt
python tools/gpu/shard_floor_probe.py > gpurun_out/r05a/shard_floor_probe.txt 2>&1
timeout 2400 python -m pytest tests -m gpu -x -q --durations=25 > gpurun_out/r05a/pytest.txt 2>&1; echo "pytest rc=$?" >> gpurun_out/r05a/pytest.txt
tail -5 gpurun_out/r05a/smoke.txt; tail -15 gpurun_out/r05a/pytest.txt
