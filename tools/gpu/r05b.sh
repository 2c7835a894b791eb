#!/bin/bash
# round 5, second GPU call: the suite again (driver's order, -x), then the two-rank gloo bench line three times for its shard_floor windows
mkdir -p gpurun_out/r05b
timeout 2400 python -m pytest tests -m gpu -x -q --durations=15 > gpurun_out/r05b/pytest.txt 2>&1; echo "pytest rc=$?" >> gpurun_out/r05b/pytest.txt
for i in 1 2 3; do
  MPMHIP_DIST_BACKEND=gloo OMP_NUM_THREADS=1 python bench.py --gpus 2 --scene cube-8k --steps 20 --warmup 5 --advance 0 --no-cpu-baseline --no-weak \
    2> gpurun_out/r05b/bench2_$i.err | python -c "import json,sys; o=json.loads(sys.stdin.read()); print(json.dumps({'value':o['value'],'shard_floor':o.get('shard_floor')}))" >> gpurun_out/r05b/shard_floor.txt
done
tail -12 gpurun_out/r05b/pytest.txt; cat gpurun_out/r05b/shard_floor.txt
