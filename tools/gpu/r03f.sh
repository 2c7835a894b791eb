#!/bin/bash
# round 3, batch f: micro-benchmarks (VALU issue rates, grid barrier vs kernel boundary), late-state kernel breakdown, S3 parity record, slow tests
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03f; mkdir -p $O; cd $R
hipcc --offload-arch=gfx950 -O3 tools/ubench_valu.hip -o /tmp/ubench_valu 2>/dev/null && timeout 120 /tmp/ubench_valu | tee $O/ubench_valu.txt
hipcc --offload-arch=gfx950 -O3 tools/ubench_gridbar.hip -o /tmp/ubench_gridbar 2>/dev/null && timeout 120 /tmp/ubench_gridbar | tee $O/ubench_gridbar.txt
python bench.py --scene sheet-500k --steps 200 --warmup 20 --no-cpu-baseline --advance 0 --pre-advance 2000 > $O/bench_late.json 2>/dev/null
python -c "
import json
d=json.load(open('$O/bench_late.json'))
print('late state:', d['value'], d['ms_per_step']*1e3, 'us;', [(k['name'], round(k['ms']*1e3,2), k.get('launches')) for k in d['kernels']], d['config'].get('rebins'))"
timeout 1500 python -m pytest tests/test_dist.py tests/test_gpu_fullsize.py -q --durations=12 2>&1 | tail -25 | tee $O/pytest_slow.txt
python tools/gpu/full_parity.py garment-120k-aniso 1000 > $O/full_parity_garment.log 2>&1; tail -14 $O/full_parity_garment.log
