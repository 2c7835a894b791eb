#!/bin/bash
# round 3, batch m: adaptive scan depth (draped state), parity subset, default bench line
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03m; mkdir -p $O; cd $R
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_ref_golden.py tests/test_gpu_edges.py tests/test_gpu_golden.py tests/test_gpu_fuzz.py -q -x 2>&1 | tail -4 | tee $O/pytest.txt
python bench.py --scene sheet-500k --steps 200 --warmup 20 --no-cpu-baseline --advance 0 --pre-advance 2000 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('late:', round(d['value']), round(d['ms_per_step']*1e3,1), 'us;', [(k['name'], round(k['ms']*1e3,2)) for k in d['kernels'][:3]])" | tee $O/late.txt
for sc in garment-120k-aniso demo-250; do python bench.py --scene $sc --steps 400 --warmup 40 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('$sc', round(d['value']), round(d['ms_per_step']*1e3,1), d.get('value_draped'), [(k['name'], round(k['ms']*1e3,2)) for k in d['kernels'][:3]])"; done | tee -a $O/late.txt
python bench.py > $O/bench_default.json 2> $O/bench_default.err; python -c "
import json; d=json.load(open('$O/bench_default.json')); print(d['value'], d['ms_per_step'], d.get('value_draped'), d['cpu_baseline'], d['roofline']['frac'], d['roofline'].get('traffic_frac'))"
