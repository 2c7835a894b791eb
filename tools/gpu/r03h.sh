#!/bin/bash
# round 3, batch h: S4 / S3 1000-substep parity records (oracle at a sane thread count), dist + api tests, late-state with the small-bin splat
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03h; mkdir -p $O; cd $R
python tools/gpu/full_parity.py sheet-500k 1000 > $O/full_parity_sheet.log 2>&1; grep "^substep\|^first" $O/full_parity_sheet.log | tail -18
python tools/gpu/full_parity.py garment-120k-aniso 1000 > $O/full_parity_garment.log 2>&1; grep "^substep\|^first" $O/full_parity_garment.log | tail -5
cp gpurun_out/full_parity_*.json $O/ 2>/dev/null
timeout 1200 python -m pytest tests/test_dist.py tests/test_gpu_api.py tests/test_bench_contract.py tests/test_gpu_edges.py tests/test_gpu_parity.py -q --durations=5 2>&1 | tail -12 | tee $O/pytest.txt
python bench.py --scene sheet-500k --steps 200 --warmup 20 --no-cpu-baseline --advance 0 --pre-advance 2000 > $O/bench_late.json 2>/dev/null
python -c "
import json
d=json.load(open('$O/bench_late.json'))
print('late state:', d['value'], d['ms_per_step']*1e3, 'us;', [(k['name'], round(k['ms']*1e3,2), k.get('launches')) for k in d['kernels']])"
python bench.py --steps 200 --warmup 40 > $O/bench.json 2> $O/bench.err; python -c "
import json
d=json.load(open('$O/bench.json')); print(d['value'], d.get('value_draped'), d['cpu_baseline'], d['roofline'])"
MPMHIP_LIB=$R/mpmavatar_amd/lib/variants/libmpmhip_dbg.so python tools/gpu/wgtrace.py sheet-500k 2200 r03h_late > $O/wgtrace_sheet_late.md 2>/dev/null; sed -n 5,20p $O/wgtrace_sheet_late.md
