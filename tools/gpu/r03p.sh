#!/bin/bash
# round 3, batch p: final state -- smoke, whole GPU suite, default bench line
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03p; mkdir -p $O; cd $R
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep smoke
timeout 1500 python -m pytest tests -m gpu -q > $O/pytest_gpu.log 2>&1; tail -4 $O/pytest_gpu.log
python bench.py > $O/bench.json 2> $O/bench.err; python -c "
import json; d=json.load(open('$O/bench.json')); print(d['value'], d['ms_per_step'], d.get('value_draped'), d['cpu_baseline']['value'], d['cpu_baseline']['cores'], d['cpu_baseline']['serial']['value'], d['roofline']['frac'], d['roofline']['traffic_frac'], d['roofline']['traffic_source'])"
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --advance 0 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('20 steps:', d['value'])"
