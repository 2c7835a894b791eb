#!/bin/bash
# end-of-round check as the driver runs it: GPU suite, smoke, default bench, driver-args bench
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04m; mkdir -p $O; cd $R
timeout 2700 python -m pytest tests -m gpu -q > $O/pytest.txt 2>&1; tail -4 $O/pytest.txt
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1; tail -2 $O/smoke.txt
( time python bench.py > $O/bench_default.json 2> $O/bench_default.err ) 2>&1 | grep real
python bench.py --steps 20 --warmup 5 > $O/bench_20_5.json 2> $O/bench_20_5.err
python - <<PY
import json
for f in ("bench_default","bench_20_5"):
    o=json.loads([l for l in open("$O/"+f+".json") if l.startswith("{")][-1])
    print(f, round(o["value"]), o["windows"], round(o["roofline"]["frac"],3), round(o.get("value_draped",0)), round(o["cpu_baseline"]["value"],1), [(k["name"],round(k["ms"]*1e3,2)) for k in o["kernels"] if k["name"].startswith("k_")])
PY
