#!/bin/bash
# after the split of fast.hip: the whole GPU suite (no -x), and the headline / garment / cube rates
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04i; mkdir -p $O; cd $R
timeout 2700 python -m pytest tests -m gpu -q > $O/pytest.txt 2>&1; tail -8 $O/pytest.txt
for sc in sheet-500k garment-120k-aniso cube-8k; do
  python bench.py --scene $sc --steps 400 --warmup 40 --no-cpu-baseline > $O/$sc.json 2> $O/$sc.err
  python - <<PY
import json
o=json.loads([l for l in open("$O/$sc.json") if l.startswith("{")][-1])
print("$sc", round(o["value"]), "draped", round(o.get("value_draped",0)), [(k["name"],round(k["ms"]*1e3,2)) for k in o.get("kernels",[]) if k["name"].startswith("k_")])
PY
done
