#!/bin/bash
# paired chunks: GPU tests, then bench with pairing on / off on the scenes of DESIGN 4
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04b; mkdir -p $O; cd $R
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest.txt 2>&1; tail -15 $O/pytest.txt
for sc in sheet-500k garment-120k-aniso block-512k demo-250 cube-8k; do
  for pair in 1 0; do
    MPMHIP_PAIR=$pair MPMHIP_VERBOSE=1 python bench.py --scene $sc --steps 400 --warmup 40 --no-cpu-baseline > $O/${sc}_pair$pair.json 2> $O/${sc}_pair$pair.err
    python - <<PY
import json
o=json.loads([l for l in open("$O/${sc}_pair$pair.json") if l.startswith("{")][-1])
print("$sc pair=$pair", round(o["value"]), "draped", round(o.get("value_draped",0)), "ovh", round(o.get("event_overhead_ms",0)*1e3,2), [(k["name"],round(k["ms"]*1e3,2)) for k in o.get("kernels",[])])
PY
    grep "re-sort" $O/${sc}_pair$pair.err | head -2
  done
done
