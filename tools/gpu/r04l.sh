#!/bin/bash
# g2p first-round stagger (p2g stagger auto on)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04l; mkdir -p $O; cd $R
for rep in 1 2 3; do
for st in 0 1 2 3; do
  MPMHIP_G2P_STAGGER=$st python bench.py --scene sheet-500k --steps 400 --warmup 40 --no-cpu-baseline --no-kernels > $O/s.json 2> $O/s.err
  python - <<PY
import json
o=json.loads([l for l in open("$O/s.json") if l.startswith("{")][-1])
print("g2p stagger $st", round(o["value"]), "draped", round(o.get("value_draped",0)))
PY
done
done
