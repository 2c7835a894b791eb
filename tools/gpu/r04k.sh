#!/bin/bash
# first-round stagger of p2g re-tested with the round-4 kernels (headline scene, t = 0 and draped)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04k; mkdir -p $O; cd $R
for rep in 1 2 3; do
for st in "" "1,5" "2,5" "3,5" "2,3"; do
  if [ -n "$st" ]; then export MPMHIP_P2G_STAGGER=$st; else unset MPMHIP_P2G_STAGGER; fi
  python bench.py --scene sheet-500k --steps 400 --warmup 40 --no-cpu-baseline --no-kernels > $O/s.json 2> $O/s.err
  python - <<PY
import json
o=json.loads([l for l in open("$O/s.json") if l.startswith("{")][-1])
print("stagger '$st'", round(o["value"]), "draped", round(o.get("value_draped",0)))
PY
done
done
