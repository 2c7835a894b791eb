#!/bin/bash
# g2p tile reads: ds_read_b96 (what hipcc picks) vs ds_read_b128 (occupancy 4) vs ds_read_b128 forced to 5 waves (21 spilled VGPRs)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04n; mkdir -p $O; cd $R
V=$R/mpmavatar_amd/lib/variants
for rep in 1 2; do
for sc in sheet-500k garment-120k-aniso; do
  for tag in b128 b96 b128w5; do
    if [ $tag = b128 ]; then unset MPMHIP_LIB; else export MPMHIP_LIB=$V/libmpmhip_$tag.so; fi
    python bench.py --scene $sc --steps 400 --warmup 40 --no-cpu-baseline > $O/${sc}_$tag.json 2> $O/${sc}_$tag.err
    python - <<PY
import json
o=json.loads([l for l in open("$O/${sc}_$tag.json") if l.startswith("{")][-1])
print("$sc $tag", round(o["value"]), "draped", round(o.get("value_draped",0)), [(k["name"],round(k["ms"]*1e3,2)) for k in o.get("kernels",[]) if k["name"].startswith("k_")])
PY
  done
done
done
