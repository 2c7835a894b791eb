#!/bin/bash
# where did the generalised tile lose 2 us?  r03 library vs the round-4 code (two-pass g2p) with pairing off, with the old strides, with pairing on
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04e; mkdir -p $O; cd $R
run() { # name lib pair scene
  export MPMHIP_LIB=$2
  MPMHIP_PAIR=$3 python bench.py --scene $4 --steps 400 --warmup 40 --no-cpu-baseline --advance 0 > $O/$4_$1.json 2> $O/$4_$1.err
  python - <<PY
import json
o=json.loads([l for l in open("$O/$4_$1.json") if l.startswith("{")][-1])
print("$4 $1", round(o["value"]), [(k["name"],round(k["ms_events"]*1e3,2)) for k in o.get("kernels",[]) if k["name"].startswith("k_")])
PY
}
V=$R/mpmavatar_amd/lib/variants
for rep in 1 2; do
for sc in sheet-500k garment-120k-aniso; do
  run r03 $V/libmpmhip_r03.so 0 $sc
  run new_pair0 $V/libmpmhip_old2p.so 0 $sc
  run new_s99_pair0 $V/libmpmhip_old2p_s99.so 0 $sc
  run new_pair1 $V/libmpmhip_old2p.so 1 $sc
done
done
