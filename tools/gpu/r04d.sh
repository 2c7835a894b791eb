#!/bin/bash
# g2p cloth single sweep (rolled) vs two-pass vs rolled-i only; pairing policy p2g-only
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04d; mkdir -p $O; cd $R
#timeout 900 python -m pytest tests/test_gpu_ref_golden.py tests/test_gpu_parity.py tests/test_gpu_edges.py -m gpu -x -q > $O/pytest.txt 2>&1; tail -5 $O/pytest.txt
run() { # name lib pair scene
  if [ -n "$2" ]; then export MPMHIP_LIB=$2; else unset MPMHIP_LIB; fi
  MPMHIP_PAIR=$3 python bench.py --scene $4 --steps 400 --warmup 40 --no-cpu-baseline > $O/$4_$1.json 2> $O/$4_$1.err
  python - <<PY
import json
o=json.loads([l for l in open("$O/$4_$1.json") if l.startswith("{")][-1])
print("$4 $1", round(o["value"]), "draped", round(o.get("value_draped",0)), [(k["name"],round(k["ms_events"]*1e3,2)) for k in o.get("kernels",[]) if k["name"].startswith("k_")])
PY
}
V=$R/mpmavatar_amd/lib/variants
for sc in sheet-500k garment-120k-aniso demo-250; do
  run roll2_pair1 "" 1 $sc
  run roll2_pair0 "" 0 $sc
  run old2p_pair1 $V/libmpmhip_old2p.so 1 $sc
done
run roll2_pair1 "" 1 block-512k
run roll2_pair1 "" 1 cube-8k
