#!/bin/bash
# G2P2G: correctness (traditional scenes of the suite) and rate, fused launch on / off
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04f; mkdir -p $O; cd $R
timeout 1200 python -m pytest tests/test_gpu_ref_golden.py tests/test_gpu_parity.py tests/test_gpu_edges.py tests/test_gpu_fuzz.py tests/test_gpu_api.py tests/test_gpu_golden.py -m gpu -x -q > $O/pytest.txt 2>&1; tail -8 $O/pytest.txt
for sc in cube-8k block-512k garment-120k-iso; do
  for on in 1 0; do
    MPMHIP_G2P2G=$on python bench.py --scene $sc --steps 400 --warmup 40 --no-cpu-baseline > $O/${sc}_g$on.json 2> $O/${sc}_g$on.err
    python - <<PY
import json
try:
  o=json.loads([l for l in open("$O/${sc}_g$on.json") if l.startswith("{")][-1])
  print("$sc g2p2g=$on", round(o["value"]), "draped", round(o.get("value_draped",0)), [(k["name"],round(k["ms"]*1e3,2),round(k["ms_events"]*1e3,2)) for k in o.get("kernels",[]) if k["name"].startswith("k_")])
except Exception as e:
  print("$sc g2p2g=$on FAILED", e); print(open("$O/${sc}_g$on.err").read()[-800:])
PY
  done
done
