#!/bin/bash
# body-face splat split over the stress and p2g launches: correctness subset, then on / off on the scenes with a body
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04g; mkdir -p $O; cd $R
timeout 1500 python -m pytest tests/test_gpu_ref_golden.py tests/test_gpu_parity.py tests/test_gpu_edges.py tests/test_gpu_api.py tests/test_gpu_golden.py tests/test_gpu_fuzz.py -m gpu -x -q > $O/pytest.txt 2>&1; tail -4 $O/pytest.txt
for rep in 1 2; do
for sc in garment-120k-aniso sheet-500k demo-250; do
  for on in 1 0; do
    MPMHIP_SPLIT_SPLAT=$on python bench.py --scene $sc --steps 400 --warmup 40 --no-cpu-baseline > $O/${sc}_s$on.json 2> $O/${sc}_s$on.err
    python - <<PY
import json
o=json.loads([l for l in open("$O/${sc}_s$on.json") if l.startswith("{")][-1])
print("$sc split=$on", round(o["value"]), "draped", round(o.get("value_draped",0)), [(k["name"],round(k["ms"]*1e3,2)) for k in o.get("kernels",[]) if k["name"].startswith("k_")])
PY
  done
done
done
