#!/bin/bash
# after the ds_read_b128 selection: parity subset + rates of all scenes
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04o; mkdir -p $O; cd $R
timeout 1500 python -m pytest tests/test_gpu_ref_golden.py tests/test_gpu_parity.py tests/test_gpu_edges.py tests/test_gpu_g2p2g.py tests/test_gpu_fuzz.py tests/test_gpu_api.py -m gpu -q > $O/pytest.txt 2>&1; tail -3 $O/pytest.txt
for sc in sheet-500k garment-120k-aniso block-512k demo-250 cube-8k garment-120k-iso; do
  python bench.py --scene $sc --steps 400 --warmup 40 --no-cpu-baseline > $O/$sc.json 2> $O/$sc.err
  python - <<PY
import json
o=json.loads([l for l in open("$O/$sc.json") if l.startswith("{")][-1])
print("$sc", round(o["value"]), "draped", round(o.get("value_draped",0)), [(k["name"],round(k["ms"]*1e3,2)) for k in o.get("kernels",[]) if k["name"].startswith("k_")])
PY
done
