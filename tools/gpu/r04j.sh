#!/bin/bash
# g2p staging in (block, node) order vs k-fastest
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04j; mkdir -p $O; cd $R
timeout 900 python -m pytest tests/test_gpu_ref_golden.py tests/test_gpu_parity.py tests/test_gpu_edges.py tests/test_gpu_g2p2g.py -m gpu -x -q > $O/pytest.txt 2>&1; tail -3 $O/pytest.txt
V=$R/mpmavatar_amd/lib/variants
for rep in 1 2; do
for sc in sheet-500k garment-120k-aniso block-512k cube-8k; do
  for lib in "" $V/libmpmhip_korder.so; do
    if [ -n "$lib" ]; then export MPMHIP_LIB=$lib; tag=korder; else unset MPMHIP_LIB; tag=border; fi
    python bench.py --scene $sc --steps 400 --warmup 40 --no-cpu-baseline > $O/${sc}_$tag.json 2> $O/${sc}_$tag.err
    python - <<PY
import json
o=json.loads([l for l in open("$O/${sc}_$tag.json") if l.startswith("{")][-1])
print("$sc $tag", round(o["value"]), "draped", round(o.get("value_draped",0)), [(k["name"],round(k["ms"]*1e3,2)) for k in o.get("kernels",[]) if k["name"].startswith("k_")])
PY
  done
done
done
