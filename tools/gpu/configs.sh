#!/bin/bash
# bench every BASELINE.json config stand-in (fast mode), one JSON line each
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out
for sc in cube-8k garment-120k-iso garment-120k-aniso block-512k demo-250 sheet-500k; do
  python bench.py --scene $sc --steps 200 --warmup 40 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/cfg_$sc.json
  python - <<PY
import json
o=json.load(open("gpurun_out/cfg_$sc.json"))
print("$sc", round(o["value"]), "substeps/s", round(o["ms_per_step"]*1e3,1), "us", "frac", round(o.get("substep_frac_of_hbm_peak",0),3), {k["name"][:12]: round(k["ms"]*1e3,1) for k in o["kernels"]}, "fallback", o["config"]["fallback_particles"], "rebins", o["config"]["rebins"])
PY
done
