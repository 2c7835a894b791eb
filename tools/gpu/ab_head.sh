#!/bin/bash
# ab_head.sh [scenes...]: parity subset with the working tree's library, then an alternating A/B of it against
# lib/variants/libmpmhip_head.so (the previous commit's build), kernel-stamp microseconds, 400 substeps.
SCENES=${@:-sheet-500k garment-120k-aniso demo-250}
python -m pytest tests/test_gpu_parity.py tests/test_gpu_ref_golden.py tests/test_gpu_edges.py -m gpu -q -x 2>&1 | tail -2
HEADLIB=$PWD/mpmavatar_amd/lib/variants/libmpmhip_head.so
for rep in 1 2; do for scene in $SCENES; do for v in head new; do
  if [ $v = head ]; then export MPMHIP_LIB=$HEADLIB; else unset MPMHIP_LIB; fi
  python bench.py --scene $scene --steps 400 --warmup 40 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; o=json.loads(sys.stdin.read()); print(\"$scene $v\", round(o[\"value\"]), round(o.get(\"value_draped\") or 0), [(k[\"name\"],round(k[\"ms\"]*1e3,2)) for k in o[\"kernels\"] if k[\"name\"].startswith(\"k_\")])"
done; done; done
