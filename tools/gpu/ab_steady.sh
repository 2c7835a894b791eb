#!/bin/bash
# ab_steady.sh [pre_advance]: A/B of the working tree's library against lib/variants/libmpmhip_head.so in the STEADY state of the headline
# scene (after `pre_advance` substeps, default 8000: the sheet lies folded over the sphere), kernel-stamp microseconds, 400 substeps.
PRE=${1:-8000}
HEADLIB=$PWD/mpmavatar_amd/lib/variants/libmpmhip_head.so
for rep in 1 2; do for v in head new; do
  if [ $v = head ]; then export MPMHIP_LIB=$HEADLIB; else unset MPMHIP_LIB; fi
  python bench.py --pre-advance $PRE --steps 400 --warmup 40 --no-cpu-baseline --advance 0 2>/dev/null | python -c "import sys,json; o=json.loads(sys.stdin.read()); print(\"sheet-500k after $PRE: $v\", round(o[\"value\"]), [(k[\"name\"],round(k[\"ms\"]*1e3,2)) for k in o[\"kernels\"] if k[\"name\"].startswith(\"k_\")])"
done; done
