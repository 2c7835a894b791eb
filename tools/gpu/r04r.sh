#!/bin/bash
# r04r: kernarg preload (-mllvm -amdgpu-kernarg-preload-count=16: the chunk-record pointer and count arrive in SGPRs with the wave,
# the record load does not wait for the kernarg s_load).  A/B of lib/variants/libmpmhip_preload.so against the default build.
V=$PWD/mpmavatar_amd/lib/variants/libmpmhip_${1:-preload}.so
MPMHIP_LIB=$V python -m pytest tests/test_gpu_parity.py tests/test_gpu_ref_golden.py -m gpu -q -x 2>&1 | tail -1
for rep in 1 2; do for scene in sheet-500k garment-120k-aniso cube-8k; do for v in base var; do
  if [ $v = var ]; then export MPMHIP_LIB=$V; else unset MPMHIP_LIB; fi
  python bench.py --scene $scene --steps 400 --warmup 40 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; o=json.loads(sys.stdin.read()); print(\"$scene $v\", round(o[\"value\"]), round(o.get(\"value_draped\") or 0), [(k[\"name\"],round(k[\"ms\"]*1e3,2)) for k in o[\"kernels\"] if k[\"name\"].startswith(\"k_\")])"
done; done; done
