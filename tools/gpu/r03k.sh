#!/bin/bash
# round 3, batch k: p2g + g2p as one launch (device-side wait) A/B
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03k; mkdir -p $O; cd $R
V=$R/mpmavatar_amd/lib/variants
one() { local label=$1 sc=$2; shift 2
  env "$@" timeout 300 python bench.py --scene $sc --steps 400 --warmup 40 --no-cpu-baseline --no-kernels --advance 0 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('%-22s %-20s %8.0f /s  %6.2f us' % ('$label','$sc',d['value'],d['ms_per_step']*1e3))"; }
for sc in sheet-500k garment-120k-aniso cube-8k block-512k demo-250; do
  one separate $sc MPMHIP_MERGE_G2P=0
  one merged $sc MPMHIP_MERGE_G2P=1
  one merged-wpe5 $sc MPMHIP_MERGE_G2P=1 MPMHIP_LIB=$V/libmpmhip_m5.so
done 2>&1 | tee $O/ab.txt
MPMHIP_MERGE_G2P=1 timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_ref_golden.py tests/test_gpu_edges.py tests/test_gpu_golden.py tests/test_gpu_api.py -q -x 2>&1 | tail -5 | tee $O/pytest.txt
