#!/bin/bash
# round 3, batch d: g2p with all record-dependent loads issued up front (MFLAG off = default), p2g first-round stagger A/B, cov test
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03d; mkdir -p $O; cd $R
V=$R/mpmavatar_amd/lib/variants
one() {  # label, scene, env...
  local label=$1 sc=$2; shift 2
  env "$@" python bench.py --scene $sc --steps 400 --warmup 40 --no-cpu-baseline --advance 0 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
k={x['phase']:x['ms']*1e3 for x in d.get('kernels',[])}
print('%-28s %-20s %8.0f /s  %6.2f us | stress %5.2f p2g %5.2f g2p %5.2f' % ('$label','$sc',d['value'],d['ms_per_step']*1e3,k.get('compute_stress_from_F_trial',0),k.get('p2g',0),k.get('g2p_v',0)))"
}
for sc in sheet-500k garment-120k-aniso demo-250 block-512k cube-8k; do
  one default $sc A=1
  one mflag $sc MPMHIP_G2P_MFLAG=1
  one stagger4x2 $sc MPMHIP_P2G_STAGGER=4,2
  one stagger8x2 $sc MPMHIP_P2G_STAGGER=8,2
  one stagger3x3 $sc MPMHIP_P2G_STAGGER=3,3
  one stagger2x5 $sc MPMHIP_P2G_STAGGER=2,5
done 2>&1 | tee $O/ab.txt
timeout 600 python -m pytest tests/test_gpu_api.py -q -k "cov" 2>&1 | tail -15 | tee $O/pytest_cov.txt
timeout 900 python -m pytest tests/test_gpu_ref_golden.py tests/test_gpu_parity.py tests/test_gpu_edges.py tests/test_gpu_golden.py -q 2>&1 | tail -6 | tee $O/pytest.txt
MPMHIP_LIB=$V/libmpmhip_dbg.so python tools/gpu/wgtrace.py sheet-500k 100 r03d > $O/wgtrace_sheet.md 2>/dev/null; sed -n 5,32p $O/wgtrace_sheet.md
MPMHIP_LIB=$V/libmpmhip_dbg.so MPMHIP_P2G_STAGGER=4,2 python tools/gpu/wgtrace.py sheet-500k 100 r03d_st > $O/wgtrace_sheet_stagger.md 2>/dev/null; sed -n 5,20p $O/wgtrace_sheet_stagger.md
