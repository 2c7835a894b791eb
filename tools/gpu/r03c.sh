#!/bin/bash
# round 3, batch c: A/B of six wavefronts per SIMD (MPMHIP_W6), g2p without the m_flag hop (MPMHIP_G2P_MFLAG=0), kernarg preload
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03c; mkdir -p $O; cd $R
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
  one base $sc A=1
  one w6 $sc MPMHIP_W6=1
  one nomflag $sc MPMHIP_G2P_MFLAG=0
  one w6+nomflag $sc MPMHIP_W6=1 MPMHIP_G2P_MFLAG=0
  one preload $sc MPMHIP_LIB=$V/libmpmhip_preload.so
  one preload+w6+nomflag $sc MPMHIP_LIB=$V/libmpmhip_preload.so MPMHIP_W6=1 MPMHIP_G2P_MFLAG=0
done 2>&1 | tee $O/ab.txt
echo "--- draped state"
for cfg in "A=1" "MPMHIP_W6=1 MPMHIP_G2P_MFLAG=0"; do
  env $cfg python bench.py --scene sheet-500k --steps 200 --warmup 40 --no-cpu-baseline --no-kernels 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$cfg', d['value'], d.get('value_draped'), d.get('draped'))"
done 2>&1 | tee -a $O/ab.txt
MPMHIP_W6=1 MPMHIP_G2P_MFLAG=0 timeout 900 python -m pytest tests/test_gpu_ref_golden.py tests/test_gpu_parity.py tests/test_gpu_edges.py -x -q 2>&1 | tail -4 | tee $O/pytest_w6.txt
MPMHIP_LIB=$V/libmpmhip_dbg.so MPMHIP_W6=1 MPMHIP_G2P_MFLAG=0 python tools/gpu/wgtrace.py sheet-500k 100 r03c_w6 > $O/wgtrace_sheet_w6.md 2>/dev/null; sed -n 5,32p $O/wgtrace_sheet_w6.md
