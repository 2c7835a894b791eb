#!/bin/bash
# round 3, batch o: occupancy of the stress kernel (a single partial round of wavefronts that load, compute and store in lock step)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03o; mkdir -p $O; cd $R
V=$R/mpmavatar_amd/lib/variants
for v in default s2 s3 s4 s5 s6; do
  L=$V/libmpmhip_$v.so; [ $v = default ] && L=$R/mpmavatar_amd/lib/libmpmhip.so
  for sc in sheet-500k garment-120k-aniso; do
  MPMHIP_LIB=$L python bench.py --scene $sc --steps 400 --warmup 40 --no-cpu-baseline --advance 0 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); k={x['phase']:x['ms']*1e3 for x in d['kernels']}
print('%-8s %-20s %7.0f /s %6.2f us | stress %5.2f p2g %5.2f g2p %5.2f' % ('$v','$sc',d['value'],d['ms_per_step']*1e3,k.get('compute_stress_from_F_trial',0),k.get('p2g',0),k.get('g2p_v',0)))"
  done
done | tee $O/ab.txt
