#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04c; mkdir -p $O; cd $R
export MPMHIP_LIB=$R/mpmavatar_amd/lib/variants/libmpmhip_dbg.so
for pair in 1 0; do
  MPMHIP_PAIR=$pair python tools/gpu/wgtrace.py sheet-500k 100 r04c_pair$pair > $O/wg_sheet_pair$pair.txt 2>&1
  MPMHIP_PAIR=$pair python tools/gpu/wgtrace.py block-512k 100 r04c_pair$pair > $O/wg_block_pair$pair.txt 2>&1
done
head -40 $O/wg_sheet_pair1.txt
