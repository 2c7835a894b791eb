#!/bin/bash
# round 3, batch b: workgroup timelines (debug build), contraction-free variant against the reference sequences, per-particle columns
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03b; mkdir -p $O; cd $R
V=$R/mpmavatar_amd/lib/variants
MPMHIP_LIB=$V/libmpmhip_dbg.so python tools/gpu/wgtrace.py sheet-500k 100 r03b > $O/wgtrace_sheet_early.md 2> $O/wg1.err; head -60 $O/wgtrace_sheet_early.md
MPMHIP_LIB=$V/libmpmhip_dbg.so python tools/gpu/wgtrace.py sheet-500k 2200 r03b_late > $O/wgtrace_sheet_late.md 2> $O/wg2.err
MPMHIP_LIB=$V/libmpmhip_dbg.so python tools/gpu/wgtrace.py garment-120k-aniso 100 r03b > $O/wgtrace_garment.md 2> $O/wg3.err
MPMHIP_LIB=$V/libmpmhip_dbg.so python tools/gpu/wgtrace.py block-512k 100 r03b > $O/wgtrace_block.md 2> $O/wg4.err
MPMHIP_LIB=$V/libmpmhip_dbg.so python tools/gpu/wgtrace.py cube-8k 100 r03b > $O/wgtrace_cube.md 2> $O/wg5.err
python tools/gpu/ref_seq_report.py > $O/ref_seq_report.md 2> /dev/null
MPMHIP_LIB=$V/libmpmhip_nofma.so python tools/gpu/ref_seq_report.py > $O/ref_seq_report_nofma.md 2> /dev/null
grep "jelly\|gamma0" $O/ref_seq_report.md; echo; grep "jelly\|gamma0" $O/ref_seq_report_nofma.md
for sc in sheet-500k garment-120k-aniso cube-8k; do
  echo "== $sc default / nofma"; python bench.py --scene $sc --steps 400 --warmup 40 --no-cpu-baseline --no-kernels --advance 0 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"
  MPMHIP_LIB=$V/libmpmhip_nofma.so python bench.py --scene $sc --steps 400 --warmup 40 --no-cpu-baseline --no-kernels --advance 0 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"
done
timeout 600 python -m pytest tests/test_gpu_api.py -q -k "cov" 2>&1 | tail -3
