#!/bin/bash
# the radix sort of the re-sort: unit tests, then re-sort time (bench's event bracket) with it (tile = 4 / 8 / 16 slices of 256
# pairs) and with the library's sort, and a kernel trace of the draped state for the launch-by-launch picture
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03t
V=$GRAFT_REPO_ROOT/mpmavatar_amd/lib/variants
rm -f gpurun_out/r03t/tests.txt gpurun_out/r03t/bench.txt
for lib in default $V/libmpmhip_rs4.so $V/libmpmhip_rs16.so; do
  [ $lib = default ] && unset MPMHIP_LIB || export MPMHIP_LIB=$lib
  timeout 900 python -m pytest tests/test_gpu_sort.py -x -q 2>&1 | tail -4 | tee -a gpurun_out/r03t/tests.txt
done
for cfg in radix8:default radix4:$V/libmpmhip_rs4.so radix16:$V/libmpmhip_rs16.so rocprim:default; do
  sort=${cfg%%:*}; lib=${cfg#*:}
  [ $lib = default ] && unset MPMHIP_LIB || export MPMHIP_LIB=$lib
  for scene in sheet-500k garment-120k-aniso block-512k; do
    MPMHIP_SORT=$sort timeout 600 python bench.py --scene $scene --steps 400 --warmup 40 --no-cpu-baseline --pre-advance 2000 --advance 0 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$sort $scene', round(d['value']), [(k['name'], round(k['ms']*1e3,1)) for k in d['kernels']])" | tee -a gpurun_out/r03t/bench.txt
  done
done
unset MPMHIP_LIB
cd /tmp && export TMPDIR=/tmp
rm -rf $GRAFT_REPO_ROOT/gpurun_out/r03t/trace
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r03t/trace -o t -- python $GRAFT_REPO_ROOT/bench.py --scene sheet-500k --steps 300 --warmup 40 --no-cpu-baseline --no-kernels --pre-advance 2000 --advance 0 > $GRAFT_REPO_ROOT/gpurun_out/r03t/trace.log 2>&1
cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests/test_gpu_edges.py tests/test_gpu_fuzz.py tests/test_gpu_parity.py tests/test_gpu_api.py -x -q 2>&1 | tail -5 | tee -a gpurun_out/r03t/tests.txt
