#!/bin/bash
# re-sort after the launch diet (k_keys = keys + first histogram + flag clearing + grid clearing; two launches per radix pass; last
# pass flags the blocks; count + compact instead of scan + total + compact; one topology kernel): timing, trace, the whole GPU suite
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03u
rm -f gpurun_out/r03u/tests.txt gpurun_out/r03u/bench.txt
timeout 900 python -m pytest tests/test_gpu_sort.py -x -q 2>&1 | tail -4 | tee -a gpurun_out/r03u/tests.txt
for rep in 1 2; do
for sort in radix rocprim; do
  for scene in sheet-500k garment-120k-aniso block-512k demo-250 cube-8k; do
    MPMHIP_SORT=$sort timeout 600 python bench.py --scene $scene --steps 400 --warmup 40 --no-cpu-baseline --pre-advance 2000 --advance 0 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$sort $scene', round(d['value']), [(k['name'], round(k['ms']*1e3,1)) for k in d['kernels']])" | tee -a gpurun_out/r03u/bench.txt
  done
done
done
cd /tmp && export TMPDIR=/tmp
rm -rf $GRAFT_REPO_ROOT/gpurun_out/r03u/trace
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r03u/trace -o t -- python $GRAFT_REPO_ROOT/bench.py --scene sheet-500k --steps 300 --warmup 40 --no-cpu-baseline --no-kernels --pre-advance 2000 --advance 0 > $GRAFT_REPO_ROOT/gpurun_out/r03u/trace.log 2>&1
cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -5 | tee -a gpurun_out/r03u/tests.txt
