#!/bin/bash
# PMC passes (separate runs, kernel-trace only) for the fast path; summaries printed per kernel
R=$GRAFT_REPO_ROOT; cd /tmp; export TMPDIR=/tmp
mkdir -p $R/gpurun_out/pmc
i=0
for set in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU" \
           "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
           "FETCH_SIZE" "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum" "GRBM_GUI_ACTIVE GRBM_COUNT"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $set -d $R/gpurun_out/pmc/p$i -o pmc --output-format csv -- python $R/bench.py --steps 40 --warmup 8 --no-cpu-baseline --no-kernels > $R/gpurun_out/pmc/p$i.log 2>&1
done
python - <<'PY'
import csv, glob, collections, os
R=os.environ['GRAFT_REPO_ROOT']
agg=collections.defaultdict(lambda: collections.defaultdict(list))
for f in sorted(glob.glob(R+'/gpurun_out/pmc/p*/**/*counter_collection.csv', recursive=True)):
    for row in csv.DictReader(open(f)):
        k=row['Kernel_Name'].split('(')[0].replace('mpm::(anonymous namespace)::','')
        if not k.startswith('k_') : continue
        agg[k][row['Counter_Name']].append(float(row['Counter_Value']))
for k,v in agg.items():
    print(k, {c: round(sum(x)/len(x),1) for c,x in v.items()})
PY
