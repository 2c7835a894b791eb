#!/bin/bash
# r04s: where do k_p2g / k_g2p / k_stress_elem wait?  Memory-path counters (TCP = vector L1, TCC = L2, TA/TD = texture address / data,
# UTCL1 = L1 TLB) in separate rocprofv3 passes (kernel-trace only), headline scene.  A pass whose counter set the device rejects is skipped.
# (The TA_* / TD_* sets hang rocprofv3 on this image until the timeout: not in the list.)
R=$GRAFT_REPO_ROOT; cd /tmp; export TMPDIR=/tmp
SC=${1:-sheet-500k}
O=$R/gpurun_out/r04s_$SC; mkdir -p $O
i=0
while read -r set; do
  [ -z "$set" ] && continue
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $set -d $O/pmc_$i -o pmc --output-format csv -- python $R/bench.py --scene $SC --steps 40 --warmup 8 --no-cpu-baseline --no-kernels --advance 0 > $O/pmc_$i.log 2>&1 || echo "pass $i ($set) failed: $(tail -1 $O/pmc_$i.log | cut -c1-200)"
done <<'SETS'
SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_INSTS_SMEM
SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_INSTS_LDS_ATOMIC SQ_INSTS_LDS
SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_DATA_FIFO_FULL
TCP_TOTAL_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_TCC_ATOMIC_WITHOUT_RET_REQ_sum
TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_TA_TCP_STATE_READ_sum TCP_GATE_EN1_sum
TCP_TCC_READ_REQ_LATENCY_sum TCP_TCP_LATENCY_sum TCP_TCC_WRITE_REQ_LATENCY_sum TCP_TOTAL_READ_sum
TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_REQUEST_sum TCP_UTCL1_STALL_INFLIGHT_MAX_sum
TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_ATOMIC_sum
TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_DRAM_sum TCC_EA0_WRREQ_sum TCC_TAG_STALL_sum
TCC_BUSY_sum TCC_CYCLE_sum TCC_EA0_RDREQ_LEVEL_sum TCC_EA0_ATOMIC_LEVEL_sum
GRBM_GUI_ACTIVE
SETS
python - "$O" "$SC" <<'PY'
import csv, glob, collections, sys, json
O, SC = sys.argv[1], sys.argv[2]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in sorted(glob.glob(O + '/pmc_*/**/*counter_collection.csv', recursive=True)):
    for row in csv.DictReader(open(f)):
        k = row['Kernel_Name'].split('(')[0].replace('mpm::(anonymous namespace)::', '').replace('void ', '')
        i = row['Kernel_Name'].find('k_'); k = row['Kernel_Name'][i:].split('<')[0].split('(')[0] if i >= 0 else ''
        if k not in ('k_p2g', 'k_g2p', 'k_stress_elem', 'k_stress_elem_splat'): continue
        agg[k][row['Counter_Name']].append(float(row['Counter_Value']))
out = {k: {c: sum(x) / len(x) for c, x in v.items()} for k, v in agg.items()}
json.dump({"scene": SC, "source": "tools/gpu/r04s.sh: rocprofv3 --kernel-trace --pmc <set>, one pass per set, averages per launch", "kernels": out},
          open(O + '/counters.json', 'w'), indent=1)
for k, v in out.items():
    print('==', k)
    for c, x in sorted(v.items()): print(f'   {c:42s} {x:16.1f}')
PY
