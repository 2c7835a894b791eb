#!/bin/bash
# round-2 experiment batch: full GPU test-suite, reference-sequence report, g2p packed-math A/B, early/late kernel trace
R=$GRAFT_REPO_ROOT; cd $R; O=$R/gpurun_out; mkdir -p $O
python -m pytest tests -m gpu -q 2>&1 | tail -60 > $O/r02c_pytest.log; tail -15 $O/r02c_pytest.log
python tools/gpu/ref_seq_report.py 2>/dev/null | grep "^|" > $O/r02c_ref_seq_report.md; tail -5 $O/r02c_ref_seq_report.md
V=$R/mpmavatar_amd/lib/variants
for scene in sheet-500k block-512k garment-120k-aniso demo-250; do
  for lib in default g2p_scalar g2p_pk5; do
    L=$R/mpmavatar_amd/lib/libmpmhip.so; [ $lib != default ] && L=$V/libmpmhip_$lib.so
    for rep in 1 2; do
      MPMHIP_LIB=$L python bench.py --scene $scene --steps 400 --warmup 40 --no-cpu-baseline --advance 0 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        o=json.loads(l); print('$scene $lib us/step', round(o['ms_per_step']*1e3,2), {k['name']:round(k['ms']*1e3,1) for k in o.get('kernels',[])})
"
    done
  done
done 2>&1 | tee $O/r02c_ab.log
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $O/r02c_trace -o late -- python $R/bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-kernels --advance 2000 > $O/r02c_trace.log 2>&1
python - <<'PY'
import csv, glob, os, collections, re
R=os.environ['GRAFT_REPO_ROOT']
f=glob.glob(R+'/gpurun_out/r02c_trace/**/late_kernel_trace.csv', recursive=True)[0]
rows=list(csv.DictReader(open(f)))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
def name(r):
    m=re.search(r"\b(k_\w+(?:<[^>]*>)?)\(", r['Kernel_Name']); return m.group(1) if m else r['Kernel_Name'][:40]
ks=[(name(r),(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3) for r in rows]
idx=[i for i,(n,_) in enumerate(ks) if n.startswith('k_p2g')]
def window(lo,hi,label):
    agg=collections.defaultdict(list)
    for n,d in ks[idx[lo]:idx[hi]]: agg[n].append(d)
    print(label, {n:(len(v),round(sum(v)/len(v),1)) for n,v in agg.items() if sum(v)>50})
window(30,200,'early (substeps 30-200)')
window(len(idx)-190,len(idx)-1,'late (last 190 substeps)')
PY
