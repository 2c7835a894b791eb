#!/bin/bash
# r04t: what limits the four concurrent simulations of the finite-difference step?  (a) hardware queues (GPU_MAX_HW_QUEUES),
# (b) kernel trace of the concurrent leg: per-queue gaps between consecutive kernels and the overlap between queues.
R=$GRAFT_REPO_ROOT; cd /tmp; export TMPDIR=/tmp
for q in ${QUEUES:-default 8 2}; do
  if [ $q = default ]; then unset GPU_MAX_HW_QUEUES; else export GPU_MAX_HW_QUEUES=$q; fi
  echo "== GPU_MAX_HW_QUEUES=$q"; python $R/tools/gpu/fd_bench.py 200 200 128 2 400 2>&1 | grep -E "^(sequential|concurrent)" | cut -c1-200
done
if [ -n "$TRACE_QUEUES" ]; then export GPU_MAX_HW_QUEUES=$TRACE_QUEUES; else unset GPU_MAX_HW_QUEUES; fi
O=$R/gpurun_out/r04t; mkdir -p $O
timeout 600 rocprofv3 --kernel-trace -d $O/trace -o fd --output-format csv -- python $R/tools/gpu/fd_bench.py 200 200 128 1 200 > $O/trace.log 2>&1
python - "$O" <<'PY'
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + '/trace/**/fd_kernel_trace.csv', recursive=True)[0]
rows = [r for r in csv.DictReader(open(f))]
hot = [r for r in rows if any(k in r['Kernel_Name'] for k in ('k_p2g', 'k_g2p', 'k_stress_elem'))]
byq = collections.defaultdict(list)
for r in hot: byq[r['Queue_Id']].append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'][:0] + ('p2g' if 'k_p2g' in r['Kernel_Name'] else 'g2p' if 'k_g2p' in r['Kernel_Name'] else 'stress')))
print('queues with hot kernels:', {q: len(v) for q, v in byq.items()})
# the concurrent leg = the time span in which more than one queue is active; take the last 60 % of the trace of queues other than the busiest-first
for q, v in sorted(byq.items()):
    v.sort()
    tail = v[len(v) // 2:]
    dur = collections.defaultdict(list); gaps = []
    for i, (s, e, n) in enumerate(tail):
        dur[n].append((e - s) / 1e3)
        if i: gaps.append((s - tail[i - 1][1]) / 1e3)
    med = lambda x: sorted(x)[len(x) // 2] if x else float('nan')
    print(f'queue {q}: second half of its launches: median duration us ' + ', '.join(f'{n} {med(d):.1f}' for n, d in dur.items()) + f'; median gap to the previous kernel of the queue {med(gaps):.1f} us (p90 {sorted(gaps)[int(0.9 * len(gaps))]:.1f})')
# overlap: sample the last third of the trace
allk = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp'])) for r in hot)
t0 = allk[2 * len(allk) // 3][0]; t1 = allk[-1][1]
ev = []
for s, e in allk:
    if e > t0: ev += [(max(s, t0), 1), (e, -1)]
ev.sort(); cur = 0; last = t0; hist = collections.Counter()
for t, d in ev:
    hist[cur] += t - last; last = t; cur += d
tot = sum(hist.values())
print('kernels in flight over the last third of the trace:', {k: f'{100 * v / tot:.0f}%' for k, v in sorted(hist.items())})
PY
