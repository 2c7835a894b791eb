#!/usr/bin/env python
"""Per-substep kernel timeline from a rocprofv3 --kernel-trace CSV: start/end of each kernel relative to the substep's
first kernel, and the idle gaps between consecutive kernels.  usage: python tools/timeline.py <kernel_trace.csv> [n_steps]"""
import csv, re, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
nshow = int(sys.argv[2]) if len(sys.argv) > 2 else 2
ev = []
for r in rows:
    m = re.search(r"\b(k_\w+(?:<[^>]*>)?)\(", r["Kernel_Name"])
    ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), m.group(1) if m else r["Kernel_Name"][:40], r["Stream_Id"]))
ev.sort()
first = [i for i, e in enumerate(ev) if e[2].startswith("k_stress_elem") or e[2].startswith("k_stress_trad")]
i0 = first[len(first) // 2]
t0 = ev[i0][0]
steps = 0
prev_end = None
for e in ev[i0:]:
    if e[2].startswith("k_stress") and prev_end is not None:
        steps += 1
        if steps > nshow: break
    gap = (e[0] - prev_end) / 1e3 if prev_end is not None else 0.0
    print(f"{(e[0]-t0)/1e3:8.1f} -> {(e[1]-t0)/1e3:8.1f}  dur {(e[1]-e[0])/1e3:6.1f}  gap {gap:5.1f}  {e[2]} (stream {e[3]})")
    prev_end = max(prev_end or 0, e[1])
# average gaps / durations over the steady half
agg = collections.defaultdict(list); gaps = collections.defaultdict(list)
prev = None
for e in ev[i0:]:
    agg[e[2]].append((e[1] - e[0]) / 1e3)
    if prev is not None: gaps[f"{prev[2]} -> {e[2]}"].append((e[0] - prev[1]) / 1e3)
    prev = e
print("\navg duration (us):", {k: round(sum(v) / len(v), 1) for k, v in agg.items() if len(v) > 20})
print("avg gap (us):", {k: round(sum(v) / len(v), 1) for k, v in gaps.items() if len(v) > 20})
