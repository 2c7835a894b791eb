#!/usr/bin/env python
"""Condense rocprofv3 outputs under gpurun_out/prof into small tracked summaries under profiles/.

usage: python tools/summarize_prof.py <tag>      (e.g. r01b)
"""
import csv, glob, json, os, re, sys, collections

tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
src = "gpurun_out/prof"
os.makedirs("profiles", exist_ok=True)
out = []
rows = list(csv.DictReader(open(f"{src}/fast_kernel_stats.csv")))
out.append(f"# rocprofv3 --kernel-trace --stats  (bench.py --steps 200 --warmup 40, sheet-500k, fast mode) [{tag}]\n")
out.append("| kernel | calls | avg us | min us | max us | % |\n|---|---|---|---|---|---|")
for r in rows[:16]:
    m = re.search(r"\b(k_\w+(?:<[^>]*>)?)\(", r["Name"])
    name = m.group(1) if m else re.sub(r"[<(].*", "", r["Name"])
    out.append(f"| {name[:60]} | {r['Calls']} | {float(r['AverageNs'])/1e3:.2f} | {float(r['MinNs'])/1e3:.2f} | {float(r['MaxNs'])/1e3:.2f} | {float(r['Percentage']):.2f} |")
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in sorted(glob.glob(f"{src}/pmc_*/pmc_counter_collection.csv")):
    for row in csv.DictReader(open(f)):
        m = re.search(r"\b(k_\w+(?:<[^>]*>)?)\(", row["Kernel_Name"])
        if m:
            agg[m.group(1)][row["Counter_Name"]].append(float(row["Counter_Value"]))
if agg:
    out.append("\n# PMC passes (separate runs, --kernel-trace --pmc ...), averages per launch\n")
    out.append("FETCH_SIZE / WRITE_SIZE are in KiB as reported; on gfx950 FETCH_SIZE under-reports wide coalesced reads by 2x "
               "(MI355X_MICROARCH.md, HBM section), so `hbm_read_MB` below is 2 x FETCH_SIZE.\n")
    out.append("| kernel | FETCH_SIZE KiB | WRITE_SIZE KiB | hbm_read_MB (2x) | hbm_write_MB | L2 hit % | VALU insts | LDS insts | wave-cycles | wait-any |")
    out.append("|---|---|---|---|---|---|---|---|---|---|")
    for k, v in agg.items():
        a = {c: sum(x) / len(x) for c, x in v.items()}
        if "FETCH_SIZE" not in a:
            continue
        hit = a.get("TCC_HIT_sum", 0); miss = a.get("TCC_MISS_sum", 0)
        out.append(f"| {k} | {a.get('FETCH_SIZE',0):.0f} | {a.get('WRITE_SIZE',0):.0f} | {2*a.get('FETCH_SIZE',0)/1024:.1f} | {a.get('WRITE_SIZE',0)/1024:.1f} | "
                   f"{100*hit/max(hit+miss,1):.0f} | {a.get('SQ_INSTS_VALU',0):.0f} | {a.get('SQ_INSTS_LDS',0):.0f} | {a.get('SQ_WAVE_CYCLES',0):.0f} | {a.get('SQ_WAIT_ANY',0):.0f} |")
pmc = {k: {c: sum(x) / len(x) for c, x in v.items()} for k, v in agg.items() if "FETCH_SIZE" in v}
if pmc:
    # machine-readable per-launch HBM traffic for bench.py's roofline.traffic (bytes; reads = 2 x FETCH_SIZE KiB, see above)
    json.dump({"tag": tag, "source": "rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE (separate passes), sheet-500k fast mode",
               "kernels": {k: {"hbm_read_bytes": 2 * a.get("FETCH_SIZE", 0) * 1024, "hbm_write_bytes": a.get("WRITE_SIZE", 0) * 1024,
                               "launches_sampled": len(agg[k]["FETCH_SIZE"])} for k, a in pmc.items()}},
              open(f"profiles/{tag}_pmc.json", "w"), indent=1)
open(f"profiles/{tag}_rocprof_summary.md", "w").write("\n".join(out) + "\n")
for name in ("bench_fast.json", "bench_baseline.json"):
    p = f"gpurun_out/{name}"
    if os.path.exists(p):
        line = [l for l in open(p) if l.startswith("{")]
        if line:
            open(f"profiles/{tag}_{name}", "w").write(json.dumps(json.loads(line[-1]), indent=1) + "\n")
print(open(f"profiles/{tag}_rocprof_summary.md").read())
