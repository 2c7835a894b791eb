#!/usr/bin/env python
"""Condense the rocprofv3 outputs of tools/gpu/profile_scene.sh into small tracked summaries under profiles/.

usage: python tools/summarize_prof.py <tag> [scene ...]      (e.g. r02a sheet-500k garment-120k-aniso)
writes profiles/<tag>_<scene>_rocprof_summary.md, profiles/<tag>_<scene>_bench.json and, for the headline scene,
profiles/<tag>_pmc.json (per-launch HBM traffic per kernel, read by bench.py's roofline.traffic).
"""
import collections, csv, glob, json, os, re, sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r02"
# counter -> bytes factors measured on known byte counts in this code's access pattern (tools/ubench_fetch.hip,
# profiles/r04_traffic_calibration.json); without the file: the guide's x2 for reads, x1 for writes
try:
    _cal = json.load(open("profiles/r04_traffic_calibration.json"))
    RF, WF = float(_cal["read_factor_soa4"]), float(_cal["write_factor_soa4"])
except Exception:  # noqa: BLE001
    RF, WF = 2.0, 1.0
scenes = sys.argv[2:] or ["sheet-500k"]
os.makedirs("profiles", exist_ok=True)


def kname(full):
    m = re.search(r"\b(k_\w+(?:<[^>]*>)?)\(", full)
    return m.group(1) if m else re.sub(r"[<(].*", "", full)[:48]


for scene in scenes:
    src = f"gpurun_out/prof_{tag}_{scene}"
    out = []
    st = glob.glob(f"{src}/**/fast_kernel_stats.csv", recursive=True)
    rows = list(csv.DictReader(open(st[0]))) if st else []
    out.append(f"# rocprofv3 --kernel-trace --stats -- python bench.py --scene {scene} --steps 200 --warmup 40 --no-kernels --advance 0   [{tag}]\n")
    out.append("| kernel | calls | avg us | min us | max us | % |\n|---|---|---|---|---|---|")
    for r in rows[:14]:
        out.append(f"| {kname(r['Name'])} | {r['Calls']} | {float(r['AverageNs'])/1e3:.2f} | {float(r['MinNs'])/1e3:.2f} | {float(r['MaxNs'])/1e3:.2f} | {float(r['Percentage']):.2f} |")
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in sorted(glob.glob(f"{src}/pmc_*/**/pmc_counter_collection.csv", recursive=True)):
        for row in csv.DictReader(open(f)):
            k = kname(row["Kernel_Name"])
            if k.startswith("k_"):
                agg[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
    hot = [kname(r["Name"]) for r in rows[:6]]
    if agg:
        out.append("\n# PMC passes (each its own run: rocprofv3 --kernel-trace --pmc <set>), averages per launch\n")
        out.append("FETCH_SIZE / WRITE_SIZE are KiB as reported, multiplied with the factors measured on known byte counts in this code's access "
                   f"pattern (profiles/r04_traffic_calibration.json: reads x {RF:.3f}, writes x {WF:.3f}).  SQ_* cycle counters are quad-cycles summed over waves.\n")
        out.append("| kernel | hbm_read_MB (2x) | hbm_write_MB | L2 hit % | waves | VALU / wave | LDS / wave | VMEM rd+wr / wave | VALU busy % of wave time | wait-any % | LDS bank conflict % |")
        out.append("|---|---|---|---|---|---|---|---|---|---|---|")
        for k, v in agg.items():
            a = {c: sum(x) / len(x) for c, x in v.items()}
            if "FETCH_SIZE" not in a or (hot and k not in hot and a.get("SQ_WAVES", 0) < 500):
                continue
            w = max(a.get("SQ_WAVES", 0), 1)
            wc = max(a.get("SQ_WAVE_CYCLES", 0), 1)
            hit, miss = a.get("TCC_HIT_sum", 0), a.get("TCC_MISS_sum", 0)
            out.append(f"| {k} | {RF*a.get('FETCH_SIZE',0)/1024:.1f} | {WF*a.get('WRITE_SIZE',0)/1024:.1f} | {100*hit/max(hit+miss,1):.0f} | {w:.0f} | "
                       f"{a.get('SQ_INSTS_VALU',0)/w:.0f} | {a.get('SQ_INSTS_LDS',0)/w:.0f} | {(a.get('SQ_INSTS_VMEM_RD',0)+a.get('SQ_INSTS_VMEM_WR',0))/w:.0f} | "
                       f"{100*a.get('SQ_ACTIVE_INST_VALU',0)/wc:.0f} | {100*a.get('SQ_WAIT_ANY',0)/wc:.0f} | {100*a.get('SQ_LDS_BANK_CONFLICT',0)/max(a.get('SQ_ACTIVE_INST_LDS',1),1):.0f} |")
        if scene == "sheet-500k":
            pmc = {k: {c: sum(x) / len(x) for c, x in v.items()} for k, v in agg.items() if "FETCH_SIZE" in v}
            json.dump({"tag": tag, "source": "rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE (separate passes), sheet-500k fast mode",
                       "kernels": {k: {"hbm_read_bytes": RF * a.get("FETCH_SIZE", 0) * 1024, "hbm_write_bytes": WF * a.get("WRITE_SIZE", 0) * 1024,
                                       "read_factor": RF, "write_factor": WF,
                                       "launches_sampled": len(agg[k]["FETCH_SIZE"])} for k, a in pmc.items()}},
                      open(f"profiles/{tag}_pmc.json", "w"), indent=1)
    p = f"{src}/bench_fast.json"
    if os.path.exists(p):
        line = [l for l in open(p) if l.startswith("{")]
        if line:
            o = json.loads(line[-1])
            json.dump(o, open(f"profiles/{tag}_{scene}_bench.json", "w"), indent=1)
            out.append(f"\n# bench.py --scene {scene} --steps 400 --warmup 40\n")
            out.append(f"value {o['value']:.0f} substeps/s ({o['ms_per_step']*1e3:.1f} us/substep), whole-substep fraction of 8 TB/s (algorithmic bytes) "
                       f"{o.get('substep_frac_of_hbm_peak', float('nan')):.3f}; after {o.get('draped', {}).get('advance', 0)} more substeps: "
                       f"{o.get('value_draped', float('nan')):.0f} substeps/s, {o.get('draped', {}).get('rebins_in_window')} re-sorts in the window\n")
            out.append("| launch of the timed loop (HIP events) | us | algorithmic MB | frac of 8 TB/s |\n|---|---|---|---|")
            for k in o.get("kernels", []):
                out.append(f"| {k['name']} | {k['ms']*1e3:.1f} | {k.get('alg_bytes', 0)/1e6:.1f} | {k.get('frac', float('nan')):.3f} |")
    open(f"profiles/{tag}_{scene}_rocprof_summary.md", "w").write("\n".join(out) + "\n")
    print(open(f"profiles/{tag}_{scene}_rocprof_summary.md").read())
