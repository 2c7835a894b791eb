#!/usr/bin/env python
"""gpurun_out/r04_fetch/pmc_*/**/pmc_counter_collection.csv -> profiles/r04_traffic_calibration.json

factor = known bytes / (counter x 1024): what FETCH_SIZE / WRITE_SIZE (KiB as reported) must be multiplied with for the access
pattern of each kernel of tools/ubench_fetch.hip.  tools/summarize_prof.py applies `read_factor_soa4` / `write_factor_soa4`."""
import collections, csv, glob, json, re

N = 1 << 26
KNOWN = {"k_read_soa4<16>": 64.0 * N, "k_read_soa4<1>": 4.0 * N, "k_read_16": 64.0 * N, "k_read_gather12": 16.0 * N,
         "k_write_soa4<16>": 64.0 * N, "k_atomic4": 4.0 * N}
res = collections.defaultdict(dict)
for ctr, d in (("FETCH_SIZE", "pmc_fetch"), ("WRITE_SIZE", "pmc_write")):
    acc = collections.defaultdict(list)
    for f in glob.glob(f"gpurun_out/r04_fetch/{d}/**/pmc_counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(f)):
            if row["Counter_Name"] != ctr:
                continue
            m = re.search(r"\b(k_\w+(?:<[^>]*>)?)", row["Kernel_Name"])
            if m:
                acc[m.group(1)].append(float(row["Counter_Value"]))
    for k, v in acc.items():
        if k in KNOWN:
            kib = sum(v) / len(v)
            res[k][ctr + "_KiB"] = kib
            res[k]["known_bytes"] = KNOWN[k]
            res[k][ctr + "_factor"] = KNOWN[k] / (kib * 1024) if kib > 0 else None
out = {"source": "tools/ubench_fetch.hip under rocprofv3 --kernel-trace --pmc FETCH_SIZE | WRITE_SIZE (tools/gpu/r04_fetch_calib.sh); "
                 "buffers of 256 MiB per component (> Infinity Cache), every byte touched once",
       "kernels": res,
       "read_factor_soa4": res.get("k_read_soa4<16>", {}).get("FETCH_SIZE_factor"),
       "read_factor_16B": res.get("k_read_16", {}).get("FETCH_SIZE_factor"),
       "write_factor_soa4": res.get("k_write_soa4<16>", {}).get("WRITE_SIZE_factor")}
json.dump(out, open("profiles/r04_traffic_calibration.json", "w"), indent=1)
print(json.dumps(out, indent=1))
