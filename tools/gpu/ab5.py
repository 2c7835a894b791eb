#!/usr/bin/env python
"""Alternating A/B of library variants through bench.py (round 5).   python tools/gpu/ab5.py --libs default,nod3 --scenes sheet-500k [--reps 2]
    [--advance 2000] [--steps 200] [--env "A=1,B=2"]
A variant name is a file lib/variants/libmpmhip_<name>.so (tools/build_variants.py) or `default`; `name@K=V;K2=V2` adds environment variables.
Prints per run: substeps/s, draped substeps/s, kernel-stamp us of the three hot launches."""
import argparse, json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ap = argparse.ArgumentParser()
ap.add_argument("--libs", default="default"); ap.add_argument("--scenes", default="sheet-500k"); ap.add_argument("--reps", type=int, default=2)
ap.add_argument("--advance", type=int, default=0); ap.add_argument("--steps", type=int, default=200); ap.add_argument("--pre-advance", type=int, default=0)
ap.add_argument("--out", default=None)
a = ap.parse_args()
rows = []
for sc in a.scenes.split(","):
    for rep in range(a.reps):
        for spec in a.libs.split(","):
            name, _, envs = spec.partition("@")
            env = dict(os.environ)
            if name != "default":
                env["MPMHIP_LIB"] = os.path.join(ROOT, "mpmavatar_amd", "lib", "variants", f"libmpmhip_{name}.so")
            for kv in filter(None, envs.split(";")):
                k, _, v = kv.partition("="); env[k] = v
            cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--scene", sc, "--steps", str(a.steps), "--warmup", "40", "--no-cpu-baseline",
                   "--advance", str(a.advance), "--pre-advance", str(a.pre_advance)]
            r = subprocess.run(cmd, env=env, capture_output=True, text=True, cwd=ROOT)
            line = [l for l in r.stdout.splitlines() if l.startswith("{")]
            if not line:
                print(sc, spec, "FAILED", r.stderr[-500:]); continue
            o = json.loads(line[-1])
            k = {x["phase"]: round(1e3 * x["ms"], 2) for x in o.get("kernels", [])}
            row = dict(scene=sc, lib=spec, rep=rep, value=round(o["value"]), draped=round(o.get("value_draped", 0)), kernels_us=k)
            rows.append(row); print(json.dumps(row), flush=True)
if a.out:
    json.dump(rows, open(a.out, "w"), indent=1)
