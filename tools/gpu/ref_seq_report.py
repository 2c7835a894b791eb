"""HIP (both back ends) vs the reference-run sequence fixtures, every checkpoint: the table of DESIGN.md section 2.
    python tools/gpu/ref_seq_report.py > gpurun_out/ref_seq_report.md"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import refgolden as rg
from mpmavatar_amd import harness

print("| fixture | substep | fast: x | fast: v | baseline: x | baseline: v | fast vs reference(fp32 builtins): v | reference envelope(s): v | v max | per-particle v, floor 1e-3 m/s: fast / baseline | per-particle v, floor 1e-3 vmax: fast / baseline |")
print("|---|---|---|---|---|---|---|---|---|---|---|")
for name in rg.names("seq"):
    z = rg.load(name)
    sims = {}
    for mode in ("fast", "baseline"):
        sc = rg.scene_from_npz(z)
        sims[mode] = harness.build_solver(sc, "cuda:0", mode=mode)
    for cp in z["checkpoints"]:
        row = []
        pp, pps = [], []
        valt = float("nan")
        for mode in ("fast", "baseline"):
            sim = sims[mode]
            harness.run(sim, int(cp) - sim.steps_done)
            x, v = sim.state.particle_x.cpu().numpy(), sim.state.particle_v.cpu().numpy()
            row += [rg.rel(x, z[f"s{cp}_particle_x"]), rg.rel(v, z[f"s{cp}_particle_v"])]
            pp.append(rg.rel_pp(v, z[f"s{cp}_particle_v"])); pps.append(rg.rel_pp_scaled(v, z[f"s{cp}_particle_v"]))
            if mode == "fast" and f"alt_s{cp}_particle_v" in z.files:
                valt = rg.rel(v, z[f"alt_s{cp}_particle_v"])
        envs = [rg.rel(z[k], z[f"s{cp}_particle_v"]) for k in sorted(z.files) if k.endswith(f"_s{cp}_particle_v") and k.startswith("alt")]
        print(f"| {name} | {cp} | " + " | ".join(f"{e:.1e}" for e in row) + f" | {valt:.1e} | " + ", ".join(f"{e:.1e}" for e in envs) +
              f" | {np.abs(z[f's{cp}_particle_v']).max():.2e} | {pp[0]:.1e} / {pp[1]:.1e} | {pps[0]:.1e} / {pps[1]:.1e} |", flush=True)
