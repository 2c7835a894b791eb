#!/usr/bin/env python
"""Where does the time of the fused loop go early and late in the headline run?  Kernel ablations (MPMHIP_DBG bits, set at
run time; the ablated substeps compute garbage, so every measurement restarts from a saved state)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from mpmavatar_amd import harness, scenes

def measure(sim, flags, n=120):
    sv = sim.solver
    sv._call("mpmhip_set_debug_flags", flags)
    sv.enable_profiling(True, fused=True); sv.time_profile.clear()
    harness.run(sim, n, fused=True)
    sv.enable_profiling(False)
    sv._call("mpmhip_set_debug_flags", 0)
    return {k: round(1e3 * sum(v) / len(v), 1) for k, v in sv.time_profile.items() if k in ("compute_stress_from_F_trial", "p2g", "g2p_v")}

for upto in (100,):
    print(f"== state after {upto} substeps", flush=True)
    for flags, what in ((0, "full"), (256, "no splat workgroups"), (128, "no LDS atomics"), (2, "no scatter arithmetic"), (1, "no flush"),
                        (2 | 256 | 1, "loads + barriers only"), (8, "no vertex-force gather"), (16, "no stress loads"), (2048, "no clearing workgroups"),
                        (2 | 256 | 1 | 2048 | 8 | 16, "record + x, m, C, v loads + barriers")):
        sim = harness.build_solver(scenes.REGISTRY["sheet-500k"](), "cuda:0")
        harness.run(sim, upto, fused=True)
        st = sim.solver.stats()
        print(f"{what:28s} {measure(sim, flags)}  chunks? active blocks {st['n_active_blocks']} collider nodes {st['n_collider_nodes']}", flush=True)
        del sim
        torch.cuda.empty_cache()
