#!/usr/bin/env python
"""Per-phase times of the headline scene early and late in a long run (profiling mode: one launch per phase)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from mpmavatar_amd import harness, scenes
sim = harness.build_solver(scenes.REGISTRY["sheet-500k"](), "cuda:0")
done = 0
for upto in (200, 1500, 2800):
    harness.run(sim, upto - done, fused=True); done = upto
    sim.solver.enable_profiling(True); sim.solver.time_profile.clear()
    harness.run(sim, 40, fused=False); done += 40
    sim.solver.enable_profiling(False)
    st = sim.solver.stats()
    tp = {k: round(1e3 * sum(v) / len(v), 1) for k, v in sim.solver.time_profile.items() if k != "rebin"}
    print(f"after {upto}: {tp} collider nodes {st['n_collider_nodes']} active blocks {st['n_active_blocks']}", flush=True)
