#!/usr/bin/env python
"""Substep rate of the headline scene over a long run, in windows (the sheet starts flat and drapes over the sphere)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from mpmavatar_amd import harness, scenes
n, win = (int(sys.argv[1]) if len(sys.argv) > 1 else 3000), 250
sim = harness.build_solver(scenes.REGISTRY[sys.argv[2] if len(sys.argv) > 2 else "sheet-500k"](), "cuda:0")
harness.run(sim, 20, fused=True); torch.cuda.synchronize()
last = sim.solver.stats()
for k in range(0, n, win):
    t0 = time.perf_counter(); harness.run(sim, win, fused=True); torch.cuda.synchronize(); dt = time.perf_counter() - t0
    st = sim.solver.stats()
    print(f"substeps {k}-{k + win}: {1e6 * dt / win:.1f} us/substep  rebins +{st['rebins'] - last['rebins']}  fallback +{st['n_fallback_particles'] - last['n_fallback_particles']}  active blocks {st['n_active_blocks']}", flush=True)
    last = st
