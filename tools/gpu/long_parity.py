#!/usr/bin/env python
"""Headline scene, 1000 substeps: fast back end against the reference-structured baseline kernels (positions /
velocities), plus conservation checks.  (The serial CPU oracle needs ~20 min for this; tests cover it at small sizes.)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from mpmavatar_amd import harness, scenes
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
sc = scenes.REGISTRY["sheet-500k"]()
a = harness.build_solver(sc, "cuda:0", mode="fast")
b = harness.build_solver(sc, "cuda:0", mode="baseline")
rel = lambda x, y: float(np.abs(x - y).max() / max(np.abs(y).max(), 1e-3))
for k in range(0, n, 250):
    harness.run(a, 250, fused=True); harness.run(b, 250, fused=True)
    xa, xb = a.state.particle_x.cpu().numpy(), b.state.particle_x.cpu().numpy()
    va, vb = a.state.particle_v.cpu().numpy(), b.state.particle_v.cpu().numpy()
    st = a.solver.stats()
    print(f"substep {k + 250}: rel dx {rel(xa, xb):.2e}  rel dv {rel(va, vb):.2e}  max|v| {np.abs(vb).max():.3f}  rebins {st['rebins']} fallback {st['n_fallback_particles']}", flush=True)
