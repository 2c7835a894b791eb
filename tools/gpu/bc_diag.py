#!/usr/bin/env python
"""One substep of a sheet under a velocity cuboid in both back ends; dense grid_v_out compared with the oracle node by node."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from mpmavatar_amd import harness, scenes
from oracle.scene_adapter import oracle_from_scene, run_scene

sc = scenes.sheet(n=12, n_grid=40, collider_subdiv=2, span=(0.6, 1.4), y=1.2, sphere_r=0.2, sphere_c=(1.0, 0.97, 1.0), name="diag")
c = sc.x.mean(0)
sc.bcs = list(sc.bcs) + [("velocity_cuboid", {"point": [float(c[0]) + 0.1, float(c[1]), float(c[2])], "size": [0.05, 0.3, 0.3],
                                               "velocity": [-0.8, 0.0, 0.0], "start_time": 0.0, "end_time": 999.0, "reset": 0})]
sc.dt = 1e-4
print("bcs", sc.bcs)
for nsteps in (1, 2):
    o = oracle_from_scene(sc); run_scene(o, sc, nsteps)
    G = sc.n_grid
    og = np.asarray(o.grid_v_out).reshape(G, G, G, 3); om = np.asarray(o.grid_m).reshape(G, G, G)
    for mode in ("baseline", "fast"):
        t = harness.build_solver(sc, "cuda:0", mode=mode)
        harness.run(t, nsteps, fused=False)
        m, vi, vo = t.solver.export_grid()
        m, vo = m.cpu().numpy(), vo.cpu().numpy()
        v = t.state.particle_v.cpu().numpy()
        live = om > 0
        d = np.abs(vo - og).max(-1) * live
        print(f"steps {nsteps} {mode}: particle dv {np.abs(v - o.v).max():.2e}; live nodes {live.sum()}; nodes differing >1e-3: {(d > 1e-3).sum()}; mass diff {np.abs(m - om).max():.2e}")
        for idx in np.argwhere(d > 1e-3)[:12]:
            i, j, k = idx
            print("   node", idx, "pos", idx * np.float32(sc.grid_lim / G), "gpu", vo[i, j, k], "oracle", og[i, j, k], "m", om[i, j, k])
