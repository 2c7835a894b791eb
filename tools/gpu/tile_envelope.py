#!/usr/bin/env python
"""Is the distance of the fixed-point-tile run from the oracle inside the scene's own sensitivity?  One scene, N substeps, four runs in
lock step: the OpenMP oracle (16 threads), the same oracle with another thread count (another summation order of its atomic adds --
the scene's chaos floor), the fast back end with the packed fixed-point tile (default) and with the fp64 tile (MPMHIP_P2G_TILE=f64).
    python tools/gpu/tile_envelope.py <scene> [n_substeps]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from mpmavatar_amd import harness, scenes
from oracle.scene_adapter import omp_threads, oracle_from_scene, run_scene

name = sys.argv[1] if len(sys.argv) > 1 else "demo-250"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
sc = scenes.REGISTRY[name]()
th = omp_threads()
oa, ob = oracle_from_scene(sc, omp=True, n_threads=th), oracle_from_scene(sc, omp=True, n_threads=max(2, th - 5))
os.environ.pop("MPMHIP_P2G_TILE", None)
fx = harness.build_solver(sc, "cuda:0", mode="fast")
os.environ["MPMHIP_P2G_TILE"] = "f64"
f64 = harness.build_solver(sc, "cuda:0", mode="fast")
os.environ.pop("MPMHIP_P2G_TILE", None)
rel = lambda x, y: float(np.abs(x - y).max() / max(np.abs(y).max(), 1e-3))
done, t0 = 0, time.time()
print(f"# {name}: distances from the oracle ({th} threads): oracle with {max(2, th - 5)} threads | HIP fixed-point tile | HIP fp64 tile   (rel dx / rel dv)")
for m in [10, 50, 100, 200, 300, 400, 500, 600, 700, 800, 900, 1000]:
    if m > n: break
    run_scene(oa, sc, m - done, k0=done); run_scene(ob, sc, m - done, k0=done)
    harness.run(fx, m - done, fused=True); harness.run(f64, m - done, fused=True)
    done = m
    row = [f"substep {m:5d}:"]
    for x, v in ((ob.x, ob.v), (fx.state.particle_x.cpu().numpy(), fx.state.particle_v.cpu().numpy()),
                 (f64.state.particle_x.cpu().numpy(), f64.state.particle_v.cpu().numpy())):
        row.append(f"{rel(x, oa.x):.2e} / {rel(v, oa.v):.2e}")
    print("  |  ".join(row) + f"   [{time.time() - t0:.0f} s]", flush=True)
