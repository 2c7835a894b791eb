import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
from mpmavatar_amd import harness, scenes
from oracle.scene_adapter import omp_threads, oracle_from_scene, run_scene
sc = scenes.REGISTRY["demo-250"]()
ne, nt, nv = sc.n_elements, sc.n_traditional, sc.n_vertices
o = oracle_from_scene(sc, omp=True, n_threads=omp_threads())
fx = harness.build_solver(sc, "cuda:0", mode="fast")
done = 0
for m in (200, 400, 600):
    run_scene(o, sc, m - done, k0=done); harness.run(fx, m - done, fused=True); done = m
    x, v = fx.state.particle_x.cpu().numpy().astype(np.float64), fx.state.particle_v.cpu().numpy().astype(np.float64)
    dx = x - o.x; dv = v - o.v
    e = np.abs(dx).max(axis=1)
    for name, sl in (("elements", slice(0, ne)), ("sand", slice(ne, ne + nt)), ("vertices", slice(ne + nt, None))):
        i = int(np.argmax(e[sl])) + (sl.start or 0)
        print(f"substep {m} {name:9s} max|dx| {e[sl].max():.2e} mean|dx| {np.abs(dx[sl]).mean():.2e} mean dx {dx[sl].mean(axis=0)}  worst #{i} x {o.x[i]} dx {dx[i]} v {o.v[i]} dv {dv[i]}", flush=True)
    print("   mean dv by class:", dv[:ne].mean(axis=0), dv[ne:ne+nt].mean(axis=0), dv[ne+nt:].mean(axis=0))
