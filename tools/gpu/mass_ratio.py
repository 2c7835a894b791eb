#!/usr/bin/env python
"""Fixed-point p2g tile against particle-mass ratios inside one chunk.

A cloth sheet with a sand block lying directly on it (same 4^3 blocks, so the same chunks); the sand's particle volume -- mass and
internal force alike -- is scaled so that (sand mass) / (cloth vertex mass) sweeps 1e+6 ... 1e-6.  60 substeps, the fast back end
with the fixed-point tile and with MPMHIP_P2G_TILE=f64 against the CPU oracle: per-class relative error of x and v.
    python tools/gpu/mass_ratio.py [substeps]"""
import os, subprocess, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def scene(ratio):
    from mpmavatar_amd import scenes
    sc = scenes.demo_mix(n_grid=64, n_sheet=24, sand=(24, 3, 12), hold=False, n_steps=60)
    ne, nt = sc.n_elements, sc.n_traditional
    x = sc.x.copy()
    x[ne:ne + nt, 1] -= (x[ne:ne + nt, 1].min() - 1.262)       # the sand's lowest layer 0.4 cells above the sheet (y = 1.25)
    x[ne:ne + nt, 0] += 0.25; x[ne:ne + nt, 2] += 0.02         # over the middle of the sheet
    sc.x = x
    vol = sc.vol.copy()
    v_cloth = float(vol[ne + nt:].mean())
    vol[ne:ne + nt] = np.float32(v_cloth * ratio)
    sc.vol = vol
    sc.name = f"mix-ratio-{ratio:g}"
    return sc


def worker(ratio, n):
    import torch
    from mpmavatar_amd import harness
    sc = scene(ratio)
    sim = harness.build_solver(sc, "cuda:0", mode="fast")
    harness.run(sim, n, fused=True)
    np.savez(sys.argv[4], x=sim.state.particle_x.cpu().numpy(), v=sim.state.particle_v.cpu().numpy(),
             dropped=sim.solver.stats()["n_dropped"])


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "--worker":
        worker(float(sys.argv[2]), int(sys.argv[3]))
        sys.exit(0)
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 60
    from oracle.scene_adapter import oracle_from_scene, run_scene
    rel = lambda a, b: float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-3))
    rows = []
    for ratio in (1e6, 1e4, 1e2, 1.0, 1e-2, 1e-4, 1e-6):
        sc = scene(ratio)
        o = oracle_from_scene(sc)
        run_scene(o, sc, n)
        ne, nt = sc.n_elements, sc.n_traditional
        cl = np.r_[0:ne, ne + nt:sc.n_particles]
        sd = np.arange(ne, ne + nt)
        row = {"ratio": ratio}
        for tile in ("fx", "f64", "auto"):
            env = dict(os.environ)
            env.pop("MPMHIP_P2G_TILE", None)
            if tile != "auto":
                env["MPMHIP_P2G_TILE"] = tile
            out = f"/tmp/mr_{tile}.npz"
            r = subprocess.run([sys.executable, __file__, "--worker", str(ratio), str(n), out], env=env, capture_output=True, text=True)
            if r.returncode:
                print(r.stderr[-1500:]); sys.exit(1)
            g = np.load(out)
            row[tile] = {"x_cloth": rel(g["x"][cl], o.x[cl]), "v_cloth": rel(g["v"][cl], o.v[cl]), "x_sand": rel(g["x"][sd], o.x[sd]),
                         "v_sand": rel(g["v"][sd], o.v[sd])}
        rows.append(row)
        print(json.dumps(row), flush=True)
