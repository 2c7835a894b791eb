import sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, "tests")
import numpy as np
from mpmavatar_amd import harness, scenes
from oracle.scene_adapter import oracle_from_scene, run_scene
import refgolden as rg
for name, mk, n in (("small_sheet", scenes.small_sheet, 200), ("small_garment", scenes.small_garment, 100),
                    ("demo_mix", lambda: scenes.demo_mix(n_grid=48, n_sheet=16, sand=(16, 3, 8)), 100)):
    sc = mk(); o = oracle_from_scene(sc)
    sims = {m: harness.build_solver(mk(), "cuda:0", mode=m) for m in ("fast", "baseline")}
    done = 0
    for cp in (20, 50, 100, 200):
        if cp > n: break
        run_scene(o, sc, cp - done, k0=done)
        for m, sim in sims.items(): harness.run(sim, cp - done)
        done = cp
        row = []
        for m, sim in sims.items():
            v = sim.state.particle_v.cpu().numpy(); x = sim.state.particle_x.cpu().numpy()
            row.append("%s: x %.1e v rel %.1e abs %.1e d %.1e" % (m, rg.rel(x, o.x), rg.rel(v, o.v), np.abs(v - o.v).max(),
                       rg.rel(sim.state.particle_d.cpu().numpy(), o.d) if sc.n_elements else 0))
        print(name, cp, "vmax %.3f" % np.abs(o.v).max(), " | ".join(row), flush=True)
for f in ("ref_seq_sheet", "ref_seq_garment", "ref_seq_demo"):
    z = rg.load(f)
    print(f, [(int(c), float(np.abs(z[f"alt_s{c}_particle_v"] - z[f"s{c}_particle_v"]).max())) for c in z["checkpoints"]])
