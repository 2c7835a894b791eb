#!/usr/bin/env python
"""Randomised cross-check of the two kernel back ends (fast vs reference-structured baseline): random blobs / sheets,
grid sizes, materials, velocities, time steps and re-sort policies.  Prints the worst case; exits 1 on a mismatch.
    python tools/gpu/fuzz.py [n_cases] [seed]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from mpmavatar_amd import harness, scenes
from mpmavatar_amd.scenes import _trad_scene

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 30
only = int(sys.argv[3]) if len(sys.argv) > 3 else None   # re-run one case of a seed with extra diagnostics
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
rel = lambda a, b: float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-3))
worst = (0.0, None)
bad = 0
for case in range(n_cases):
    kind = rng.choice(["blob", "sheet", "garment", "demo"])
    n_grid = int(rng.choice([24, 30, 32, 40, 48, 50]))
    steps = int(rng.integers(20, 120))
    ri = int(rng.choice([0, 0, 7, -1000000]))
    dt_scale = float(rng.choice([1.0, 3.0, 10.0]))
    if kind == "blob":
        n = int(rng.integers(1, 4000))
        dx = 2.0 / n_grid
        c = rng.uniform(0.5, 1.5, 3)
        pts = (c + rng.uniform(-1, 1, (n, 3)) * rng.uniform(0.5, 6) * dx).clip(3 * dx, 2 - 3 * dx).astype(np.float32)
        r = pts - pts.mean(0)
        vel = (np.cross(rng.normal(0, 6, 3), r) + rng.normal(0, 3.0, 3)).astype(np.float32)
        mat = str(rng.choice(["jelly", "sand", "metal", "foam", "plasticine"]))
        params = {"friction_angle": 35.0} if mat == "sand" else ({"yield_stress": 2.0, "hardening": 1, "xi": 0.1, "plastic_viscosity": 0.5} if mat != "jelly" else {})
        if rng.random() < 0.3: params["rpic_damping"] = float(rng.choice([0.2, -1.0]))
        sc = _trad_scene("fuzz", pts, (dx / 2) ** 3, n_grid, material=mat, v=vel, E=float(rng.choice([20.0, 100.0])), bcs=[("bounding_box", {})], params=params)
        vtol = 2e-3
    elif kind == "sheet":
        sc = scenes.sheet(n=int(rng.integers(8, 40)), n_grid=n_grid, collider_subdiv=int(rng.integers(1, 4)), span=(0.6, 1.4), y=1.2,
                          sphere_r=0.2, sphere_c=(1.0, 0.97, 1.0), name="fuzz")
        sc.v = (sc.v + rng.normal(0, 2.5, 3).astype(np.float32)).astype(np.float32)
        vtol = 5e-2
    elif kind == "garment":
        sc = scenes.garment_cylinder(n_theta=int(rng.integers(12, 48)), n_h=int(rng.integers(8, 30)), n_grid=n_grid, aniso=bool(rng.random() < 0.7),
                                     collider_subdiv=2, name="fuzz")
        vtol = 5e-2
    else:
        sc = scenes.demo_mix(n_grid=n_grid, n_sheet=int(rng.integers(8, 24)), sand=(int(rng.integers(4, 20)), int(rng.integers(2, 5)), int(rng.integers(4, 12))),
                             hold=(int(rng.integers(0, 30)), int(rng.integers(1, 9)), int(rng.integers(1, 200))) if rng.random() < 0.7 else False)
        vtol = 5e-2
    # random grid boundary conditions and model scalars on top
    if rng.random() < 0.5:
        lo = float(sc.x[:, 1].min())
        sc.bcs = list(sc.bcs) + [("surface_collider", {"point": [0.0, lo - float(rng.uniform(0.0, 0.05)), 0.0], "normal": [0.0, 1.0, 0.0],
                                                        "surface": str(rng.choice(["sticky", "slip", "cut"])), "friction": 0.0})]
    if rng.random() < 0.3:
        c = sc.x.mean(0)
        sc.bcs = list(sc.bcs) + [("velocity_cuboid", {"point": [float(c[0]) + 0.1, float(c[1]), float(c[2])], "size": [0.05, 0.3, 0.3],
                                                       "velocity": [float(rng.normal(0, 0.5)), 0.0, 0.0], "start_time": 0.0,
                                                       "end_time": float(rng.choice([0.002, 999.0])), "reset": int(rng.integers(0, 2))})]
    if rng.random() < 0.2:
        sc.params = dict(sc.params, grid_v_damping_scale=float(rng.choice([0.9, 0.99])))
    os.environ["MPMHIP_FUSE_GRID"] = str(int(rng.random() < 0.8))
    os.environ["MPMHIP_FUSE_TRAD"] = str(int(rng.random() < 0.8))
    os.environ["MPMHIP_PREDICTIVE_SORT"] = str(int(rng.random() < 0.8))
    for kv in os.environ.get("FUZZ_ENV", "").split():   # replay with some switches forced, e.g. FUZZ_ENV="MPMHIP_FUSE_GRID=0"
        k, v = kv.split("="); os.environ[k] = v
    sc.dt = 1e-4 * dt_scale
    if kind != "blob" and dt_scale > 3: sc.dt = 3e-4
    fused = bool(rng.random() < 0.7)
    if only is not None:
        if case != only:
            continue
        from oracle.scene_adapter import oracle_from_scene, run_scene
        print("  scene:", kind, "bcs", sc.bcs, "params", sc.params, "env", {k: os.environ[k] for k in ("MPMHIP_FUSE_GRID", "MPMHIP_FUSE_TRAD", "MPMHIP_PREDICTIVE_SORT")}, flush=True)
        o = oracle_from_scene(sc); run_scene(o, sc, steps)
        if os.environ.get("FUZZ_CPU_TWIN"):   # no GPU: how far apart are the fp32 oracle and the float64 twin on this case?
            from oracle.twin import TwinMPM
            tw = TwinMPM(sc); o3 = oracle_from_scene(sc)
            for k in range(steps):
                run_scene(tw, sc, 1, k0=k); run_scene(o3, sc, 1, k0=k)
                if k < 6 or k % 8 == 0 or k == steps - 1:
                    print(f"    step {k + 1}: oracle-twin dx {rel(o3.x, tw.x):.1e} dv {rel(o3.v, tw.v):.1e}  max|v| {np.abs(tw.v).max():.2f}", flush=True)
            sys.exit(0)
        if os.environ.get("FUZZ_TRACE") == "finalize":   # fused element finalize (in the stress kernel) vs the stand-alone kernel
            for kk in (1, 2, 3, 4, 6, 8, 12, 16, 24, 32, 48, 64):
                if kk > steps: break
                os.environ["MPMHIP_DBG"] = "0"; A = harness.build_solver(sc, "cuda:0", mode="fast", rebin_interval=-1000000); harness.run(A, kk, fused=True)
                os.environ["MPMHIP_DBG"] = "64"; B = harness.build_solver(sc, "cuda:0", mode="fast", rebin_interval=-1000000); harness.run(B, kk, fused=True)
                ne = sc.n_elements
                f = lambda t, n: getattr(t.state, n).cpu().numpy().reshape(sc.n_particles if n != "particle_d" else ne, -1)
                out = []
                for n in ("particle_x", "particle_v", "particle_C", "particle_d"):
                    a, b = f(A, n), f(B, n)
                    d = np.abs(a - b).max(1)
                    out.append(f"{n[9:]} max {d.max():.1e} at {int(d.argmax())}")
                print(f"    after {kk} substeps: " + "; ".join(out), flush=True)
            sys.exit(0)
        if os.environ.get("FUZZ_TRACE") == "finalize2":
            def run(dbg, kk):
                os.environ["MPMHIP_DBG"] = dbg
                t = harness.build_solver(sc, "cuda:0", mode="fast", rebin_interval=-1000000); harness.run(t, kk, fused=True)
                return t.state.particle_d.cpu().numpy().reshape(sc.n_elements, 9), t.state.particle_x.cpu().numpy()
            for kk in (8, 9, 10, 11, 12):
                A1, _ = run("0", kk); A2, _ = run("0", kk); B1, xb = run("64", kk); B2, _ = run("64", kk)
                o4 = oracle_from_scene(sc); run_scene(o4, sc, kk)
                od = np.asarray(o4.d).reshape(sc.n_elements, 9)
                dd = np.abs(A1 - B1).max(1); e = int(dd.argmax())
                print(f"    k={kk}: A1-A2 {np.abs(A1 - A2).max():.1e}  B1-B2 {np.abs(B1 - B2).max():.1e}  A1-B1 {dd.max():.1e} at {e}  A1-oracle {np.abs(A1 - od).max():.1e}  B1-oracle {np.abs(B1 - od).max():.1e}", flush=True)
                print("       A d[e]:", np.round(A1[e], 5), "\n       B d[e]:", np.round(B1[e], 5), "\n       O d[e]:", np.round(od[e], 5), flush=True)
            sys.exit(0)
        if os.environ.get("FUZZ_TRACE") == "modes":   # how the same fast solver is driven
            for label, drive in (("one fused call", lambda t: harness.run(t, steps, fused=True)),
                                 ("per-step calls, no read-back", lambda t: harness.run(t, steps, fused=False)),
                                 ("fused calls of 10 with a read-back in between", lambda t: [(harness.run(t, min(10, steps - k), fused=True), t.state.particle_x.sum().item()) for k in range(0, steps, 10)]),
                                 ("fused calls of 10 + stats()", lambda t: [(harness.run(t, min(10, steps - k), fused=True), t.solver.stats()) for k in range(0, steps, 10)]),
                                 ("fused calls of 10 + synchronize()", lambda t: [(harness.run(t, min(10, steps - k), fused=True), t.solver.synchronize()) for k in range(0, steps, 10)]),
                                 ("fused calls of 10, no read-back", lambda t: [harness.run(t, min(10, steps - k), fused=True) for k in range(0, steps, 10)])):
                t = harness.build_solver(sc, "cuda:0", mode="fast", rebin_interval=-1000000); drive(t)
                st = t.solver.stats()
                print(f"  {label:48s}: dx {rel(t.state.particle_x.cpu().numpy(), o.x):.1e} dv {rel(t.state.particle_v.cpu().numpy(), o.v):.1e} rebins {st['rebins']} fallback {st['n_fallback_particles']}", flush=True)
            sys.exit(0)
        if os.environ.get("FUZZ_TRACE"):   # step-by-step divergence of the baseline back end from the oracle
            o2 = oracle_from_scene(sc)
            tb = (harness.build_solver(sc, "cuda:0", mode="fast", rebin_interval=-1000000) if os.environ["FUZZ_TRACE"] == "fast"
                  else harness.build_solver(sc, "cuda:0", mode="baseline"))
            for k in range(steps):
                run_scene(o2, sc, 1, k0=k); harness.run(tb, 1, fused=False)
                if k == 0:
                    G = sc.n_grid
                    og = np.asarray(o2.grid_v_out).reshape(G, G, G, 3); om = np.asarray(o2.grid_m).reshape(G, G, G)
                    m, vi, vo = tb.solver.export_grid(); vo = vo.cpu().numpy()
                    d = np.abs(vo - og).max(-1) * (om > 0)
                    print("    nodes differing after step 1:", int((d > 1e-3).sum()), "of", int((om > 0).sum()))
                    for i, j, kk in np.argwhere(d > 1e-3)[:10]:
                        print("     node", (i, j, kk), "gpu", vo[i, j, kk], "oracle", og[i, j, kk], "m", om[i, j, kk])
                if k < 12 or k % 8 == 0:
                    print(f"    step {k + 1}: {tb.solver._mode if hasattr(tb.solver, '_mode') else ''} gpu-oracle dx {rel(tb.state.particle_x.cpu().numpy(), o2.x):.1e} dv {rel(tb.state.particle_v.cpu().numpy(), o2.v):.1e}", flush=True)
        for label, kw in (("fast adaptive", dict(mode="fast", rebin_interval=0)), ("fast single sort", dict(mode="fast", rebin_interval=-1000000)),
                          ("fast every 5", dict(mode="fast", rebin_interval=-5)), ("baseline", dict(mode="baseline"))):
            t = harness.build_solver(sc, "cuda:0", **kw); harness.run(t, steps, fused=fused)
            x, v = t.state.particle_x.cpu().numpy(), t.state.particle_v.cpu().numpy()
            st = t.solver.stats()
            print(f"  {label:18s} vs oracle: dx {rel(x, o.x):.1e} dv {rel(v, o.v):.1e}  rebins {st['rebins']} fallback {st['n_fallback_particles']} dropped {st['n_dropped']} max|v| {np.abs(o.v).max():.2f}", flush=True)
    a = harness.build_solver(sc, "cuda:0", mode="fast", rebin_interval=ri)
    b = harness.build_solver(sc, "cuda:0", mode="baseline")
    harness.run(a, steps, fused=fused); harness.run(b, steps, fused=True)
    xa, xb = a.state.particle_x.cpu().numpy(), b.state.particle_x.cpu().numpy()
    va, vb = a.state.particle_v.cpu().numpy(), b.state.particle_v.cpu().numpy()
    # velocities: relative to at least 0.05 m/s (a scene pinned by a plane collider has |v| ~ 1e-4: rounding noise only)
    ex, ev = (rel(xa, xb), float(np.abs(va - vb).max() / max(np.abs(vb).max(), 0.05))) if xa.size else (0.0, 0.0)
    st = a.solver.stats()
    dropped = st["n_dropped"]
    if dropped and ri < 0:   # single sort + particles that travelled past the active blocks: outside the solver's contract
        print(f"skip case {case}: fixed-interval mode outran its active blocks ({dropped} dropped contributions)", flush=True)
        continue
    ok = np.isfinite(xa).all() and ex < 2e-4 and ev < vtol and dropped == 0
    note = ""
    if not ok and kind != "blob" and dropped == 0 and np.isfinite(xa).all():
        # cloth at rest sits on the return mapping's discontinuity (R22 = 1): rounding noise picks the branch and two correct
        # implementations drift apart.  Call it a failure only if the fast back end is further from the baseline than the
        # baseline is from the CPU oracle (two implementations with the reference's own structure).
        from oracle.scene_adapter import oracle_from_scene, run_scene
        o = oracle_from_scene(sc); run_scene(o, sc, steps)
        ebx, ebv = rel(xb, o.x), float(np.abs(vb - o.v).max() / max(np.abs(o.v).max(), 0.05))
        if ex <= 3 * ebx and ev <= 3 * ebv:
            ok, note = True, f"  [sensitive case: baseline vs oracle dx {ebx:.1e} dv {ebv:.1e}]"
    desc = f"case {case}: {kind} n_p={sc.n_particles} grid={n_grid} steps={steps} dt={sc.dt:g} rebin={ri} fused={fused} mat={sc.params.get('material')} -> dx {ex:.1e} dv {ev:.1e} rebins {st['rebins']} fallback {st['n_fallback_particles']}"
    print(("ok   " if ok else "FAIL ") + desc + note, flush=True)
    bad += 0 if ok else 1
    if ex > worst[0]: worst = (ex, desc)
print("worst:", worst[1])
sys.exit(1 if bad else 0)
