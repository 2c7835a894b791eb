#!/usr/bin/env python
"""Node-level distance of the fast back end from the reference's traced substeps (tests/golden/ref_trace_*.npz), by node mass:
the packed fixed-point tile of p2g (default) rounds every contribution to a unit that is fixed per chunk, so nodes that only
receive tiny weights keep fewer significant bits than with the fp64 tile (MPMHIP_P2G_TILE=f64).
    python tools/gpu/grid_nodes.py            (run once per tile mode: the switch is read when the solver is built)"""
import os, sys
import numpy as np, torch
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import refgolden as rg
from mpmavatar_amd import harness

def one(name):
    z = rg.load(name)
    sc = rg.scene_from_npz(z)
    sim = harness.build_solver(sc, "cuda:0", mode="fast")
    st, md = sim.state, sim.model
    for kind, kw in rg.pre_ops(z):
        getattr(sim.solver, rg.PRE_OPS[kind])(st, **kw)
    harness.run(sim, 1)
    pre = rg.state_after(z, -1)
    dev = st.particle_x.device
    put = lambda dst, a: dst.copy_(torch.as_tensor(np.ascontiguousarray(a), device=dev).reshape(dst.shape))
    for f in ("particle_x", "particle_v", "particle_C", "particle_F_trial", "particle_F", "particle_d", "particle_stress"):
        if getattr(st, f).numel():
            put(getattr(st, f), pre[f])
    for f in ("mu", "lam", "yield_stress"):
        put(getattr(md, f), pre[f])
    harness.run(sim, 1)
    m, v_in, v_out = (a.detach().cpu().numpy().astype(np.float64) for a in sim.solver.export_grid())
    rm, rv = z["post_grid_m"].astype(np.float64), z["post_grid_v_out"].astype(np.float64)
    act = rm > 1e-13
    vmax, mmax = np.abs(rv[act]).max(), rm.max()
    ev = np.abs(v_out - rv).max(axis=-1) / max(vmax, 1e-3)
    ep = np.abs(m[..., None] * v_out - rm[..., None] * rv).max(axis=-1) / max(np.abs(rm[..., None] * rv).max(), 1e-12)
    em = np.abs(m - rm) / mmax
    out = [f"{name:24s} nodes {int(act.sum()):5d}  mass {em[act].max():.1e}  momentum {ep[act].max():.1e}  v:"]
    for lo in (1e-1, 1e-2, 1e-3, 1e-4, 1e-6, 0.0):
        sel = act & (rm >= lo * mmax)
        out.append(f">={lo:g}: {ev[sel].max():.1e} ({int(sel.sum())})")
    pv = rg.rel(sim.state.particle_v.detach().cpu().numpy(), z["post_particle_v"])
    out.append(f" particle v {pv:.1e}")
    print("  ".join(out), flush=True)
    sim.solver.close()

print("tile:", os.environ.get("MPMHIP_P2G_TILE", "fixed"))
for n in rg.names("trace"):
    one(n)
