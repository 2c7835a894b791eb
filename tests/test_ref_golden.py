"""The CPU oracle (oracle/mpm_oracle.c) and the float64 twin against fixtures produced by the reference's own source.

tests/golden/ref_*.npz were written by tests/golden/make_golden_ref.py, which imports
/root/reference/warp_mpm/{mpm_data_structure,mpm_utils,mpm_solver}.py UNCHANGED (over a NumPy stand-in for the ``warp``
module) and runs ``MPMWARP.p2g2p``.  This is what pins the oracle to the reference:

* ``ref_trace_*``: kernel by kernel.  Before every group of launches the oracle is loaded with the REFERENCE'S state, runs its
  restatement of that kernel, and must reproduce the arrays the reference's kernel wrote (so errors cannot accumulate or cancel).
* ``ref_seq_*``: whole substeps, tens of them, of the small test scenes.

Tolerances: fp32 with a different operation order (and float64-LAPACK vs Givens/Jacobi factorizations inside): 2e-5 relative
per kernel, 1e-4 (the north star's bound) on x and v for the sequences.
"""
import json

import numpy as np
import pytest

import refgolden as rg
from oracle.scene_adapter import oracle_from_scene

TRACES = rg.names("trace")
SEQS = rg.names("seq")

ORACLE_OF = {"particle_x": "x", "particle_v": "v", "particle_C": "C", "particle_F": "F", "particle_F_trial": "F_trial",
             "particle_stress": "stress", "particle_d": "d", "vertex_force": "vertex_force", "grid_m": "grid_m",
             "grid_v_in": "grid_v_in", "grid_v_out": "grid_v_out", "mu": "mu", "lam": "lam", "yield_stress": "yield_stress",
             "mesh_points": "mesh_points", "mesh_velocities": "mesh_velocities"}
COL_OF = {"weight": "weight", "mesh_v_in": "v_in", "mesh_v_out": "v_out", "mesh_normal": "normal"}


def _oracle_array(o, key):
    if key in ORACLE_OF:
        return getattr(o, ORACLE_OF[key])
    kind, field = key.split("_", 1)
    k = int(kind[3:])
    return o.mesh_colliders[k][COL_OF[field]] if kind.startswith("col") else o.movers[k][field]


def set_state(o, st):
    for key, val in st.items():
        if key == "time":
            o.sim.time = float(val)
            continue
        a = _oracle_array(o, key)
        a[...] = np.asarray(val, np.float32).reshape(a.shape)


def check(o, st, keys, tol, what):
    for key in keys:
        a = _oracle_array(o, key)
        ref = np.asarray(st[key]).reshape(a.shape)
        assert np.isfinite(ref).all(), (what, key)
        err = rg.rel(a, ref, floor=1e-6 if key.startswith(("grid_m", "col", "mov")) else 1e-3)
        assert err < tol, f"{what}: {key} differs from the reference by {err:.2e}"


def build_oracle(z):
    sc = rg.scene_from_npz(z)
    o = oracle_from_scene(sc)
    for kind, kw in rg.pre_ops(z) if "preops_json" in z.files else []:
        kw = dict(kw)
        getattr(o, rg.PRE_OPS[kind])(**kw)
    return sc, o


# groups of reference launches -> the oracle call that restates them and the arrays they write
def _groups(names):
    i, out = 0, []
    while i < len(names):
        q = names[i]
        j = i + 1
        if ".add_mesh_collider." in q or ".add_particle_mover." in q:
            owner = q.split(".<locals>.")[0]
            # one collider / mover = one run of launches ending with its `collide` / `normalize_grid` kernel
            last = "collide" if "mesh_collider" in q else "normalize_grid"
            while not names[j - 1].endswith("." + last) or names[j - 1].split(".<locals>.")[0] != owner:
                j += 1
            # (the mesh collider also has a normalize_grid; its run ends at collide)
        elif q.endswith((".apply_force", ".modify_particle_v_before_p2g")):
            while j < len(names) and names[j].endswith((".apply_force", ".modify_particle_v_before_p2g")):
                j += 1
        out.append((i, j - 1, q))
        i = j
    return out


@pytest.mark.parametrize("name", TRACES)
def test_oracle_kernel_by_kernel(name, oracle_lib):
    z = rg.load(name)
    sc, o = build_oracle(z)
    names = rg.launches(z)
    dt = np.float32(sc.dt)
    inp = rg.step_inputs(sc, 1)  # the traced substep is the second one
    n_bc = n_col = n_mov = 0
    seen = set()
    mesh_uploads = 0
    for first, last, q in _groups(names):
        before, after = rg.state_after(z, first - 1), rg.state_after(z, last)
        set_state(o, before)
        written = sorted({k[4:] for i in range(first, last + 1) for k in z.files if k.startswith(f"L{i:02d}_")})
        base = q.split(".")[-1]
        if q == "zero_grid":
            o.zero_grid()
            written = ["grid_m", "grid_v_in", "grid_v_out"]
        elif q == "set_vec3_to_zero":
            o.vertex_force[:] = 0  # memset in orc_p2g2p
            written = ["vertex_force"]
        elif base in ("apply_force", "modify_particle_v_before_p2g"):
            o.pre_p2g(dt)
        elif q == "set_vec3_to_vec3":  # body mesh upload (memcpy in orc_p2g2p): points first, then velocities
            (o.mesh_points if mesh_uploads == 0 else o.mesh_velocities)[...] = inp["mesh_x" if mesh_uploads == 0 else "mesh_v"]
            mesh_uploads += 1
        elif q == "compute_stress_from_F_trial":
            o.compute_stress(dt)
        elif q == "p2g_apic_with_stress":
            o.p2g(dt)
        elif q == "grid_normalization_and_gravity":
            o.grid_update(dt)
        elif q == "add_damping_via_grid":
            o.damping(np.float32(o.sim.grid_v_damping_scale))
        elif ".add_mesh_collider." in q:
            o.mesh_collide(n_col)
            n_col += 1
        elif ".add_particle_mover." in q:
            o.particle_move(n_mov, inp.get("joint_traditional_v"), inp["joint_verts_v"], inp["joint_faces_v"])
            n_mov += 1
        elif base == "collide":  # grid boundary conditions, in registration order
            o.apply_bc(n_bc, dt)
            n_bc += 1
        elif q == "g2p_v":
            o.g2p_v(dt)
        elif q == "g2p_e":
            o.g2p_e(dt)
        else:
            raise AssertionError(f"reference launched a kernel this test does not know: {q}")
        seen.add(q if "." not in q else q.split(".", 1)[1])
        check(o, after, written, 2e-5, f"{name}: {q}")
        # nothing else may have been touched
        untouched = [k for k in after if k != "time" and k not in written]
        check(o, after, untouched, 1e-12, f"{name}: {q} (arrays it must not write)")
    assert {"compute_stress_from_F_trial", "p2g_apic_with_stress", "grid_normalization_and_gravity", "g2p_v", "g2p_e"} <= seen


@pytest.mark.parametrize("name", TRACES)
def test_oracle_whole_traced_substep(name, oracle_lib):
    """Same fixtures, one call of the oracle's p2g2p from the reference's pre-state (launch order, host-side updates)."""
    z = rg.load(name)
    sc, o = build_oracle(z)
    set_state(o, rg.state_after(z, -1))
    o.p2g2p(sc.dt, **rg.step_inputs(sc, 1))
    post = {k[5:]: z[k] for k in z.files if k.startswith("post_")}
    # (errors of stress -> p2g -> v_in / m add up on nodes that hold almost no mass: the grid velocity gets 2e-4)
    check(o, post, [k for k in post if k != "time" and not k.startswith("grid_v")], 5e-5, name)
    check(o, post, ["grid_v_in", "grid_v_out"], 2e-4, name)
    assert abs(o.time - float(z["time_after"])) < 1e-12


seq_bound = rg.seq_bound


@pytest.mark.parametrize("name", SEQS)
def test_oracle_sequences(name, oracle_lib):
    """Whole substeps.  Positions: 1e-4 everywhere (measured <= 2e-6).  Velocities: 1e-4 for the elastic solid and for cloth
    without the shear-friction discontinuity (gamma = 0; measured <= 4e-6 over 100-200 substeps); within the reference's
    own fp32 sensitivity envelope for plastic flow and for cloth with gamma > 0."""
    z = rg.load(name)
    sc = rg.scene_from_npz(z)
    o = oracle_from_scene(sc)
    done = 0
    strict = name.endswith(("_jelly", "_gamma0"))
    for cp in z["checkpoints"]:
        for k in range(done, int(cp)):
            o.p2g2p(sc.dt, **rg.step_inputs(sc, k))
        done = int(cp)
        ex = rg.rel(o.x, z[f"s{cp}_particle_x"])
        ev = rg.rel(o.v, z[f"s{cp}_particle_v"])
        bound = 1e-4 if strict else seq_bound(z, cp)
        assert ex < 1e-4 and ev < bound, f"{name}: substep {cp}: x {ex:.2e}, v {ev:.2e} (bound {bound:.2e})"
        if cp <= 10:
            scale_C = max(float(np.abs(z[f"s{cp}_particle_v"]).max()), 1e-3) * sc.n_grid / sc.grid_lim   # |v| / dx
            assert rg.rel(o.C, z[f"s{cp}_particle_C"], floor=scale_C) < 2e-4
            assert rg.rel(o.F_trial, z[f"s{cp}_particle_F_trial"]) < 1e-4
            assert rg.rel(o.d, z[f"s{cp}_particle_d"]) < max(1e-4, seq_bound(z, cp, "particle_d"))


def test_cloth_sensitivity_is_the_references_own():
    """The claim behind the envelope, checked on the fixtures: with gamma > 0 the reference run with fp32-accurate svd3 / qr3
    leaves the fp64-accurate one by >= 3e-4 in v within 40 substeps (the R22 = 1 discontinuity, mpm_utils.py:196-204),
    while the same scenes with gamma = 0 stay together to 1e-5 -- and positions agree to 1e-5 in every case."""
    for name, lo, hi in (("ref_seq_sheet", 3e-4, None), ("ref_seq_garment", 3e-4, None), ("ref_seq_sheet_gamma0", None, 1e-5),
                         ("ref_seq_garment_gamma0", None, 1e-5)):
        z = rg.load(name)
        cps = [int(c) for c in z["checkpoints"]]
        spread_v = max(rg.rel(z[f"alt_s{c}_particle_v"], z[f"s{c}_particle_v"]) for c in cps)
        spread_x = max(rg.rel(z[f"alt_s{c}_particle_x"], z[f"s{c}_particle_x"]) for c in cps)
        assert spread_x < 1e-5, (name, spread_x)
        if lo is not None:
            assert spread_v > lo, (name, spread_v)
        if hi is not None:
            assert spread_v < hi, (name, spread_v)


def test_fixture_inventory():
    """Every live kernel of the path appears in at least one trace."""
    seen = set()
    for n in TRACES:
        seen |= {q.split(".<locals>.")[-1] if "<locals>" not in q else q.split(".", 1)[1] for q in rg.launches(rg.load(n))}
    need = {"zero_grid", "set_vec3_to_zero", "compute_stress_from_F_trial", "p2g_apic_with_stress",
            "grid_normalization_and_gravity", "add_damping_via_grid", "g2p_v", "g2p_e", "set_vec3_to_vec3",
            "add_mesh_collider.<locals>.zero_grid", "add_mesh_collider.<locals>.compute_mesh",
            "add_mesh_collider.<locals>.normalize_grid", "add_mesh_collider.<locals>.collide",
            "add_particle_mover.<locals>.zero_grid", "add_particle_mover.<locals>.add_velocity_traditional",
            "add_particle_mover.<locals>.add_velocity_verts", "add_particle_mover.<locals>.add_velocity_faces",
            "add_particle_mover.<locals>.normalize_grid", "add_bounding_box.<locals>.collide",
            "add_surface_collider.<locals>.collide", "set_velocity_on_cuboid.<locals>.collide",
            "add_impulse_on_particles.<locals>.apply_force",
            "enforce_particle_velocity_translation.<locals>.modify_particle_v_before_p2g",
            "enforce_particle_velocity_rotation.<locals>.modify_particle_v_before_p2g"}
    assert need <= seen, need - seen
    mats = {json.loads(str(rg.load(n)["scene_meta"]))["params"]["material"] for n in TRACES}
    assert {"jelly", "metal", "sand", "foam", "snow", "plasticine", "cloth"} <= mats


def test_third_convention_mcadams_svd_givens_qr():
    """VERDICT r3 item 4c.  tests/golden/alt3_ref_seq_*.npz: the reference's own source run a third time with svd3 / qr3 evaluated by
    the PUBLISHED algorithm behind Warp's builtins (McAdams et al. 2011: fp32 Jacobi eigenanalysis with approximate Givens
    quaternions, sorted singular values, Givens-quaternion QR; tests/golden/warp_standin SVD_MODE "mcadams" / QR_MODE "givens").
    What it says about the envelopes (distances of the reference from ITSELF, max over the checkpoints, velocities):
      * traditional materials (svd3): the third convention stays CLOSER to the fp64-accurate run than the second did -- jelly 5e-7
        (alt 3e-6), metal 2e-5 (4e-4), sand 3e-5 (7e-4): the bounds of the plastic cases do not change;
      * friction cloth (qr3): FARTHER -- sheet 2.9e-3 (alt 1.2e-3), garment 2.6e-3 (7.2e-4): if Warp's qr3 rounds like the published
        Givens-quaternion scheme, the reference's own sensitivity envelope for gamma > 0 is 2.4-3.6x WIDER than the one the tests
        use.  The tests keep the tighter envelope (alt / alt2): the oracle and the HIP path pass it;
      * positions agree to 3e-6 everywhere, and cloth without the R22 discontinuity (gamma = 0) stays at 1.2e-5 over 200 substeps."""
    import glob
    import os
    rows = {}
    for path in sorted(glob.glob(os.path.join(rg.GOLDEN, "alt3_ref_seq_*.npz"))):
        name = os.path.basename(path)[5:-4]
        z, a = rg.load(name), np.load(path)
        cps = [int(c) for c in a["checkpoints"]]
        e3 = max(rg.rel(a[f"alt3_s{c}_particle_v"], z[f"s{c}_particle_v"]) for c in cps)
        e1 = max(rg.rel(z[f"alt_s{c}_particle_v"], z[f"s{c}_particle_v"]) for c in cps)
        ex = max(rg.rel(a[f"alt3_s{c}_particle_x"], z[f"s{c}_particle_x"]) for c in cps)
        rows[name] = (e3, e1, ex)
        print(f"{name}: v envelope alt3 (mcadams / givens) {e3:.2e}, alt (fp32 Jacobi / Gram-Schmidt) {e1:.2e}; x {ex:.2e}")
        assert ex < 1e-5, (name, ex)
    assert {"ref_seq_cube_jelly", "ref_seq_cube_sand", "ref_seq_cube_metal", "ref_seq_sheet", "ref_seq_garment"} <= set(rows)
    for name in ("ref_seq_cube_jelly", "ref_seq_cube_sand", "ref_seq_cube_metal"):
        assert rows[name][0] <= rows[name][1], (name, rows[name])          # svd3: bounds unchanged
    assert rows["ref_seq_cube_jelly"][0] < 1e-4                              # the strict case stays strict
    for name in ("ref_seq_sheet", "ref_seq_garment"):
        assert 1.5 * rows[name][1] < rows[name][0] < 6 * rows[name][1], (name, rows[name])   # qr3: 2.4-3.6x wider, not used to loosen
    if "ref_seq_sheet_gamma0" in rows:
        assert rows["ref_seq_sheet_gamma0"][0] < 1e-4                   # (measured 1.2e-5 over 200 substeps: inside the north-star bound)
