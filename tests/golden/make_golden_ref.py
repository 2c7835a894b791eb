"""Fixtures produced by the REFERENCE'S OWN SOURCE: tests/golden/ref_*.npz  (build container only).

    python tests/golden/make_golden_ref.py [name ...]        # default: every fixture (several minutes each)

The reference's substep is Warp DSL, Warp is not installable here -- but the kernel bodies are plain Python.  This script
puts ``tests/golden/warp_standin`` (a NumPy stand-in for the ``warp`` module: serial tid loop, fp32 scalars, see its
docstring for exactly what it assumes) in front of ``sys.path`` and imports

    /root/reference/warp_mpm/mpm_data_structure.py, mpm_utils.py, mpm_solver.py     UNCHANGED, from where they lie,

builds ``MPMStateStruct`` / ``MPMModelStruct`` / ``MPMWARP`` through the call sequence of the reference's drivers
(train_material_params.py:403-506) and calls ``MPMWARP.p2g2p`` -- every launch, every kernel body and every host-side
line that runs is the reference's.  Two kinds of fixture:

  ref_trace_<case>.npz  one substep (the second of the run, so that every array holds leftovers) from a RANDOM state; the complete state before it and, after every ``wp.launch`` the
                        reference issues, the arrays that kernel writes (so each kernel is pinned on its own:
                        compute_stress_from_F_trial for every material, p2g_apic_with_stress,
                        grid_normalization_and_gravity, add_damping_via_grid, the mesh collider's four kernels, the
                        particle mover's five, every grid BC, the pre-p2g particle operations, g2p_v, g2p_e)
  ref_seq_<case>.npz    whole-substep sequences (tens of substeps) of the small test scenes, state at checkpoints

Only data is written (inputs + the reference's outputs); nothing of the reference travels.  ``svd3`` / ``qr3`` are the
stand-in's (Warp's are out of tree): every trace fixture is generated under BOTH SVD and BOTH QR conventions of the
stand-in and the script refuses to write it unless the outputs agree to 2e-6 -- inside the tested domain (det F > 0,
non-degenerate triangles) the reference's results do not depend on those conventions.
"""
from __future__ import annotations

import json
import os
import sys
import time

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference/warp_mpm"
sys.path.insert(0, os.path.join(HERE, "warp_standin"))
sys.path.insert(0, REF)
sys.path.insert(0, ROOT)

import warp as wp  # noqa: E402  (the stand-in)
import mpm_solver as ref_solver  # noqa: E402  (the reference, unchanged)
from mpm_data_structure import MPMModelStruct, MPMStateStruct  # noqa: E402
from mpm_solver import MPMWARP  # noqa: E402

from mpmavatar_amd import garment, scenes  # noqa: E402  (scene descriptions only: NumPy)

assert ref_solver.__file__.startswith("/root/reference/"), ref_solver.__file__


# ------------------------------------------------------------------------------------------------ driving the reference
class RefSim:
    pass


def build_reference(sc) -> RefSim:
    """Same call sequence as mpmavatar_amd.harness.build_solver (= train_material_params.py:403-506), on the reference."""
    def t(a, dt=torch.float32):
        a = np.ascontiguousarray(a)
        return torch.zeros(a.shape, dtype=dt) if a.size == 0 else torch.as_tensor(a, dtype=dt)  # (empty numpy views carry odd strides)

    n_p, n_e, n_v, n_t = sc.n_particles, sc.n_elements, sc.n_vertices, sc.n_traditional
    dev = "cpu"
    state = MPMStateStruct()
    state.init(n_p, n_e, n_v, device=dev, requires_grad=True)
    flags = np.zeros((3, n_p), np.int32)
    flags[0, n_e:n_e + n_t] = 1
    flags[1, n_e + n_t:] = 1
    flags[2, :n_e] = 1
    D_inv = np.linalg.inv(sc.d.astype(np.float64)).astype(np.float32) if n_e else np.zeros((0, 3, 3), np.float32)
    state.from_torch(t(sc.x), t(sc.vol), t(D_inv), t(sc.R_inv), t(sc.faces.astype(np.float32)), flags[0], flags[1],
                     flags[2], torch.zeros((n_p - n_v, 6)), device=dev, requires_grad=True, n_grid=sc.n_grid,
                     grid_lim=sc.grid_lim)
    if sc.selection is not None:
        state.particle_selection = wp.from_numpy(np.asarray(sc.selection, np.int32), dtype=int)
    model = MPMModelStruct()
    model.init(n_p, device=dev, requires_grad=True)
    model.init_other_params(n_grid=sc.n_grid, grid_lim=sc.grid_lim, device=dev)
    solver = MPMWARP(n_p, n_e, n_v, n_grid=sc.n_grid, grid_lim=sc.grid_lim, mesh_vertices=sc.mesh_vertices,
                     mesh_faces=sc.mesh_faces, num_joint_t=0, num_joint_v=sc.num_joint_v, num_joint_f=sc.num_joint_f,
                     device=dev)
    solver.set_parameters_dict(model, state, sc.params, device=dev)
    state.reset_state(n_v, t(sc.x).clone(), t(sc.d).clone(), None, t(sc.v).clone(), tensor_R_inv=t(sc.R_inv).clone(),
                      device=dev, requires_grad=True)
    ones = torch.ones(n_p, dtype=torch.float32)
    state.reset_density(ones * sc.density, None, dev, update_mass=True)
    solver.set_E_nu_from_torch(model, ones * sc.E, ones * sc.nu, ones * sc.gamma, ones * sc.kappa, dev)
    solver.prepare_mu_lam(model, state, dev)
    if sc.mesh_vertices is not None:
        solver.add_mesh_collider(solver.mesh.id, n_grid=model.n_grid, friction=sc.mesh_friction)
    if (sc.num_joint_v > 0 or sc.num_joint_f > 0) if sc.has_mover is None else sc.has_mover:
        solver.add_particle_mover(n_grid=model.n_grid)
    for kind, kw in sc.bcs:
        {"bounding_box": solver.add_bounding_box, "surface_collider": solver.add_surface_collider,
         "velocity_cuboid": solver.set_velocity_on_cuboid}[kind](**kw)
    sim = RefSim()
    sim.scene, sim.solver, sim.state, sim.model, sim.steps_done = sc, solver, state, model, 0
    sim.pre = []
    return sim


def step_kwargs(sc, step):
    """Arguments of p2g2p at substep `step`, as harness.run / oracle.scene_adapter.run_scene build them."""
    kw = {}
    if sc.mesh_vertices is not None:
        kw["mesh_x"] = torch.from_numpy((sc.mesh_vertices + np.float32(sc.dt * step) * sc.mesh_v).astype(np.float32))
        kw["mesh_v"] = torch.from_numpy(np.ascontiguousarray(sc.mesh_v, np.float32))
    if sc.joint_verts_v is not None:
        kw["joint_verts_v"] = torch.from_numpy(np.ascontiguousarray(sc.joint_verts_v, np.float32))
        kw["joint_faces_v"] = torch.from_numpy(np.ascontiguousarray(sc.joint_faces_v, np.float32).reshape(-1, 3))
    if sc.joint_t_hold > 0:
        kw["joint_traditional_v"] = torch.zeros((sc.joint_t_count(step), 3), dtype=torch.float32)
    return kw


def run_reference(sim, n):
    sc = sim.scene
    for _ in range(n):
        sim.solver.p2g2p(sim.model, sim.state, sc.dt, device="cpu", **step_kwargs(sc, sim.steps_done))
        sim.steps_done += 1


STATE_FIELDS = ("particle_x", "particle_v", "particle_C", "particle_F", "particle_F_trial", "particle_stress",
                "particle_d", "vertex_force", "grid_m", "grid_v_in", "grid_v_out")
MODEL_FIELDS = ("mu", "lam", "yield_stress")


def full_state(sim) -> dict:
    out = {}
    for f in STATE_FIELDS:
        out[f] = getattr(sim.state, f).numpy().copy()
    for f in MODEL_FIELDS:
        out[f] = getattr(sim.model, f).numpy().copy()
    for k, p in enumerate(sim.solver.mesh_collider_params):
        for f in ("weight", "mesh_v_in", "mesh_v_out", "mesh_normal"):
            out[f"col{k}_{f}"] = getattr(p, f).numpy().copy()
    for k, p in enumerate(sim.solver.particle_mover_params):
        for f in ("weight", "velocity"):
            out[f"mov{k}_{f}"] = getattr(p, f).numpy().copy()
    if hasattr(sim.solver, "mesh"):
        out["mesh_points"] = sim.solver.mesh.points.numpy().copy()
        out["mesh_velocities"] = sim.solver.mesh.velocities.numpy().copy()
    return out


def trace_substep(sim):
    """One reference substep with a hook on wp.launch: [(kernel qualname, {array: value after the launch})]."""
    records = []
    real_launch = wp.launch
    before = full_state(sim)

    def hooked(kernel, dim, inputs=(), **kw):
        nonlocal before
        real_launch(kernel, dim, inputs, **kw)
        after = full_state(sim)
        changed = {k: v for k, v in after.items() if not np.array_equal(v, before[k], equal_nan=True)}
        records.append((kernel.func.__qualname__, changed))
        before = after

    wp.launch = hooked
    try:
        run_reference(sim, 1)
    finally:
        wp.launch = real_launch
    return records


# ------------------------------------------------------------------------------------------------ scene <-> npz
SCENE_ARRAYS = ("x", "v", "vol", "faces", "d", "R_inv", "mesh_vertices", "mesh_faces", "mesh_v", "joint_verts_v",
                "joint_faces_v", "selection")
SCENE_SCALARS = ("name", "n_grid", "grid_lim", "n_elements", "n_traditional", "n_vertices", "density", "E", "nu", "gamma",
                 "kappa", "mesh_friction", "num_joint_v", "num_joint_f", "dt", "n_steps", "has_mover", "joint_t_hold",
                 "joint_t_start", "joint_t_every", "joint_t_rate")


def scene_to_dict(sc) -> dict:
    out = {}
    for k in SCENE_ARRAYS:
        v = getattr(sc, k)
        if v is not None:
            out["scene_" + k] = np.asarray(v)
    meta = {k: getattr(sc, k) for k in SCENE_SCALARS}
    meta["params"] = sc.params
    meta["bcs"] = sc.bcs
    out["scene_meta"] = np.array(json.dumps(meta))
    return out


def save(name, payload):
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **payload)
    print(f"wrote {path}: {os.path.getsize(path) / 1024:.0f} KiB", flush=True)


# ------------------------------------------------------------------------------------------------ random states
def _rand_F(rng, n, amp):
    F = np.eye(3)[None] + amp * rng.standard_normal((n, 3, 3))
    assert (np.linalg.det(F) > 0.2).all()
    return F.astype(np.float32)


def trace_scene_traditional(material, seed, n=160):
    """Random blob of traditional particles with every grid-side feature switched on."""
    rng = np.random.default_rng(seed)
    n_grid, grid_lim = 20, 2.0
    pts = (np.array([0.22, 0.35, 0.7]) + rng.uniform(0, 1, (n, 3)) * np.array([1.1, 0.5, 0.6])).astype(np.float32)  # reaches the x = 0 wall's padding
    params = {"material": material, "g": [0.0, -9.8, 0.0], "density": 1.0, "grid_v_damping_scale": 0.97,
              "rpic_damping": 0.15}
    if material == "sand":
        params["friction_angle"] = 35.0
    if material in ("metal", "foam", "plasticine"):
        params.update({"yield_stress": 1.5, "hardening": 1, "xi": 0.2, "plastic_viscosity": 0.4, "softening": 0.15})
    sc = scenes._trad_scene(f"trace-{material}", pts, 0.02 ** 3, n_grid, material=material,
                            v=rng.uniform(-1.0, 1.0, (n, 3)), E=80.0, params=params,
                            bcs=[("bounding_box", {}),
                                 ("surface_collider", {"point": [0.0, 0.45, 0.0], "normal": [0.0, 1.0, 0.0]}),
                                 ("surface_collider", {"point": [0.8, 0.0, 0.0], "normal": [1.0, 0.2, 0.0], "surface": "slip",
                                                       "friction": 0.3}),
                                 ("velocity_cuboid", {"point": [1.0, 0.7, 1.0], "size": [0.12, 0.12, 0.12],
                                                      "velocity": [0.2, 0.1, -0.3]})])
    sc.nu = 0.25
    extra = {"F_trial": _rand_F(rng, n, 0.12), "C": rng.uniform(-3, 3, (n, 3, 3)).astype(np.float32)}
    return sc, extra


def trace_scene_cloth(seed, n_side=9, with_trad=0):
    """Randomly deformed cloth patch over a moving body mesh, joints driven by the mover; directors d perturbed so that
    both sides of every branch of anisotropy_return_mapping (R22 > 1, cone inside / outside) occur."""
    rng = np.random.default_rng(seed)
    n_grid = 20
    verts, faces = garment.grid_sheet(n_side, n_side, 0.7, 1.3, 0.7, 1.3, 1.0)
    rest = verts.copy()
    init_dir, rest_dir, e_vol, v_vol = garment.compute_dir_vol(rest, faces, thickness=1e-5)
    R_inv = garment.compute_rest_dir_inv(rest_dir)
    verts = (verts + rng.uniform(-0.012, 0.012, verts.shape)).astype(np.float32)   # stretched / sheared triangles
    elts = verts[faces].mean(1).astype(np.float32)
    d = init_dir.copy()
    d[:, :, 0] = verts[faces[:, 1]] - verts[faces[:, 0]]
    d[:, :, 1] = verts[faces[:, 2]] - verts[faces[:, 0]]
    nrm = np.cross(d[:, :, 0], d[:, :, 1])
    nrm /= np.linalg.norm(nrm, axis=1, keepdims=True)
    scale = rng.uniform(0.9, 1.1, (faces.shape[0], 1))                       # R22 on both sides of 1
    shear = rng.uniform(-0.08, 0.08, (faces.shape[0], 2))
    shear[::3] *= 0.02                                                      # some inside the friction cone
    d[:, :, 2] = (nrm * scale + d[:, :, 0] / np.linalg.norm(d[:, :, 0], axis=1, keepdims=True) * shear[:, :1]
                  + d[:, :, 1] / np.linalg.norm(d[:, :, 1], axis=1, keepdims=True) * shear[:, 1:]).astype(np.float32)
    n_t = with_trad
    tpts = (np.array([0.8, 1.1, 0.8]) + rng.uniform(0, 0.4, (n_t, 3))).astype(np.float32)
    x = np.concatenate([elts, tpts, verts], 0).astype(np.float32)
    vol = np.concatenate([e_vol, np.full(n_t, 0.02 ** 3, np.float32), v_vol]).astype(np.float32)
    mv, mf = garment.icosphere(1, 0.22, (1.0, 0.85, 1.0))
    mesh_v = (np.array([[0.3, 0.2, -0.1]]) + 0.2 * rng.standard_normal(mv.shape)).astype(np.float32)
    njv, njf = n_side, 4
    jv = rng.uniform(-0.5, 0.5, (njv, 3)).astype(np.float32)
    jf = rng.uniform(-0.5, 0.5, (njf, 3)).astype(np.float32)
    params = {"material": "sand" if n_t else "cloth", "g": [0.0, -9.8, 0.0], "density": 1.0, "grid_v_damping_scale": 1.1,
              "friction_angle": 40.0}
    sc = scenes.Scene(name=f"trace-cloth-{with_trad}", n_grid=n_grid, grid_lim=2.0, n_elements=faces.shape[0], n_traditional=n_t,
                      n_vertices=verts.shape[0], x=x, v=rng.uniform(-0.8, 0.8, x.shape).astype(np.float32), vol=vol,
                      faces=faces.astype(np.int32), d=d.astype(np.float32), R_inv=R_inv, params=params, mesh_vertices=mv,
                      mesh_faces=mf, mesh_v=mesh_v, mesh_friction=0.5, num_joint_v=njv, num_joint_f=njf, joint_verts_v=jv,
                      joint_faces_v=jf, bcs=[("bounding_box", {})], E=60.0, gamma=300.0, kappa=400.0,
                      joint_t_hold=(n_t // 3 if n_t else 0), joint_t_start=0, joint_t_every=1, joint_t_rate=0)
    n_p = x.shape[0]
    extra = {"C": rng.uniform(-2, 2, (n_p, 3, 3)).astype(np.float32)}
    if n_t:
        extra["F_trial"] = _rand_F(rng, faces.shape[0] + n_t, 0.1)
    return sc, extra


def apply_extra(sim, extra):
    """Random C / F_trial written straight into the reference's arrays (its API has no setter for F_trial)."""
    if "C" in extra:
        sim.state.particle_C.numpy()[...] = extra["C"]
    if "F_trial" in extra:
        n = sim.state.particle_F_trial.numpy().shape[0]
        sim.state.particle_F_trial.numpy()[...] = extra["F_trial"][:n]


def add_pre_ops(sim, sc):
    """Pre-p2g particle operations (mpm_solver.py:1058-1328), registered on the reference; described in the fixture."""
    s, st = sim.solver, sim.state
    ops = [("impulse", dict(force=[0.002, 0.0, -0.001], dt=sc.dt, point=[1.0, 0.6, 1.0], size=[0.2, 0.2, 0.2], num_dt=5)),
           ("vel_translation", dict(point=[0.8, 0.5, 0.8], size=[0.1, 0.1, 0.1], velocity=[0.1, 0.2, 0.3], start_time=0.0,
                                    end_time=1.0)),
           ("vel_rotation", dict(point=[1.2, 0.6, 1.2], normal=[0.0, 1.0, 0.0], half_height_and_radius=[0.2, 0.15],
                                 rotation_scale=2.0, translation_scale=0.1, start_time=0.0, end_time=1.0))]
    s.add_impulse_on_particles(st, device="cpu", **ops[0][1])
    s.enforce_particle_velocity_translation(st, device="cpu", **ops[1][1])
    s.enforce_particle_velocity_rotation(st, device="cpu", **ops[2][1])
    return ops


def _trace_payload(sc, extra, with_pre):
    sim = build_reference(sc)
    ops = add_pre_ops(sim, sc) if with_pre else []
    run_reference(sim, 1)          # the traced substep is the SECOND one: grids, vertex_force, collider fields hold leftovers
    apply_extra(sim, extra)
    pre = full_state(sim)
    pre["time"] = np.array(sim.solver.time)
    t0 = time.time()
    records = trace_substep(sim)
    post = full_state(sim)
    print(f"   {sc.name}: {len(records)} launches in {time.time() - t0:.1f} s", flush=True)
    return sim, ops, pre, records, post


def make_trace(name, sc, extra, with_pre=False):
    results = {}
    for svd_mode, qr_mode in (("lapack", "householder"), ("rot", "gs")):
        wp.SVD_MODE, wp.QR_MODE = svd_mode, qr_mode
        results[(svd_mode, qr_mode)] = _trace_payload(sc, extra, with_pre)
    (sim, ops, pre, records, post), (_, _, _, records_b, post_b) = results.values()
    worst = 0.0
    for k in post:
        a, b = post[k].astype(np.float64), post_b[k].astype(np.float64)
        if a.size:
            worst = max(worst, float(np.abs(a - b).max() / max(np.abs(a).max(), 1e-3)))
    assert worst < 2e-6, f"{name}: outputs depend on the svd3/qr3 convention of the stand-in ({worst:.2e})"
    payload = scene_to_dict(sc)
    for k, v in extra.items():
        payload["extra_" + k] = v
    for k, v in pre.items():
        payload["pre_" + k] = v
    payload["launches"] = np.array(json.dumps([q for q, _ in records]))
    for i, (q, changed) in enumerate(records):
        for k, v in changed.items():
            payload[f"L{i:02d}_{k}"] = v
    for k, v in post.items():
        payload["post_" + k] = v
    payload["preops_json"] = np.array(json.dumps(ops))
    payload["convention_spread"] = np.array(worst)
    payload["time_after"] = np.array(sim.solver.time)
    save(name, payload)


def permuted_scene(sc, seed=11):
    """The same cloth scene with the particles enumerated in another order (joint entries stay in front of their class, as the
    mover's launch ranges require): the serial kernels then accumulate their atomic_adds in another order -- rounding-level
    differences in the grid sums, nothing else.  Returns the scene and the map from original to permuted particle index."""
    import copy
    assert sc.n_traditional == 0
    rng = np.random.default_rng(seed)
    n_e, n_v, njf, njv = sc.n_elements, sc.n_vertices, sc.num_joint_f, sc.num_joint_v
    pe = np.concatenate([rng.permutation(njf), njf + rng.permutation(n_e - njf)])      # new element k = old element pe[k]
    pv = np.concatenate([rng.permutation(njv), njv + rng.permutation(n_v - njv)])
    inv_v = np.empty(n_v, np.int64)
    inv_v[pv] = np.arange(n_v)
    out = copy.deepcopy(sc)
    order = np.concatenate([pe, n_e + pv])
    out.x, out.v, out.vol = sc.x[order], sc.v[order], sc.vol[order]
    out.faces = inv_v[sc.faces[pe]].astype(np.int32)
    out.d, out.R_inv = sc.d[pe], sc.R_inv[pe]
    if sc.joint_verts_v is not None:
        out.joint_verts_v = sc.joint_verts_v[pv[:njv]]
        out.joint_faces_v = np.asarray(sc.joint_faces_v).reshape(-1, 3)[pe[:njf]]
    new_of_old = np.empty(n_e + n_v, np.int64)
    new_of_old[order] = np.arange(n_e + n_v)
    return out, new_of_old


def make_seq(name, sc, checkpoints, extra=None, alt=False, permuted=False):
    """alt=True: the same run once more with svd3 / qr3 evaluated in fp32 instead of fp64 (and the other sign conventions),
    stored as alt_s<k>_*: how far the reference is from ITSELF when those two builtins are accurate to fp32 rounding only
    -- as any fp32 implementation, Warp's included, is.  This is the sensitivity envelope of the path (the cloth model's
    R22 = 1 discontinuity, mpm_utils.py:196-204; plastic flow sitting on the yield surface), measured on the reference's
    own source."""
    payload = scene_to_dict(sc)
    for k, v in (extra or {}).items():
        payload["extra_" + k] = v
    for tag, modes in (("", ("lapack", "householder")), ("alt_", ("rot32", "gs32")))[: 2 if alt else 1]:
        wp.SVD_MODE, wp.QR_MODE = modes
        sim = build_reference(sc)
        if extra:
            apply_extra(sim, extra)
        t0 = time.time()
        for cp in checkpoints:
            run_reference(sim, cp - sim.steps_done)
            st = full_state(sim)
            for f in ("particle_x", "particle_v", "particle_C", "particle_F_trial", "particle_d"):
                if tag == "" or f in ("particle_x", "particle_v", "particle_d"):
                    payload[f"{tag}s{cp}_{f}"] = st[f]
            ys = sim.model.yield_stress.numpy()
            print(f"   {name}{' (alt)' if tag else ''}: substep {cp} after {time.time() - t0:.0f} s"
                  + (f"; yield stress now {ys.min():.3f}..{ys.max():.3f}" if ys.max() > 0 else ""), flush=True)
    wp.SVD_MODE, wp.QR_MODE = "lapack", "householder"
    if permuted:  # second envelope sample: same builtins, other particle order (stored in the ORIGINAL order as alt2_s<k>_*)
        psc, new_of_old = permuted_scene(sc)
        sim = build_reference(psc)
        for cp in checkpoints:
            run_reference(sim, cp - sim.steps_done)
            st = full_state(sim)
            for f in ("particle_x", "particle_v"):
                payload[f"alt2_s{cp}_{f}"] = st[f][new_of_old]
            print(f"   {name} (permuted): substep {cp}", flush=True)
    payload["checkpoints"] = np.array(checkpoints)
    save(name, payload)


def _mat_params(material):
    params = {"friction_angle": 40.0} if material == "sand" else {}
    if material in ("metal", "foam", "plasticine"):
        params.update({"yield_stress": 0.5, "hardening": 1, "xi": 0.1, "plastic_viscosity": 0.5})
    return params


def _seq_cube(material):
    """Spinning cube that is also being stretched / squeezed / sheared at a few 1/s, so that within 100 substeps the plastic
    materials yield (and harden / soften) and the sand takes all three branches of its return mapping."""
    sc = scenes.small_cube(n=6, n_grid=24, material=material, params=_mat_params(material))
    S = np.array([[4.0, 1.0, 0.0], [1.0, -3.0, 0.5], [0.0, 0.5, -2.0]], np.float32)
    sc.v = (sc.v + (sc.x - sc.x.mean(0)) @ S.T).astype(np.float32)
    return sc


def _small_sheet(**kw):
    sc = scenes.sheet(n=14, n_grid=24, collider_subdiv=2, span=(0.6, 1.4), y=1.22, sphere_r=0.2, sphere_c=(1.0, 0.98, 1.0),
                      name="sheet-14x14")
    for k, v in kw.items():
        setattr(sc, k, v)
    return sc


def _small_garment(**kw):
    sc = scenes.garment_cylinder(n_theta=20, n_h=12, n_grid=24, aniso=True, collider_subdiv=1, name="garment-20x12")
    for k, v in kw.items():
        setattr(sc, k, v)
    return sc


def make_cov_from_F(name="ref_cov_from_F"):
    """MPMWARP.export_particle_cov_to_torch (mpm_solver.py:543-561 -> compute_cov_from_F, mpm_utils.py:1108-1132) on the
    reference: a strained jelly blob after three substeps (F_trial = (I + dt grad v) F is no longer the random start) with a
    random symmetric particle_cov."""
    sc, extra = trace_scene_traditional("jelly", 321, n=96)
    sim = build_reference(sc)
    apply_extra(sim, extra)
    run_reference(sim, 3)
    rng = np.random.default_rng(322)
    n = sc.n_particles - sc.n_vertices
    cov0 = rng.uniform(-1.0, 1.0, n * 6).astype(np.float32)
    sim.state.particle_cov = wp.from_numpy(cov0, dtype=float)
    F_trial = np.array(sim.state.particle_F_trial.numpy(), np.float32).reshape(n, 3, 3)
    new_cov = sim.solver.export_particle_cov_to_torch(sim.state, device="cpu").numpy().astype(np.float32)
    # cross-check in float64 (the fixture is the reference's output; this only guards the stand-in's mat33 product)
    S = np.zeros((n, 3, 3)); c = cov0.reshape(n, 6).astype(np.float64)
    S[:, 0, 0], S[:, 0, 1], S[:, 0, 2], S[:, 1, 1], S[:, 1, 2], S[:, 2, 2] = c.T
    S = S + np.triu(S, 1).transpose(0, 2, 1)
    want = np.einsum("nij,njk,nlk->nil", F_trial.astype(np.float64), S, F_trial.astype(np.float64))
    want6 = np.stack([want[:, 0, 0], want[:, 0, 1], want[:, 0, 2], want[:, 1, 1], want[:, 1, 2], want[:, 2, 2]], 1).reshape(-1)
    assert np.abs(new_cov - want6).max() < 1e-5 * max(np.abs(want6).max(), 1.0)
    save(name, {"particle_F_trial": F_trial, "particle_cov": cov0, "new_cov": new_cov})


FIXTURES = {
    "ref_cov_from_F": make_cov_from_F,
    # --- per-kernel traces from random states
    **{f"ref_trace_{m}": (lambda m=m, i=i: make_trace(f"ref_trace_{m}", *trace_scene_traditional(m, 100 + i), with_pre=(m == "jelly")))
       for i, m in enumerate(["jelly", "metal", "sand", "foam", "snow", "plasticine"])},
    "ref_trace_cloth": lambda: make_trace("ref_trace_cloth", *trace_scene_cloth(7)),
    "ref_trace_mixed": lambda: make_trace("ref_trace_mixed", *trace_scene_cloth(8, n_side=7, with_trad=90)),
    # --- whole-substep sequences of the small test scenes
    **{f"ref_seq_cube_{m}": (lambda m=m: make_seq(f"ref_seq_cube_{m}", _seq_cube(m), [1, 10, 50, 100], alt=True))
       for m in ["jelly", "sand", "metal", "foam", "plasticine"]},
    "ref_seq_sheet": lambda: make_seq("ref_seq_sheet", _small_sheet(), [1, 5, 20, 40, 80], alt=True, permuted=True),
    "ref_seq_sheet_gamma0": lambda: make_seq("ref_seq_sheet_gamma0", _small_sheet(gamma=0.0), [1, 10, 50, 100, 200], alt=True, permuted=True),
    "ref_seq_garment": lambda: make_seq("ref_seq_garment", _small_garment(), [1, 5, 20, 40, 80], alt=True, permuted=True),
    "ref_seq_garment_gamma0": lambda: make_seq("ref_seq_garment_gamma0", _small_garment(gamma=0.0), [1, 10, 50, 100], alt=True, permuted=True),
    "ref_seq_demo": lambda: make_seq("ref_seq_demo", scenes.demo_mix(n_grid=24, n_sheet=10, sand=(10, 3, 6), hold=(8, 4, 40)),
                                     [1, 5, 20, 40], alt=True),
}


def make_alt3(name, sc, checkpoints, extra=None):
    """Round 4: the same sequence a THIRD time, with svd3 / qr3 evaluated by the published algorithm behind
    Warp's builtins -- McAdams et al.'s fp32 Jacobi SVD with approximate Givens quaternions and the Givens-quaternion QR
    (warp_standin: SVD_MODE "mcadams", QR_MODE "givens") -- stored in its own small file alt3_<name>.npz as alt3_s<k>_*.
    tests/test_ref_golden.py reports its distance from the primary run beside the other two envelopes and takes it into the
    bound (refgolden.seq_bound)."""
    wp.SVD_MODE, wp.QR_MODE = "mcadams", "givens"
    payload = {}
    sim = build_reference(sc)
    if extra:
        apply_extra(sim, extra)
    t0 = time.time()
    for cp in checkpoints:
        run_reference(sim, cp - sim.steps_done)
        st = full_state(sim)
        for f in ("particle_x", "particle_v", "particle_d"):
            payload[f"alt3_s{cp}_{f}"] = st[f]
        print(f"   {name} (alt3: mcadams / givens): substep {cp} after {time.time() - t0:.0f} s", flush=True)
    wp.SVD_MODE, wp.QR_MODE = "lapack", "householder"
    payload["checkpoints"] = np.array(checkpoints)
    save("alt3_" + name, payload)


ALT3 = {
    "ref_seq_cube_jelly": lambda: make_alt3("ref_seq_cube_jelly", _seq_cube("jelly"), [1, 10, 50, 100]),
    "ref_seq_cube_sand": lambda: make_alt3("ref_seq_cube_sand", _seq_cube("sand"), [1, 10, 50, 100]),
    "ref_seq_cube_metal": lambda: make_alt3("ref_seq_cube_metal", _seq_cube("metal"), [1, 10, 50, 100]),
    "ref_seq_sheet": lambda: make_alt3("ref_seq_sheet", _small_sheet(), [1, 5, 20, 40, 80]),
    "ref_seq_garment": lambda: make_alt3("ref_seq_garment", _small_garment(), [1, 5, 20, 40, 80]),
    "ref_seq_sheet_gamma0": lambda: make_alt3("ref_seq_sheet_gamma0", _small_sheet(gamma=0.0), [1, 10, 50, 100, 200]),
}


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "--alt3":
        for n in sys.argv[2:] or list(ALT3):
            print(f"== {n} (alt3)", flush=True)
            ALT3[n]()
        sys.exit(0)
    names = sys.argv[1:] or list(FIXTURES)
    for n in names:
        print(f"== {n}", flush=True)
        FIXTURES[n]()
