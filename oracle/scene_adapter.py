"""Build the CPU oracle from a ``mpmavatar_amd.scenes.Scene`` (TEST INFRASTRUCTURE ONLY).

Follows the reference driver's setup sequence (/root/reference/train_material_params.py:403-506):
state.from_torch -> model.init_other_params -> set_parameters_dict -> reset_state -> reset_density(update_mass)
-> set_E_nu -> prepare_mu_lam -> add_mesh_collider -> add_particle_mover -> BCs.
"""
from __future__ import annotations

import numpy as np

from .oracle import OracleMPM


def omp_threads() -> int:
    """Thread count for the OpenMP build.  NOT os.cpu_count(): measured on the MI355X box (256 logical CPUs visible), the oracle
    runs 57 substeps/s on the S3 garment with 16 threads and 0.67 with 256 (18 vs 0.9 on the headline sheet) -- its parallel
    regions are short and atomics-heavy, and oversubscribing the cores the container really has costs two orders of magnitude
    (tools/gpu/oracle_threads.py, profiles/r03_oracle_threads.txt).  ORACLE_THREADS overrides."""
    import os
    if os.environ.get("ORACLE_THREADS"):
        return max(1, int(os.environ["ORACLE_THREADS"]))
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    return max(1, min(16, avail))


def oracle_from_scene(sc, omp=False, n_threads=1, fma=False) -> OracleMPM:
    o = OracleMPM(sc.n_particles, sc.n_elements, sc.n_vertices, n_grid=sc.n_grid, grid_lim=sc.grid_lim,
                  mesh_vertices=sc.mesh_vertices, mesh_faces=sc.mesh_faces, num_joint_v=sc.num_joint_v,
                  num_joint_f=sc.num_joint_f, omp=omp, n_threads=n_threads, fma=fma)
    o.x[:] = sc.x
    o.v[:] = sc.v
    o.vol[:] = sc.vol
    o.faces[:] = sc.faces.astype(np.float32)  # float-encoded indices, mpm_data_structure.py:211-215
    o.d[:] = sc.d
    o.R_inv[:] = sc.R_inv
    if getattr(sc, "selection", None) is not None:
        o.selection[:] = np.where(np.asarray(sc.selection) == 0, 0, 1)  # only selection == 0 is simulated (mpm_data_structure.py:39)
    o.set_parameters_dict(sc.params)
    o.reset_state()
    o.density[:] = sc.density
    o.mass[:] = o.density * o.vol
    o.E[:], o.nu[:], o.gamma[:], o.kappa[:] = sc.E, sc.nu, sc.gamma, sc.kappa
    o.prepare_mu_lam()
    if sc.mesh_vertices is not None:
        o.add_mesh_collider(friction=sc.mesh_friction)
    if sc.num_joint_v > 0 or sc.num_joint_f > 0:
        o.add_particle_mover()
    for kind, kw in sc.bcs:
        getattr(o, {"bounding_box": "add_bounding_box", "surface_collider": "add_surface_collider",
                    "velocity_cuboid": "set_velocity_on_cuboid"}[kind])(**kw)
    return o


def run_scene(sim, sc, n_steps=None, step_fn=None, k0=0):
    """Drive ``sim`` (oracle, twin, or anything with p2g2p/step) through the scene's substeps with the
    reference's mesh advection mesh_x + k*dt*mesh_v (train_material_params.py:622-626)."""
    n = sc.n_steps if n_steps is None else n_steps
    fn = step_fn or (sim.p2g2p if hasattr(sim, "p2g2p") else sim.step)
    for k in range(n):
        kw = {}
        if sc.mesh_vertices is not None:
            kw["mesh_x"], kw["mesh_v"] = sc.body_at(k0 + k) if hasattr(sc, "body_at") else (
                (sc.mesh_vertices + np.float32(sc.dt * (k0 + k)) * sc.mesh_v).astype(np.float32), sc.mesh_v)
        if sc.joint_verts_v is not None:
            kw["joint_verts_v"], kw["joint_faces_v"] = sc.joints_at(k0 + k) if hasattr(sc, "joints_at") else (sc.joint_verts_v, sc.joint_faces_v)
        if getattr(sc, "joint_t_hold", 0) > 0:  # staged sand release, run_demo.py:524
            kw["joint_traditional_v"] = np.zeros((sc.joint_t_count(k0 + k), 3), np.float32)
        fn(sc.dt, **kw)
