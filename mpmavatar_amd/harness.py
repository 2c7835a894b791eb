"""Drive the HIP solver through a synthetic ``Scene`` exactly the way the reference drivers drive
``warp_mpm``: the setup sequence of /root/reference/train_material_params.py:403-506 and the substep
loop of :616-631 (``mesh_x + k*dt*mesh_v`` advection, ``particle_x`` read back per frame).
"""
from __future__ import annotations

from dataclasses import dataclass

import numpy as np
import torch

from .scenes import Scene
from .warp_mpm import MPMWARP, MPMModelStruct, MPMStateStruct


@dataclass
class Sim:
    scene: Scene
    solver: MPMWARP
    state: MPMStateStruct
    model: MPMModelStruct
    mesh_x0: torch.Tensor = None
    mesh_v: torch.Tensor = None
    joint_verts_v: torch.Tensor = None
    joint_faces_v: torch.Tensor = None
    steps_done: int = 0
    mesh_static: bool = False


def build_solver(sc: Scene, device="cuda:0", mode=None, rebin_interval=0, p2g_tile="auto") -> Sim:
    dev = torch.device(device)
    t = lambda a, dt=torch.float32: torch.as_tensor(np.ascontiguousarray(a), dtype=dt, device=dev)
    n_p, n_e, n_v, n_t = sc.n_particles, sc.n_elements, sc.n_vertices, sc.n_traditional
    state = MPMStateStruct()
    state.init(n_p, n_e, n_v, device=dev, requires_grad=True)
    flags = np.zeros((3, n_p), np.int32)
    flags[0, n_e:n_e + n_t] = 1   # traditional
    flags[1, n_e + n_t:] = 1      # vertices
    flags[2, :n_e] = 1            # elements
    D_inv = np.linalg.inv(sc.d.astype(np.float64)).astype(np.float32) if n_e else np.zeros((0, 3, 3), np.float32)
    state.from_torch(t(sc.x), t(sc.vol), t(D_inv), t(sc.R_inv), t(sc.faces.astype(np.float32)), flags[0], flags[1],
                     flags[2], torch.zeros((n_p - n_v, 6)), device=dev, requires_grad=True, n_grid=sc.n_grid,
                     grid_lim=sc.grid_lim)
    if sc.selection is not None:
        state.particle_selection = t(sc.selection, torch.int32)
    model = MPMModelStruct()
    model.init(n_p, device=dev, requires_grad=True)
    model.init_other_params(n_grid=sc.n_grid, grid_lim=sc.grid_lim, device=dev)
    solver = MPMWARP(n_p, n_e, n_v, n_grid=sc.n_grid, grid_lim=sc.grid_lim, mesh_vertices=sc.mesh_vertices,
                     mesh_faces=sc.mesh_faces, num_joint_t=0, num_joint_v=sc.num_joint_v, num_joint_f=sc.num_joint_f,
                     device=dev, mode=mode, rebin_interval=rebin_interval, p2g_tile=p2g_tile)
    solver.set_parameters_dict(model, state, sc.params)
    state.reset_state(n_v, t(sc.x).clone(), t(sc.d).clone(), None, t(sc.v).clone(), tensor_R_inv=t(sc.R_inv).clone(),
                      device=dev, requires_grad=True)
    ones = torch.ones(n_p, dtype=torch.float32, device=dev)
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
    sim = Sim(sc, solver, state, model)
    if sc.mesh_vertices is not None:
        sim.mesh_x0, sim.mesh_v = t(sc.mesh_vertices), t(sc.mesh_v)
        sim.mesh_static = not np.any(np.asarray(sc.mesh_v))
    if sc.joint_verts_v is not None:
        sim.joint_verts_v, sim.joint_faces_v = t(sc.joint_verts_v), t(sc.joint_faces_v).reshape(-1, 3)
    return sim


def run(sim: Sim, n_steps: int, fused: bool = False):
    """Advance ``n_steps`` substeps.  fused=False issues one ``p2g2p`` per substep with the mesh advected
    on the torch side like the reference loop; fused=True hands runs of substeps to ``mpmhip_steps`` (split where the
    scene's staged sand release changes the length of ``joint_traditional_v``)."""
    sc, sv = sim.scene, sim.solver
    dev = sv.device  # (not via sim.state.particle_x: reading a state field pulls it back and forces a re-import)

    t = lambda a: None if a is None else torch.as_tensor(np.ascontiguousarray(a, np.float32), device=dev).reshape(-1, 3)

    def kwargs(step):
        n_jt = sc.joint_t_count(step)
        jt = torch.zeros((n_jt, 3), dtype=torch.float32, device=dev) if sc.joint_t_hold > 0 else None
        if sc.mesh_sway is not None and sc.joint_verts_v is not None:   # joints ride on the swaying body
            jv, jf = sc.joints_at(step)
            return dict(joint_traditional_v=jt, joint_verts_v=t(jv), joint_faces_v=t(jf))
        return dict(joint_traditional_v=jt, joint_verts_v=sim.joint_verts_v, joint_faces_v=sim.joint_faces_v)

    def mesh(step):
        if sim.mesh_x0 is None:
            return None, None
        if sc.mesh_sway is not None:
            mx, mv = sc.body_at(step)
            return t(mx), t(mv)
        if sim.mesh_static:   # a body at rest: no advection kernel, no temporary per call (the caller's mesh_x + k dt mesh_v with mesh_v = 0)
            return sim.mesh_x0, sim.mesh_v
        return sim.mesh_x0 + np.float32(sc.dt * step) * sim.mesh_v, sim.mesh_v

    end = sim.steps_done + n_steps
    while sim.steps_done < end:
        k0 = sim.steps_done
        n = 1
        if fused:
            n = end - k0
            if sc.joint_t_hold > 0:  # stop the fused run at the next change of the held count
                c0 = sc.joint_t_count(k0)
                n = next((j for j in range(1, n) if sc.joint_t_count(k0 + j) != c0), n)
            f0, spf = sc.frame_of(k0)
            if spf is not None:      # ... and at the next pose of a swaying body (the fused call advects x + k dt v)
                n = min(n, f0 + spf - k0)
            mx, mv = mesh(k0)
            sv.p2g2p_n(sim.model, sim.state, sc.dt, n, mesh_x=mx, mesh_v=mv, **kwargs(k0))
        else:
            mx, mv = mesh(k0)
            sv.p2g2p(sim.model, sim.state, sc.dt, mesh_x=mx, mesh_v=mv, **kwargs(k0))
        sim.steps_done += n


def algorithmic_bytes(sc: Scene, n_active=0, n_collider=0, n_mover=0) -> dict:
    """B_alg = 516 n_e + 160 n_v + 364 n_t + 56 N_active + 68 N_coll + 32 N_mov and the
    g2p-only figure B_g2p = 228 n_e + 72 n_v + 144 n_t + 12 N_active."""
    b = 516 * sc.n_elements + 160 * sc.n_vertices + 364 * sc.n_traditional + 56 * n_active + 68 * n_collider + 32 * n_mover
    g = 228 * sc.n_elements + 72 * sc.n_vertices + 144 * sc.n_traditional + 12 * n_active
    return {"substep": int(b), "g2p": int(g)}
