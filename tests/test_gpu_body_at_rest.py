"""A body that does not move during an `mpmhip_steps` call is splatted once per accumulator buffer, not once per substep (round 6:
fast_body_at_rest_begin, csrc/fast.hip; the reference's compute_mesh kernels, mpm_solver.py:829-880, run every substep whatever the body
does).  The collider field must be the same one, so: the run with kept fields = the run that splats every substep (MPMHIP_COL_KEEP=0)
= the oracle, across everything that has to drop the kept fields -- re-sorts inside a call, a new pose between two calls, a body that
starts to move, single `p2g2p` calls in between -- and the kept path must really have run."""
import os

import numpy as np
import pytest
import torch

from mpmavatar_amd import harness, scenes

pytestmark = pytest.mark.gpu

rel = lambda a, b: float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-3))


def _scene():
    sc = scenes.small_sheet()
    sc.gamma = 0.0                      # (no shear term: the 1e-4 bound on v holds without an envelope)
    sc.v = sc.v.copy()
    sc.v[:, 1] = -2.0                   # thrown at the sphere: contact within the run
    return sc


# (substeps, body velocity during the call, pose shift applied BEFORE the call, fused call?)
PLAN = [(40, 0.0, 0.0, True), (40, 0.0, 0.0, True), (3, 0.0, 0.0, False), (40, 0.0, 0.01, True), (40, 0.3, 0.0, True), (40, 0.0, 0.0, True)]


def _drive(sim, o, sc, rebin_note=None):
    dev = sim.solver.device
    pose = sc.mesh_vertices.astype(np.float32).copy()
    for n, vy, shift, fused in PLAN:
        pose = pose + np.float32(shift) * np.array([[1.0, 0.0, 0.0]], np.float32)
        mv = np.zeros_like(pose)
        mv[:, 1] = vy
        mx_t, mv_t = torch.as_tensor(pose, device=dev), torch.as_tensor(mv, device=dev)
        if fused:
            sim.solver.p2g2p_n(sim.model, sim.state, sc.dt, n, mesh_x=mx_t, mesh_v=mv_t)
        else:
            for k in range(n):
                sim.solver.p2g2p(sim.model, sim.state, sc.dt, mesh_x=mx_t + np.float32(sc.dt * k) * mv_t, mesh_v=mv_t)
        if o is not None:
            for k in range(n):
                o.p2g2p(sc.dt, mesh_x=(pose + np.float32(sc.dt * k) * mv).astype(np.float32), mesh_v=mv)
        pose = (pose + np.float32(sc.dt * n) * mv).astype(np.float32)     # the body stays where the call left it


@pytest.mark.parametrize("rebin", [0, -7])
def test_kept_collider_field_equals_per_substep_splat_equals_oracle(rebin, oracle_lib):
    from oracle.scene_adapter import oracle_from_scene
    sc = _scene()
    o = oracle_from_scene(sc)
    kept = harness.build_solver(_scene(), "cuda:0", mode="fast", rebin_interval=rebin)
    _drive(kept, o, sc)
    os.environ["MPMHIP_COL_KEEP"] = "0"
    try:
        every = harness.build_solver(_scene(), "cuda:0", mode="fast", rebin_interval=rebin)
    finally:
        del os.environ["MPMHIP_COL_KEEP"]
    _drive(every, None, sc)
    sk, se = kept.solver.stats(), every.solver.stats()
    n_rest = sum(n for n, vy, _, fused in PLAN if fused and vy == 0.0)
    assert se["kept_collider_substeps"] == 0 and sk["n_dropped"] == 0 and se["n_dropped"] == 0
    # two splats per call (one per buffer), two more after every re-sort inside a call; everything else ran on kept fields
    assert 0 < sk["kept_collider_substeps"] <= n_rest - 2 * 4, sk
    if rebin == 0:
        assert sk["kept_collider_substeps"] >= n_rest - 2 * 4 - 2 * (sk["rebins"] + 1), sk
    xk, vk = kept.state.particle_x.cpu().numpy(), kept.state.particle_v.cpu().numpy()
    xe, ve = every.state.particle_x.cpu().numpy(), every.state.particle_v.cpu().numpy()
    assert rel(xk, xe) < 1e-6 and rel(vk, ve) < 1e-5, (rel(xk, xe), rel(vk, ve))
    # the sheet hit the sphere (the collider did something): its fastest particles were slowed by more than 100 tolerances
    assert np.abs(o.v[:, 1]).min() < 0.9 * np.abs(o.v[:, 1]).max()
    assert rel(xk, o.x) < 1e-5 and rel(vk, o.v) < 1e-4, (rel(xk, o.x), rel(vk, o.v))
