"""Known-answer tests that pin the CPU oracle.

The reference ships no tests or golden vectors for this path and its Warp kernels cannot run here; beside the fixtures its own kernel
bodies produced over a NumPy stand-in of the warp module (tests/test_ref_golden.py), these analytic properties anchor the restatement in
oracle/mpm_oracle.c.  Each test cites the reference
lines whose behaviour it checks.
"""
import math

import numpy as np
import pytest

from oracle import oracle as O
from oracle.oracle import OracleMPM

DT = 1e-4


def make(n_trad=0, n_grid=16, grid_lim=2.0, **kw):
    return OracleMPM(n_trad + kw.get("n_elements", 0) + kw.get("n_vertices", 0), kw.get("n_elements", 0),
                     kw.get("n_vertices", 0), n_grid=n_grid, grid_lim=grid_lim,
                     mesh_vertices=kw.get("mesh_vertices"), mesh_faces=kw.get("mesh_faces"),
                     num_joint_v=kw.get("num_joint_v", 0), num_joint_f=kw.get("num_joint_f", 0))


def trad_block(n=5, n_grid=16, seed=0, material="jelly", E=100.0):
    rng = np.random.default_rng(seed)
    o = make(n ** 3, n_grid=n_grid)
    dx = 2.0 / n_grid
    idx = np.stack(np.meshgrid(*[np.arange(n)] * 3, indexing="ij"), -1).reshape(-1, 3)
    o.x[:] = 0.8 + (idx + rng.uniform(0.2, 0.8, idx.shape)) * (dx / 2)
    o.vol[:] = (dx / 2) ** 3
    o.set_parameters_dict({"material": material, "density": 2.0})
    o.reset_state()
    o.E[:], o.nu[:] = E, 0.3
    o.prepare_mu_lam()
    return o


# ---- K1: B-spline stencil (mpm_utils.py:499-526) -------------------------------------------------------------
def test_k1_stencil_partition_of_unity_and_moments():
    rng = np.random.default_rng(1)
    inv_dx = 8.0
    for _ in range(200):
        x = rng.uniform(0.3, 1.7, 3).astype(np.float32)
        base, w, dw = O.stencil(x, inv_dx)
        fx = x * inv_dx - base
        assert np.all(base == np.trunc(x * inv_dx - 0.5).astype(int))
        assert np.all((fx >= 0.5 - 1e-6) & (fx < 1.5 + 1e-6))
        np.testing.assert_allclose(w.sum(1), 1.0, atol=2e-6)      # sum_i w_i = 1 per axis
        np.testing.assert_allclose(dw.sum(1), 0.0, atol=2e-6)     # sum_i dw_i = 0
        nodes = np.arange(3)[None, :] - fx[:, None]               # dpos / dx
        np.testing.assert_allclose((w * nodes).sum(1), 0.0, atol=2e-6)          # first moment vanishes
        np.testing.assert_allclose((w * nodes ** 2).sum(1), 0.25, atol=2e-6)    # => APIC factor 4 inv_dx^2


# ---- K2/K3: uniform and affine velocity fields round-trip (p2g :484-557, grid :561-572, g2p :716-786) -------
@pytest.mark.parametrize("A", [np.zeros((3, 3)), np.array([[0.0, -0.3, 0.1], [0.3, 0.0, 0.2], [-0.1, -0.2, 0.0]])])
def test_k2_k3_affine_roundtrip(A):
    o = trad_block(n=6)
    v0 = np.array([0.3, -0.2, 0.1])
    c = o.x.mean(0)
    o.v[:] = v0 + (o.x - c) @ A.T
    o.C[:] = A
    o.E[:] = 0.0
    o.prepare_mu_lam()       # no stress: pure transfer test
    x0 = o.x.copy()
    o.zero_grid(); o.compute_stress(DT); o.p2g(DT); o.grid_update(DT); o.g2p_v(DT)
    # interior particles (whose whole stencil is covered by the block) recover v and C exactly (APIC)
    inner = np.all(np.abs(x0 - c) < 0.12, axis=1)
    assert inner.sum() > 8
    np.testing.assert_allclose(o.v[inner], (v0 + (x0 - c) @ A.T)[inner], atol=2e-5)
    np.testing.assert_allclose(o.C[inner], np.broadcast_to(A, (inner.sum(), 3, 3)), atol=2e-3)
    np.testing.assert_allclose(o.x, x0 + DT * o.v, atol=1e-6)


# ---- K4: conservation through p2g ---------------------------------------------------------------------------
def test_k4_mass_and_momentum_conservation():
    o = trad_block(n=5, seed=3)
    rng = np.random.default_rng(3)
    o.v[:] = rng.normal(size=o.v.shape) * 0.2
    o.F_trial[:] = np.eye(3) + rng.normal(size=o.F_trial.shape) * 0.02
    o.zero_grid(); o.compute_stress(DT); o.p2g(DT)
    np.testing.assert_allclose(o.grid_m.sum(), o.mass.sum(), rtol=1e-5)
    # sum of internal forces is zero (sum_i dweight_i = 0), C = 0: grid momentum = particle momentum
    np.testing.assert_allclose(o.grid_v_in.sum(0), (o.mass[:, None] * o.v).sum(0), rtol=2e-4, atol=1e-7)


# ---- K5: free fall -------------------------------------------------------------------------------------------
def test_k5_free_fall():
    o = trad_block(n=3)
    o.set_parameters_dict({"g": [0.0, -9.8, 0.0]})
    o.E[:] = 0.0
    o.prepare_mu_lam()
    x0 = o.x.copy()
    n = 20
    for _ in range(n):
        o.p2g2p(DT)
    np.testing.assert_allclose(o.v[:, 1], -9.8 * DT * n, rtol=1e-4)
    # symplectic Euler: x_n = x_0 + dt * sum_{k=1..n} k dt g
    np.testing.assert_allclose(o.x[:, 1], x0[:, 1] - 9.8 * DT * DT * n * (n + 1) / 2, atol=1e-6)
    assert abs(o.time - n * np.float32(DT)) < 1e-9


# ---- K6/K7: cloth constitutive model (mpm_utils.py:101-209) --------------------------------------------------
def _rot(axis, ang):
    axis = np.asarray(axis, float) / np.linalg.norm(axis)
    K = np.array([[0, -axis[2], axis[1]], [axis[2], 0, -axis[0]], [-axis[1], axis[0], 0]])
    return np.eye(3) + math.sin(ang) * K + (1 - math.cos(ang)) * K @ K


def test_k6_rigid_rotation_gives_zero_cloth_stress():
    D = np.array([[0.02, 0.005, 0.0], [0.0, 0.018, 0.0], [0.0, 0.0, 1.0]])   # columns d1, d2, unit normal
    R_inv = np.array([1 / D[0, 0], -D[0, 1] / (D[0, 0] * D[1, 1]), 1 / D[1, 1]], np.float32)
    for ang in (0.0, 0.7, 2.5):
        d = _rot([1, 2, 3], ang) @ D
        S, f1, f2, f3 = O.kirchhoff_anisotropy(R_inv, d, 1e-8, 40.0, 60.0, 500.0, 500.0)
        assert np.abs(S).max() < 1e-9 and max(np.abs(f1).max(), np.abs(f2).max(), np.abs(f3).max()) < 1e-9


def test_k7_uniaxial_stretch_closed_form():
    L, s, vol, mu, lam = 0.02, 1.1, 3e-9, 40.0, 60.0
    R_inv = np.array([1 / L, 0.0, 1 / L], np.float32)
    d = np.array([[s * L, 0, 0], [0, L, 0], [0, 0, 1.0]]).astype(np.float32)
    S, f1, f2, f3 = O.kirchhoff_anisotropy(R_inv, d, vol, mu, lam, 500.0, 500.0)
    # F2 = diag(s,1), Rot = I, J = s: K2 = 2mu(F2-I) + lam(J-1) [[F22,0],[-F12,F11]] = diag(2mu(s-1)+lam(s-1), lam(s-1)s)
    k11 = 2 * mu * (s - 1) + lam * (s - 1) * 1.0
    k22 = lam * (s - 1) * s
    # P = Q K3sym RiDT^-1 with Q = I, RiDT = diag(s,1,1): K3 = diag(k11*s, k22, 0) => P = diag(k11, k22, 0)
    np.testing.assert_allclose(f2, [-vol * k11 / L, 0, 0], rtol=2e-4, atol=1e-12)
    np.testing.assert_allclose(f3, [0, -vol * k22 / L, 0], rtol=2e-4, atol=1e-12)
    np.testing.assert_allclose(f1, -(f2 + f3), atol=1e-12)
    assert np.abs(S).max() < 1e-9          # P3 = 0: no stress carried by the director


def test_k8_cloth_return_mapping_branches():
    Q = _rot([0.3, 1, 0.2], 0.4)
    def mapped(r02, r12, r22, gamma=500.0, kappa=500.0, fc=math.tan(math.radians(40))):
        R = np.array([[0.02, 0.004, r02], [0, 0.018, r12], [0, 0, r22]])
        nd = O.anisotropy_return_mapping(Q @ R, gamma, kappa, fc)
        return (Q.T @ nd)[:, 2], (Q.T @ nd)[:, :2], R
    # R22 > 1: clamp to 1, shear kept (mpm_utils.py:196-197)
    r3, d12, R = mapped(0.01, -0.02, 1.05)
    np.testing.assert_allclose(r3, [0.01, -0.02, 1.0], atol=1e-6)
    np.testing.assert_allclose(d12, R[:, :2], atol=1e-7)          # d1, d2 untouched
    # compressed, inside the friction cone: unchanged (:203-204)
    fn = 500.0 * 0.1 ** 2
    sh = 0.5 * math.tan(math.radians(40)) * fn / 500.0
    r3, _, _ = mapped(sh, 0.0, 0.9)
    np.testing.assert_allclose(r3, [sh, 0.0, 0.9], atol=1e-6)
    # outside the cone: shear scaled onto it (:201-202)
    r3, _, _ = mapped(0.3, 0.4, 0.9)
    lim = math.tan(math.radians(40)) * fn / 500.0
    np.testing.assert_allclose(np.hypot(r3[0], r3[1]), lim, rtol=1e-4)
    np.testing.assert_allclose(r3[1] / r3[0], 0.4 / 0.3, rtol=1e-4)


def test_k8_sand_return_mapping_cases():
    """sand_return_mapping (mpm_utils.py:362-399) through the stress kernel: expansion -> F = U V^T."""
    o = trad_block(n=2, material="sand")
    o.set_parameters_dict({"friction_angle": 40.0})
    Rm = _rot([0, 0, 1], 0.3).astype(np.float32)
    o.F_trial[:] = Rm * 1.05            # pure dilation, tr(eps) > 0 -> projected onto the rotation
    o.compute_stress(DT)
    np.testing.assert_allclose(o.F, np.broadcast_to(Rm, o.F.shape), atol=2e-6)
    o.F_trial[:] = Rm @ np.diag([0.98, 0.98, 0.98]).astype(np.float32)   # isotropic compression: delta_gamma <= 0 -> elastic
    o.compute_stress(DT)
    np.testing.assert_allclose(o.F, o.F_trial, atol=1e-7)


# ---- K9: grid colliders (mpm_solver.py:564-658, 790-799, 882-917) --------------------------------------------
def _one_node_collider(v_node, mesh_v, friction):
    # one big triangle whose centroid sits on a grid node region; normal +y
    mv = np.array([[0.9, 1.0, 0.9], [1.1, 1.0, 1.3], [1.3, 1.0, 0.9]], np.float32)
    o = make(0, n_grid=16, mesh_vertices=mv, mesh_faces=np.array([[0, 1, 2]], np.int32))
    o.add_mesh_collider(friction=friction)
    o.mesh_velocities[:] = mesh_v
    o.grid_v_out[:] = v_node
    o.mesh_collide(0)
    touched = o.mesh_colliders[0]["weight"] > 1e-15
    assert touched.sum() == 27
    return o.grid_v_out[touched], o.grid_v_out[~touched], o


def test_k9_mesh_collider_projection_and_friction():
    n_up = np.array([0.0, 1.0, 0.0])
    # approaching (v.n < 0): normal component removed, tangential speed reduced by mu |vn|
    vt, vu, o = _one_node_collider([0.4, -0.5, 0.0], 0.0, 0.5)
    np.testing.assert_allclose(vt, np.broadcast_to([0.4 - 0.5 * 0.5, 0.0, 0.0], vt.shape), atol=1e-6)
    np.testing.assert_allclose(vu, np.broadcast_to([0.4, -0.5, 0.0], vu.shape))           # untouched nodes rewritten as is
    # friction large enough to stop sliding completely
    vt, _, _ = _one_node_collider([0.1, -0.5, 0.0], 0.0, 0.5)
    np.testing.assert_allclose(vt, 0.0, atol=1e-6)
    # separating (v.n > 0): unchanged
    vt, _, _ = _one_node_collider([0.4, 0.5, 0.1], 0.0, 0.5)
    np.testing.assert_allclose(vt, np.broadcast_to([0.4, 0.5, 0.1], vt.shape), atol=1e-6)
    # moving body: response in the body frame
    vt, _, _ = _one_node_collider([0.0, 0.0, 0.0], [0.0, 0.3, 0.0], 0.0)
    np.testing.assert_allclose(vt, np.broadcast_to(0.3 * n_up, vt.shape), atol=1e-6)


def test_k9_surface_collider_zeroes_below_plane_for_every_type():
    for surface, friction in (("sticky", 0.0), ("slip", 0.0), ("separate", 0.3)):
        o = make(0, n_grid=16)
        o.add_surface_collider([0.0, 0.5, 0.0], [0.0, 2.0, 0.0], surface=surface, friction=friction)
        o.grid_v_out[:] = [0.1, -0.2, 0.3]
        o.apply_bc(0, DT)
        V = o.grid_v_out.reshape(16, 16, 16, 3)
        ys = np.arange(16) * (2.0 / 16)
        assert np.all(V[:, ys < 0.5] == 0.0)                      # quirk Q1: zero write for all non-'cut' types
        assert np.all(V[:, ys >= 0.5] == np.float32([0.1, -0.2, 0.3]))
    with pytest.raises(ValueError):
        make(0).add_surface_collider([0, 0, 0], [0, 1, 0], surface="sticky", friction=0.1)


def test_k9_particle_mover_overwrites():
    o = make(0, n_grid=16, n_vertices=1, num_joint_v=1)
    o.x[0] = [1.0, 1.0, 1.0]
    o.add_particle_mover()
    o.grid_v_out[:] = [9.0, 9.0, 9.0]
    o.particle_move(0, None, np.array([[0.1, 0.2, 0.3]], np.float32), np.zeros((0, 3), np.float32))
    touched = o.movers[0]["weight"] > 1e-15
    assert touched.sum() == 27
    np.testing.assert_allclose(o.grid_v_out[touched], np.broadcast_to([0.1, 0.2, 0.3], (27, 3)), rtol=1e-6)
    assert np.all(o.grid_v_out[~touched] == 9.0)


def test_k9_bounding_box_and_cuboid():
    o = make(0, n_grid=16)
    o.add_bounding_box()
    o.grid_v_out[:] = [-1.0, 1.0, -1.0]
    o.apply_bc(0, DT)
    V = o.grid_v_out.reshape(16, 16, 16, 3)
    assert np.all(V[:3, :, :, 0] == 0) and np.all(V[3:, :, :, 0] == -1)        # inward x at low side zeroed
    assert np.all(V[:, 13:, :, 1] == 0) and np.all(V[:, :13, :, 1] == 1)       # outward y at high side zeroed
    o2 = make(0, n_grid=16)
    o2.set_velocity_on_cuboid([1.0, 1.0, 1.0], [0.3, 0.3, 0.3], [0.5, 0.0, 0.0], start_time=0.0, end_time=1.0)
    o2.apply_bc(0, DT)
    V = o2.grid_v_out.reshape(16, 16, 16, 3)
    inside = np.abs(np.arange(16) * 0.125 - 1.0) < 0.3
    assert np.all(V[np.ix_(inside, inside, inside)][..., 0] == 0.5)
    np.testing.assert_allclose(o2.sim.bc[0].point[0], 1.0 + DT * 0.5, rtol=1e-6)  # host-side modify (:975-981)


# ---- K10 / K11 ---------------------------------------------------------------------------------------------
def test_k10_position_clamp():
    o = trad_block(n=2, n_grid=16)
    dx = 2.0 / 16
    o.x[:] = [2 * dx + 1e-4, 1.0, 2.0 - 2 * dx - 1e-4]
    o.grid_v_out[:] = [-50.0, 0.0, 50.0]
    o.g2p_v(DT)
    np.testing.assert_allclose(o.x[:, 0], 2 * dx, rtol=1e-6)
    np.testing.assert_allclose(o.x[:, 2], 2.0 - 2 * dx, rtol=1e-6)


def test_k11_lame_parameters_and_mass():
    o = make(4)
    o.E[:], o.nu[:] = [100.0, 50.0, 1e4, 7.0], [0.3, 0.25, 0.4, 0.0]
    o.prepare_mu_lam()
    E, nu = o.E.astype(np.float64), o.nu.astype(np.float64)
    np.testing.assert_allclose(o.mu, E / (2 * (1 + nu)), rtol=1e-6)
    np.testing.assert_allclose(o.lam, E * nu / ((1 + nu) * (1 - 2 * nu)), rtol=1e-6)
    o.vol[:] = [1.0, 2.0, 3.0, 4.0]
    o.set_parameters_dict({"density": 1.5})
    np.testing.assert_allclose(o.mass, 1.5 * o.vol)
    with pytest.raises(TypeError):
        o.set_parameters_dict({"material": "unobtainium"})


def test_svd_and_qr_conventions():
    rng = np.random.default_rng(5)
    for _ in range(300):
        A = rng.normal(size=(3, 3)).astype(np.float32)
        U, s, V = O.svd3(A)
        np.testing.assert_allclose(U @ np.diag(s) @ V.T, A, atol=5e-6)
        assert abs(np.linalg.det(U.astype(np.float64)) - 1) < 1e-5 and abs(np.linalg.det(V.astype(np.float64)) - 1) < 1e-5
        assert s[0] >= s[1] >= abs(s[2]) - 1e-6 and np.sign(s[2]) == np.sign(np.linalg.det(A.astype(np.float64)))
        Q, R = O.qr_signfixed(A)
        np.testing.assert_allclose(Q @ R, A, atol=5e-6)
        assert R[0, 0] >= 0 and R[1, 1] >= 0 and abs(np.linalg.det(Q.astype(np.float64)) - 1) < 1e-5
        assert abs(R[1, 0]) + abs(R[2, 0]) + abs(R[2, 1]) == 0
