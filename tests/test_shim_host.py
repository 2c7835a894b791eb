"""Host-side logic of the warp_mpm shim that needs no GPU: state/model containers (CPU tensors), rebinding and version
tracking, parameter maths, argument validation, synthetic scene builders and the garment construction helpers."""
import math

import numpy as np
import pytest
import torch

from mpmavatar_amd import garment, scenes
from mpmavatar_amd.warp_mpm import MPMModelStruct, MPMStateStruct, from_torch_safe


def make_state(n_e=4, n_t=3, n_v=5):
    st = MPMStateStruct()
    n_p = n_e + n_t + n_v
    st.init(n_p, n_e, n_v, device="cpu")
    fl = np.zeros((3, n_p), np.int32)
    fl[0, n_e:n_e + n_t] = 1; fl[1, n_e + n_t:] = 1; fl[2, :n_e] = 1
    faces = torch.tensor([[0, 1, 2], [1, 2, 3], [2, 3, 4], [0, 2, 4]])
    st.from_torch(torch.rand(n_p, 3), torch.rand(n_p), torch.rand(n_e, 3, 3), torch.rand(n_e, 3), faces, fl[0], fl[1],
                  fl[2], torch.zeros(n_p - n_v, 6), device="cpu", n_grid=32, grid_lim=2.0)
    return st, fl


def test_state_init_shapes_follow_reference():
    st, _ = make_state()
    assert st.particle_x.shape == (12, 3) and st.particle_F.shape == (7, 3, 3) and st.particle_d.shape == (4, 3, 3)
    assert st.vertex_force.shape == (5, 3) and st.faces.dtype == torch.float32 and st.faces.shape == (4, 3)
    assert st.particle_selection.dtype == torch.int32 and st.grid_res == 32


def test_flags_must_match_index_blocks():
    st, fl = make_state()
    bad = fl.copy(); bad[2, 0] = 0
    with pytest.raises(RuntimeError):
        st.from_torch(torch.rand(12, 3), torch.rand(12), torch.rand(4, 3, 3), torch.rand(4, 3), torch.zeros(4, 3), bad[0],
                      bad[1], bad[2], device="cpu")


def test_reset_state_rebinds_and_resets():
    st, _ = make_state()
    v0 = st._version
    x_new = torch.rand(12, 3)
    st.particle_C.fill_(3.0)
    st.reset_state(5, x_new, torch.rand(4, 3, 3), None, torch.ones(12, 3), tensor_R_inv=torch.rand(4, 3), device="cpu")
    assert st._version > v0
    assert st.particle_x.data_ptr() == x_new.data_ptr()                  # aliases the caller's tensor (reference :283)
    assert float(st.particle_C.abs().max()) == 0.0
    assert torch.equal(st.particle_F, torch.eye(3).expand(7, 3, 3)) and torch.equal(st.particle_F_trial, st.particle_F)
    assert float(st.particle_stress.abs().max()) == 0.0 and float(st.vertex_force.abs().max()) == 0.0
    assert torch.equal(st.particle_v, torch.ones(12, 3))


def test_reset_density_and_mass():
    st, _ = make_state()
    st.reset_density(torch.full((12,), 2.0), None, "cpu", update_mass=True)
    assert torch.allclose(st.particle_mass, 2.0 * st.particle_vol)


def test_model_parameters():
    md = MPMModelStruct()
    md.init(6, device="cpu")
    md.init_other_params(n_grid=200, grid_lim=2.0, device="cpu")
    assert md.dx == pytest.approx(0.01) and md.inv_dx == pytest.approx(100.0)
    assert md.material == 0 and md.grid_v_damping_scale == 1.1 and md.softening == 0.1 and md.rpic_damping == 0.0
    v0 = md._scalar_version
    md.friction_angle = 40.0
    assert md._scalar_version > v0
    md.from_torch(torch.full((6,), 100.0), torch.full((6,), 0.3), torch.full((6,), 500.0), torch.full((6,), 500.0), device="cpu")
    assert torch.allclose(md.mu, torch.full((6,), 100.0 / 2.6)) and torch.allclose(md.lam, torch.full((6,), 100 * 0.3 / (1.3 * 0.4)))


def test_from_torch_safe_checks():
    assert from_torch_safe(torch.zeros(4, 3), dtype="vec3").shape == (4, 3)
    with pytest.raises(RuntimeError):
        from_torch_safe(torch.zeros(4, 3, dtype=torch.float64), dtype="vec3")
    with pytest.raises(RuntimeError):
        from_torch_safe(torch.zeros(4, 2), dtype="vec3")
    with pytest.raises(RuntimeError):
        from_torch_safe(torch.zeros(3, 4).t(), dtype="vec3")


def test_garment_construction_matches_reference_maths():
    verts, faces = garment.grid_sheet(5, 4, 0.0, 1.0, 0.0, 0.9, 1.0)
    d, rest, e_vol, v_vol = garment.compute_dir_vol(verts, faces, thickness=1e-5)
    assert d.shape == (24, 3, 3) and np.allclose(np.linalg.norm(d[:, :, 2], axis=1), 1.0, atol=1e-6)
    area = 0.5 * np.linalg.norm(np.cross(d[:, :, 0], d[:, :, 1]), axis=1)
    assert np.allclose(e_vol, 0.25 * 1e-5 * area, rtol=1e-5)
    assert np.isclose(v_vol.sum(), 3 * e_vol.sum(), rtol=1e-5)
    ri = garment.compute_rest_dir_inv(rest)
    # [[R11,R12],[0,R22]] @ [[i11,i12],[0,i22]] = I
    assert np.allclose(rest[:, 0] * ri[:, 0], 1) and np.allclose(rest[:, 0] * ri[:, 1] + rest[:, 1] * ri[:, 2], 0, atol=1e-4)
    assert np.allclose(garment.compute_rest_dir_inv_from_vf(verts, faces), ri, rtol=1e-5)
    scale, shift = garment.world_to_sim(verts * 3.0 + 5.0)
    p = (verts * 3.0 + 5.0) * scale + shift
    assert np.isclose((p.max(0) - p.min(0)).max(), 1.0, atol=1e-5) and np.allclose((p.max(0) + p.min(0)) / 2, 1.0, atol=1e-5)


def test_scene_sizes_match_survey():
    sc = scenes.sheet()
    assert (sc.n_vertices, sc.n_elements, sc.n_particles) == (166464, 331298, 497762) and sc.n_grid == 256
    assert sc.mesh_faces.shape == (20480, 3)
    cy = scenes.garment_cylinder()
    assert (cy.n_vertices, cy.n_elements) == (40000, 79600) and cy.num_joint_v == 400 and cy.num_joint_f == 400
    assert (cy.faces[:cy.num_joint_f] < cy.num_joint_v).all()      # joint faces only reference joint vertices
    cu = scenes.cube()
    assert cu.n_traditional == 8000 and cu.n_grid == 64
    for s in (sc, cy, cu):
        dx = s.grid_lim / s.n_grid
        assert s.x.min() > 2 * dx and s.x.max() < s.grid_lim - 2 * dx  # inside the clamp range (quirk Q6)


def test_icosphere_is_closed_and_outward():
    V, F = garment.icosphere(2, 0.5, (1, 1, 1))
    assert F.shape[0] == 20 * 16
    n = np.cross(V[F[:, 1]] - V[F[:, 0]], V[F[:, 2]] - V[F[:, 0]])
    c = V[F].mean(1) - 1.0
    assert ((n * c).sum(1) > 0).all()
    assert math.isclose(np.linalg.norm(V - 1.0, axis=1).max(), 0.5, rel_tol=1e-5)


def test_small_state_round_trip():
    """to_small_state / MPMSmallStateStruct.to_large_state (mpm_data_structure.py:523-607): the reduced state has x, v,
    C for every particle and d for the elements; going back copies those and the static fields (vol, density, mass,
    selection, D_inv, faces, R_inv), zeroes vertex_force and gives the new state its own grid size."""
    from mpmavatar_amd.warp_mpm import MPMSmallStateStruct
    st, _ = make_state()
    small = st.to_small_state(device="cpu")
    assert isinstance(small, MPMSmallStateStruct)
    assert small.particle_x.shape == (12, 3) and small.particle_C.shape == (12, 3, 3) and small.particle_d.shape == (4, 3, 3)
    assert float(small.particle_x.abs().sum()) == 0.0
    small.particle_x.copy_(torch.rand(12, 3)); small.particle_v.copy_(torch.rand(12, 3))
    small.particle_C.copy_(torch.rand(12, 3, 3)); small.particle_d.copy_(torch.rand(4, 3, 3))
    big = small.to_large_state(st, device="cpu")
    assert torch.equal(big.particle_x, small.particle_x) and torch.equal(big.particle_v, small.particle_v)
    assert torch.equal(big.particle_C.reshape(12, 3, 3), small.particle_C)
    assert torch.equal(big.particle_d.reshape(4, 3, 3), small.particle_d)
    for name in ("particle_vol", "particle_mass", "particle_density", "particle_selection", "particle_D_inv", "faces", "particle_R_inv"):
        assert torch.equal(getattr(big, name), getattr(st, name)), name
    assert float(big.vertex_force.abs().sum()) == 0.0 and big.grid_res == st.grid_res
