"""GPU parity: the HIP solver (through the C ABI via the warp_mpm shim) against the serial fp32 CPU oracle
on identical seeded inputs.  The reference itself cannot run here (Warp is NVIDIA-only); the oracle is pinned by
fixtures the reference's own kernel bodies produced (tests/test_ref_golden.py) and by tests/test_oracle_*.py.

Tolerances (BASELINE.json north_star: particle x / v within 1e-4 relative after N substeps):
  rel(a, b) = max|a-b| / max(max|b|, 1e-3).
Cloth scenes with gamma > 0 carry a documented caveat: the reference's anisotropic return mapping is discontinuous at
R22 == 1 (mpm_utils.py:196-204: shear kept above, projected to ~0 just below) and a cloth at rest sits exactly there, so
the rounding of the QR flips branches.  The reference moves by up to 1.2e-3 (relative) / 2.2e-4 m/s against ITSELF when
its svd3 / qr3 are fp32- instead of fp64-accurate (tests/test_ref_golden.py, tests/golden/ref_seq_*.npz); the velocity bounds
of the cloth cases below are a few times what was measured here (tools/gpu/dv_report.py: sheet <= 1.7e-4, garment <= 8.4e-4,
demo mix <= 4e-6), not the blanket 5e-2 of round 1.  Positions stay < 1e-6.
"""
import numpy as np
import pytest
import torch

from mpmavatar_amd import scenes

pytestmark = pytest.mark.gpu

MODES = ["baseline", "fast"]


def rel(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-3))


def _run_pair(sc, n_steps, mode, fused=False):
    from mpmavatar_amd import harness
    from oracle.scene_adapter import oracle_from_scene, run_scene
    o = oracle_from_scene(sc)
    run_scene(o, sc, n_steps)
    sim = harness.build_solver(sc, "cuda:0", mode=mode)
    harness.run(sim, n_steps, fused=fused)
    st = sim.state
    out = {k: getattr(st, k).detach().cpu().numpy() for k in
           ("particle_x", "particle_v", "particle_C", "particle_F_trial", "particle_d")}
    return o, out, sim


@pytest.mark.parametrize("mode", MODES)
@pytest.mark.parametrize("material", ["jelly", "sand", "metal", "foam", "plasticine"])
def test_cube_materials(mode, material, oracle_lib):
    params = {"friction_angle": 40.0} if material == "sand" else {}
    if material in ("metal", "foam", "plasticine"):
        params.update({"yield_stress": 2.0, "hardening": 1, "xi": 0.1, "plastic_viscosity": 0.5})
    sc = scenes.small_cube(material=material, params=params)
    o, g, _ = _run_pair(sc, 100, mode)
    assert np.isfinite(g["particle_x"]).all()
    assert rel(g["particle_x"], o.x) < 1e-4
    assert rel(g["particle_v"], o.v) < 1e-4
    assert rel(g["particle_F_trial"], o.F_trial) < 1e-4


@pytest.mark.parametrize("mode", MODES)
def test_sheet_over_sphere(mode, oracle_lib):
    sc = scenes.small_sheet()
    o, g, _ = _run_pair(sc, 200, mode)
    assert np.isfinite(g["particle_x"]).all()
    assert rel(g["particle_x"], o.x) < 1e-5
    assert rel(g["particle_v"], o.v) < 1e-3   # measured 2.3e-5 (peak 1.7e-4 on the way); see module docstring
    assert rel(g["particle_d"], o.d) < 3e-3   # measured 1.9e-5 (peak 4.5e-4)


@pytest.mark.parametrize("mode", MODES)
def test_sheet_short_strict(mode, oracle_lib):
    """Few substeps: before branch flips decorrelate velocities the strict 1e-4 bound holds on everything."""
    sc = scenes.small_sheet()
    o, g, _ = _run_pair(sc, 1, mode)
    assert rel(g["particle_x"], o.x) < 1e-6
    assert rel(g["particle_v"], o.v) < 1e-4
    assert rel(g["particle_C"], o.C) < 1e-4
    assert rel(g["particle_d"], o.d) < 1e-5


@pytest.mark.parametrize("mode", MODES)
def test_garment_with_mover(mode, oracle_lib):
    sc = scenes.small_garment()
    o, g, _ = _run_pair(sc, 100, mode)
    assert rel(g["particle_x"], o.x) < 1e-5
    assert rel(g["particle_v"], o.v) < 3e-3   # measured 5.6e-4 (fast) / 8.4e-4 (baseline); the reference against itself: 7.2e-4


@pytest.mark.parametrize("mode", MODES)
def test_demo_mix(mode, oracle_lib):
    sc = scenes.demo_mix(n_grid=48, n_sheet=16, sand=(16, 3, 8))
    o, g, _ = _run_pair(sc, 100, mode)
    assert rel(g["particle_x"], o.x) < 1e-5
    assert rel(g["particle_v"], o.v) < 1e-4   # measured 4.1e-6
    assert rel(g["particle_F_trial"], o.F_trial) < 1e-4


@pytest.mark.parametrize("mode", MODES)
def test_fused_steps_match_single_steps(mode, oracle_lib):
    sc = scenes.small_garment()
    from mpmavatar_amd import harness
    a = harness.build_solver(sc, "cuda:0", mode=mode)
    b = harness.build_solver(sc, "cuda:0", mode=mode)
    harness.run(a, 20, fused=False)
    harness.run(b, 20, fused=True)
    xa, xb = a.state.particle_x.cpu().numpy(), b.state.particle_x.cpu().numpy()
    assert rel(xa, xb) < 1e-5


@pytest.mark.parametrize("mode", MODES)
def test_grid_fields_after_one_step(mode, oracle_lib):
    """Grid-level parity: dense exports of grid_m / grid_v_out against the oracle's arrays."""
    sc = scenes.small_sheet()
    o, g, sim = _run_pair(sc, 1, mode)
    m, vi, vo = sim.solver.export_grid()
    G = sc.n_grid
    om = o.grid_m.reshape(G, G, G)
    assert rel(m.cpu().numpy(), om) < 1e-5
    act = om > 1e-14  # away from the 1e-15 mass threshold
    assert rel(vo.cpu().numpy()[act], o.grid_v_out.reshape(G, G, G, 3)[act]) < 1e-4


def test_no_device_is_loud():
    from mpmavatar_amd.warp_mpm import MPMWARP
    from mpmavatar_amd._lib import MPMHipError
    with pytest.raises(MPMHipError):
        MPMWARP(8, 0, 0, n_grid=16, grid_lim=2.0, device="cpu")


@pytest.mark.parametrize("scene", ["cube", "sheet", "garment"])
def test_out_of_margin_paths(scene, oracle_lib):
    """Particles (and body faces) that left the tile margin of the block they were sorted into: with the adaptive
    re-sort switched off (rebin_interval < 0: one sort at the start, drift flag ignored) thousands of particles end up
    on the global-memory paths of p2g / g2p / the body-face splat, which must give the same answer as the tiled ones.
    The scenes get a uniform extra velocity (~1.7 cells of travel).  Checked against the oracle and against the same
    run with the adaptive re-sort on (the fast cloth scenes are chaotic at the 1e-4 level, see the module docstring,
    so the oracle bound is loose there and the control run carries the strict one)."""
    from mpmavatar_amd import harness
    from oracle.scene_adapter import oracle_from_scene, run_scene
    n = 250

    def make():
        sc = {"cube": lambda: scenes.small_cube(), "sheet": scenes.small_sheet, "garment": scenes.small_garment}[scene]()
        drift = 1.7 * sc.grid_lim / sc.n_grid
        sc.v = (sc.v + np.float32(drift / (n * sc.dt)) * np.array([0.8, 0.0, 0.6], np.float32)).astype(np.float32)
        return sc

    sc = make()
    o = oracle_from_scene(sc)
    run_scene(o, sc, n)
    res = {}
    for ri in (0, -1000000):
        sim = harness.build_solver(make(), "cuda:0", mode="fast", rebin_interval=ri)
        harness.run(sim, n, fused=True)
        res[ri] = (sim.state.particle_x.detach().cpu().numpy(), sim.state.particle_v.detach().cpu().numpy(),
                   sim.solver.stats())
    x, v, st = res[-1000000]
    xc, vc, stc = res[0]
    assert st["rebins"] == 1 and stc["rebins"] >= 2
    assert st["n_fallback_particles"] > 1000, "scene did not drift far enough to exercise the out-of-margin paths"
    assert rel(x, o.x) < (1e-4 if scene == "cube" else 1e-3)
    assert rel(v, o.v) < (3e-4 if scene == "cube" else 1e-2)
    assert rel(x, xc) < (2e-5 if scene == "cube" else 2e-4)


@pytest.mark.parametrize("sand", [(16, 3, 8), (32, 4, 16)])  # 384: joint workgroups; 2048: second tile pass of p2g
@pytest.mark.parametrize("mode", MODES)
def test_staged_sand_release(mode, sand, oracle_lib):
    """run_demo.py:524: the mover holds the trailing sand particles at zero velocity and lets go of them in stages
    (the length of joint_traditional_v shrinks over time).  Single-step and fused driving must agree with the oracle
    across several release boundaries."""
    from mpmavatar_amd import harness
    n_sand = sand[0] * sand[1] * sand[2]
    mk = lambda: scenes.demo_mix(n_grid=48, n_sheet=16, sand=sand, hold=(20, 7, n_sand // 6))
    sc = mk()
    assert sc.joint_t_count(0) == sc.n_traditional and 0 < sc.joint_t_count(45) < sc.n_traditional
    o, g, _ = _run_pair(sc, 80, mode)
    assert rel(g["particle_x"], o.x) < 1e-5
    assert rel(g["particle_v"], o.v) < 5e-4
    held = slice(sc.n_elements + sc.n_traditional - sc.joint_t_count(80), sc.n_elements + sc.n_traditional)
    if sc.joint_t_count(80) > 0:   # still held: has not moved
        assert np.abs(g["particle_x"][held] - sc.x[held]).max() < 1e-6
    free = slice(sc.n_elements, sc.n_elements + sc.n_traditional - sc.joint_t_count(30))
    assert (sc.x[free, 1] - g["particle_x"][free, 1]).max() > 1e-5  # released sand falls
    b = harness.build_solver(mk(), "cuda:0", mode=mode)
    harness.run(b, 80, fused=True)
    assert rel(b.state.particle_x.cpu().numpy(), g["particle_x"]) < 1e-6


@pytest.mark.parametrize("mode", MODES)
def test_several_mesh_colliders_and_movers(mode, oracle_lib):
    """mpm_solver.py:385-419,421-481 loop over LISTS of mesh colliders / particle movers.  All colliders splat the solver's
    one body mesh (only the friction of their collide step differs) and every mover is handed the same joint velocities, so
    the fast back end shares the splat and repeats the collide step; the result must match the oracle, which runs the
    reference's loops literally."""
    from mpmavatar_amd import harness
    from oracle.scene_adapter import oracle_from_scene, run_scene
    sc = scenes.small_garment()
    o = oracle_from_scene(sc)
    o.add_mesh_collider(friction=0.15)
    o.add_mesh_collider(friction=0.9)
    o.add_particle_mover()
    run_scene(o, sc, 40)
    sim = harness.build_solver(sc, "cuda:0", mode=mode)
    sim.solver.add_mesh_collider(sim.solver.mesh.id, n_grid=sc.n_grid, friction=0.15)
    sim.solver.add_mesh_collider(sim.solver.mesh.id, n_grid=sc.n_grid, friction=0.9)
    sim.solver.add_particle_mover(n_grid=sc.n_grid)
    harness.run(sim, 40, fused=(mode == "fast"))
    assert rel(sim.state.particle_x.cpu().numpy(), o.x) < 1e-5
    assert rel(sim.state.particle_v.cpu().numpy(), o.v) < 2e-3   # cloth at rest: R22 = 1 sensitivity, see module docstring


@pytest.mark.parametrize("mode", MODES)
def test_swaying_body_posed_per_frame(mode, oracle_lib):
    """the body sways with 0.5 sin(2 pi t) m/s.  Like the reference drivers the scene poses it once per frame
    and moves it with the finite-difference velocity inside the frame (train_material_params.py:616-626); the joints ride
    on it.  Single-step and fused driving (split at the frame boundaries) against the oracle across three frames."""
    from mpmavatar_amd import harness
    from oracle.scene_adapter import oracle_from_scene, run_scene
    mk = lambda: scenes.garment_cylinder(n_theta=32, n_h=24, n_grid=48, collider_subdiv=2, name="garment-sway", sway=(0.5, 2.0, 20))
    sc = mk()
    assert abs(sc.body_at(45)[1][0, 0] - sc.body_at(5)[1][0, 0]) > 1e-3          # the velocity really changes from frame to frame
    o = oracle_from_scene(sc)
    run_scene(o, sc, 65)
    a = harness.build_solver(mk(), "cuda:0", mode=mode)
    harness.run(a, 65, fused=False)
    b = harness.build_solver(mk(), "cuda:0", mode=mode)
    harness.run(b, 65, fused=True)
    xa, xb = a.state.particle_x.cpu().numpy(), b.state.particle_x.cpu().numpy()
    assert rel(xa, o.x) < 1e-5 and rel(xb, xa) < 1e-6
    assert rel(a.state.particle_v.cpu().numpy(), o.v) < 5e-3   # cloth at rest on the R22 = 1 discontinuity, see module docstring
