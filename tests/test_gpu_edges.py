"""Edge cases of the substep through the C ABI, both kernel back ends, against the CPU oracle: a single particle,
a dense cluster (hundreds of particles in one cell: multi-chunk blocks, DPP segments longer than a row), particles
pinned at the domain clamp, frozen particles (selection == 1), mixed frozen / simulated particles, and a context with
no particles at all."""
import numpy as np
import pytest
import torch

from mpmavatar_amd import harness, scenes
from mpmavatar_amd.scenes import _trad_scene

pytestmark = pytest.mark.gpu

MODES = ["baseline", "fast"]


def rel(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-3))


def _pair(sc, n, mode, fused=True):
    from oracle.scene_adapter import oracle_from_scene, run_scene
    o = oracle_from_scene(sc)
    run_scene(o, sc, n)
    sim = harness.build_solver(sc, "cuda:0", mode=mode)
    harness.run(sim, n, fused=fused)
    return o, sim


@pytest.mark.parametrize("mode", MODES)
def test_single_particle_free_fall(mode, oracle_lib):
    sc = _trad_scene("one", np.array([[1.0, 1.2, 1.0]], np.float32), 1e-6, 32, n_steps=50)
    o, sim = _pair(sc, 50, mode)
    x, v = sim.state.particle_x.cpu().numpy(), sim.state.particle_v.cpu().numpy()
    assert rel(x, o.x) < 1e-6 and rel(v, o.v) < 1e-5
    t = 50 * sc.dt
    assert abs(v[0, 1] + 9.8 * t) < 1e-4 * 9.8 * t + 1e-7  # one particle feels its own grid mass only: pure gravity


@pytest.mark.parametrize("mode", MODES)
def test_dense_cluster_in_one_cell(mode, oracle_lib):
    """700 particles inside one grid cell (and a few hundred more around it): several 256-particle chunks for one block,
    equal-cell runs longer than the 16-lane DPP rows, heavy same-address LDS traffic."""
    rng = np.random.default_rng(7)
    dx = 2.0 / 32
    base = np.array([16.0, 18.0, 15.0]) * dx
    inner = base + rng.uniform(0.05, 0.95, (700, 3)) * dx
    outer = base + rng.uniform(-1.5, 2.5, (300, 3)) * dx
    pts = np.concatenate([inner, outer]).astype(np.float32)
    vel = rng.normal(0, 0.2, pts.shape).astype(np.float32)
    sc = _trad_scene("cluster", pts, (dx / 8) ** 3, 32, v=vel, E=100.0, bcs=[("bounding_box", {})], n_steps=40)
    o, sim = _pair(sc, 40, mode)
    assert rel(sim.state.particle_x.cpu().numpy(), o.x) < 1e-5
    # ~700 fp32 contributions per node: the serial sum of the oracle and the tree / atomic sums of the GPU differ at the
    # 1e-4 level in the (small) velocities; the two GPU back ends must agree much better than that
    assert rel(sim.state.particle_v.cpu().numpy(), o.v) < 5e-4
    assert rel(sim.state.particle_F_trial.cpu().numpy(), o.F_trial) < 1e-4
    other = harness.build_solver(sc, "cuda:0", mode="baseline" if mode == "fast" else "fast")
    harness.run(other, 40, fused=True)
    # (round 4: the fast back end runs this scene through the fused g2p -> p2g launch, whose FMA contraction differs from the two-launch
    # form's: 1.06e-4 measured where rounds 1-3 had 7e-5; both stay inside the 5e-4 of the oracle above)
    assert rel(sim.state.particle_v.cpu().numpy(), other.state.particle_v.cpu().numpy()) < 2e-4


@pytest.mark.parametrize("mode", MODES)
def test_particles_driven_into_the_domain_clamp(mode, oracle_lib):
    """Particles thrown at the walls: the position clamp of g2p (2 dx from the faces, mpm_utils.py:779-786) and the
    bounding-box BC act; nothing may index outside the grid."""
    rng = np.random.default_rng(11)
    dx = 2.0 / 32
    pts = (np.array([2.6 * dx, 1.0, 1.0]) + rng.uniform(-0.4, 0.4, (200, 3)) * dx).astype(np.float32)
    pts2 = (np.array([1.0, 2.0 - 2.6 * dx, 2.0 - 2.6 * dx]) + rng.uniform(-0.4, 0.4, (200, 3)) * dx).astype(np.float32)
    vel = np.concatenate([np.tile([-3.0, 0.0, 0.0], (200, 1)), np.tile([0.0, 3.0, 3.0], (200, 1))]).astype(np.float32)
    sc = _trad_scene("walls", np.concatenate([pts, pts2]), (dx / 4) ** 3, 32, v=vel, E=100.0, bcs=[("bounding_box", {})],
                     n_steps=60)
    o, sim = _pair(sc, 60, mode)
    x = sim.state.particle_x.cpu().numpy()
    assert np.isfinite(x).all() and x.min() >= 2 * dx - 1e-6 and x.max() <= 2.0 - 2 * dx + 1e-6
    assert rel(x, o.x) < 1e-5
    assert rel(sim.state.particle_v.cpu().numpy(), o.v) < 1e-4


@pytest.mark.parametrize("mode", MODES)
def test_frozen_particles(mode, oracle_lib):
    """particle_selection == 1: not simulated (mpm_utils.py: every particle kernel is guarded by selection == 0).  Half
    of a cube frozen: the frozen half keeps x and v, the other half behaves as if the frozen one were not there."""
    sc = scenes.small_cube(n=6)
    sel = np.zeros(sc.n_particles, np.int32)
    sel[::2] = 1
    sc.selection = sel
    o, sim = _pair(sc, 40, mode)
    x, v = sim.state.particle_x.cpu().numpy(), sim.state.particle_v.cpu().numpy()
    assert np.array_equal(x[::2], sc.x[::2]) and np.array_equal(v[::2], sc.v[::2])
    assert rel(x, o.x) < 1e-5 and rel(v, o.v) < 1e-4
    sc.selection = np.ones(sc.n_particles, np.int32)   # everything frozen: a substep is a no-op on the particles
    sim = harness.build_solver(sc, "cuda:0", mode=mode)
    harness.run(sim, 5, fused=True)
    assert np.array_equal(sim.state.particle_x.cpu().numpy(), sc.x)


@pytest.mark.parametrize("mode", MODES)
def test_no_particles(mode):
    """An empty context steps without touching anything (a rank of a sharded run can own nothing)."""
    sc = _trad_scene("empty", np.zeros((0, 3), np.float32), 1e-6, 16, n_steps=3)
    sim = harness.build_solver(sc, "cuda:0", mode=mode)
    harness.run(sim, 3, fused=True)
    harness.run(sim, 2, fused=False)
    assert sim.state.particle_x.shape == (0, 3)
    assert sim.solver.stats()["n_active_nodes"] == 0


@pytest.mark.parametrize("mode", MODES)
@pytest.mark.parametrize("params", [{"rpic_damping": 0.3}, {"rpic_damping": -1.0}, {"grid_v_damping_scale": 0.9},
                                    {"rpic_damping": 0.5, "grid_v_damping_scale": 0.97}])
def test_rpic_and_grid_damping(mode, params, oracle_lib):
    """rpic_damping blends / drops the APIC matrix in p2g (mpm_utils.py:528-534); grid_v_damping_scale < 1 switches on
    add_damping_via_grid (mpm_solver.py:373, mpm_utils.py:1162-1174).  The drivers leave both at their defaults."""
    sc = scenes.small_cube(n=6, params=params)
    o, sim = _pair(sc, 60, mode)
    assert rel(sim.state.particle_x.cpu().numpy(), o.x) < 1e-5
    assert rel(sim.state.particle_v.cpu().numpy(), o.v) < 1e-4
    # absolute bound: C is tiny once the grid is damped, the fp32 cancellation noise of 4/dx (M - v fx) is not
    assert np.abs(sim.state.particle_C.cpu().numpy() - o.C).max() < 1e-4


@pytest.mark.parametrize("mode", MODES)
@pytest.mark.parametrize("surface", ["sticky", "slip", "cut"])
def test_surface_collider_kinds_and_grid_mask(mode, surface, oracle_lib):
    """Plane colliders of every kind (incl. quirk Q1: slip / frictional planes also end in a zero write) plus a
    grid-node mask that pins a slab of nodes (enforce_grid_velocity_by_mask, mpm_solver.py:1330-1355)."""
    from oracle.scene_adapter import oracle_from_scene
    sc = scenes.small_cube(n=6)
    sc.bcs = []
    G = sc.n_grid
    mask = np.zeros((G, G, G), np.int32)
    mask[:, :, : G // 2 - 2] = 1   # nodes with small z are pinned: the cube's low-z side hangs on them
    o = oracle_from_scene(sc)
    sim = harness.build_solver(sc, "cuda:0", mode=mode)
    y0 = float(sc.x[:, 1].min()) - 0.01
    kw = dict(point=[0.0, y0, 0.0], normal=[0.0, 2.0, 0.0], surface=surface, friction=0.0 if surface == "sticky" else 0.3)
    sim.solver.add_surface_collider(**kw)
    o.add_surface_collider(**kw)
    sim.solver.enforce_grid_velocity_by_mask(torch.as_tensor(mask.reshape(-1)))
    o.enforce_grid_velocity_by_mask(mask)
    for _ in range(60):
        sim.solver.p2g2p(sim.model, sim.state, sc.dt)
        o.p2g2p(sc.dt)
    assert rel(sim.state.particle_x.cpu().numpy(), o.x) < 1e-5
    assert rel(sim.state.particle_v.cpu().numpy(), o.v) < 1e-4


@pytest.mark.parametrize("mode", MODES)
def test_high_valence_vertex(mode, oracle_lib):
    """A triangle fan: one vertex shared by 20 elements (the vertex-force gather walks its adjacency in batches of 8,
    the garment meshes of the other tests never exceed 6-8)."""
    from mpmavatar_amd.scenes import _cloth_scene
    n = 20
    ang = np.linspace(0.0, 2.0 * np.pi, n, endpoint=False)
    ring = np.stack([1.0 + 0.12 * np.cos(ang), np.full(n, 1.2), 1.0 + 0.12 * np.sin(ang)], 1)
    ring2 = np.stack([1.0 + 0.24 * np.cos(ang + 0.1), np.full(n, 1.2), 1.0 + 0.24 * np.sin(ang + 0.1)], 1)
    verts = np.concatenate([[[1.0, 1.2, 1.0]], ring, ring2]).astype(np.float32)
    faces = [[0, 1 + i, 1 + (i + 1) % n] for i in range(n)]
    faces += [[1 + i, 1 + n + i, 1 + (i + 1) % n] for i in range(n)]
    faces += [[1 + (i + 1) % n, 1 + n + i, 1 + n + (i + 1) % n] for i in range(n)]
    faces = np.asarray(faces, np.int32)
    sc = _cloth_scene("fan", verts, faces, 32, n_steps=60)
    sc.v[:] = np.array([0.0, 0.0, 0.0], np.float32)
    sc.v[sc.n_elements + sc.n_traditional] = [0.0, 0.6, 0.0]   # pluck the hub vertex
    o, sim = _pair(sc, 60, mode)
    assert rel(sim.state.particle_x.cpu().numpy(), o.x) < 1e-5
    # 60 elements released from rest, every one of them on the return mapping's R22 = 1 discontinuity and nothing to average
    # over: 8e-3 ... 3.1e-2 relative to the 0.05 m/s the fan has reached (4e-4 ... 1.6e-3 m/s; test_gpu_parity docstring).
    # The spread is run to run: the baseline back end scatters with global fp32 atomics, whose order is not fixed.
    assert rel(sim.state.particle_v.cpu().numpy(), o.v) < 6e-2
    o1, sim1 = _pair(sc, 1, mode)
    assert rel(sim1.state.particle_v.cpu().numpy(), o1.v) < 1e-4


@pytest.mark.parametrize("rebin_interval", [0, -1000000])
def test_particles_shuffling_between_cells(rebin_interval, oracle_lib):
    """A spinning, shearing blob: over the run every particle crosses several cells (and blocks), neighbours at
    different times.  rebin_interval = 0: adaptive re-sorts keep up; negative: ONE sort at the start, so the lane order
    inside a block becomes arbitrary (runs of equal cells break up, particles sit far outside their tile) -- the
    segmented scan, the margin checks and the global-memory fallbacks all have to cope.  Baseline kernels and oracle as
    references."""
    from oracle.scene_adapter import oracle_from_scene, run_scene
    rng = np.random.default_rng(5)
    dx = 2.0 / 32
    pts = (np.array([0.75, 0.75, 0.75]) + rng.uniform(0, 0.5, (3000, 3))).astype(np.float32)
    r = pts - pts.mean(0)
    vel = (np.cross(np.array([0.0, 0.0, 9.0]), r) + np.stack([5.0 * r[:, 1], 0 * r[:, 0], 2.0 * r[:, 0]], 1)).astype(np.float32)
    sc = _trad_scene("whirl", pts, (dx / 2) ** 3, 32, v=vel, E=20.0, bcs=[("bounding_box", {})], n_steps=80)
    sc.dt = 1e-3
    sc.params["g"] = [0.0, 0.0, 0.0]
    n = 80
    o = oracle_from_scene(sc)
    run_scene(o, sc, n)
    assert np.abs(o.x - sc.x).max() > 1.2 * dx   # they do travel across cells
    fast = harness.build_solver(sc, "cuda:0", mode="fast", rebin_interval=rebin_interval)
    harness.run(fast, n, fused=True)
    base = harness.build_solver(sc, "cuda:0", mode="baseline")
    harness.run(base, n, fused=True)
    xf, xb = fast.state.particle_x.cpu().numpy(), base.state.particle_x.cpu().numpy()
    vf, vb = fast.state.particle_v.cpu().numpy(), base.state.particle_v.cpu().numpy()
    assert np.isfinite(xf).all()
    assert rel(xf, o.x) < 1e-4 and rel(vf, o.v) < 1e-3
    assert rel(xf, xb) < 2e-5 and rel(vf, vb) < 5e-4
    st = fast.solver.stats()
    assert st["n_dropped"] == 0
    if rebin_interval < 0:
        assert st["rebins"] == 1 and st["n_fallback_particles"] > 1000


@pytest.mark.parametrize("mode", MODES)
@pytest.mark.parametrize("n_grid", [30, 34, 45])
def test_grid_size_not_a_multiple_of_the_block(mode, n_grid, oracle_lib):
    """run_demo.py uses a 250^3 grid: the 4x4x4-node blocks of the fast back end overhang the grid there.  Small
    versions with material pressed into the far corner, where the overhanging blocks are."""
    dx = 2.0 / n_grid
    rng = np.random.default_rng(3)
    hi = 2.0 - 2.2 * dx
    pts = (hi - rng.uniform(0, 5 * dx, (600, 3))).astype(np.float32)
    vel = np.tile([1.5, 1.0, 2.0], (600, 1)).astype(np.float32)
    sc = _trad_scene("corner", pts, (dx / 2) ** 3, n_grid, v=vel, E=50.0, bcs=[("bounding_box", {})], n_steps=60)
    o, sim = _pair(sc, 60, mode)
    x = sim.state.particle_x.cpu().numpy()
    assert np.isfinite(x).all() and x.max() <= 2.0 - 2 * dx + 1e-6
    assert rel(x, o.x) < 1e-5
    assert rel(sim.state.particle_v.cpu().numpy(), o.v) < 2e-4
    sc2 = scenes.sheet(n=20, n_grid=n_grid, collider_subdiv=2, n_steps=40, span=(0.7, 1.3), y=1.2, sphere_r=0.18,
                       sphere_c=(1.0, 0.99, 1.0), name=f"sheet-{n_grid}")
    o2, sim2 = _pair(sc2, 40, mode)
    assert rel(sim2.state.particle_x.cpu().numpy(), o2.x) < 1e-4


@pytest.mark.parametrize("material", ["metal", "foam", "plasticine"])
def test_out_of_margin_traditional_particles_with_hardening(material, oracle_lib):
    """The fused stress update of a traditional particle runs in the front of p2g; a particle that then turns out to
    be outside its tile margin is scattered by the global-memory path, which must reuse that stress instead of updating
    the (hardening / softening) material state a second time.  Found by tools/gpu/fuzz.py."""
    rng = np.random.default_rng(2)
    dx = 2.0 / 30
    pts = (np.array([0.8, 1.0, 1.0]) + rng.uniform(-1, 1, (1500, 3)) * 3 * dx).astype(np.float32)
    r = pts - pts.mean(0)
    # rotation + stretch / squeeze (so that the material yields and hardens) + translation (so that it leaves its tiles)
    vel = (np.cross(np.array([0.0, 3.0, 2.0]), r) + r * np.array([8.0, -6.0, 3.0]) + np.array([3.0, -1.0, 1.5])).astype(np.float32)
    params = {"yield_stress": 2.0, "hardening": 1, "xi": 0.1, "plastic_viscosity": 0.5}
    sc = _trad_scene("harden", pts, (dx / 2) ** 3, 30, material=material, v=vel, E=100.0, bcs=[("bounding_box", {})],
                     params=params, n_steps=60)
    sc.dt = 1e-3
    from oracle.scene_adapter import oracle_from_scene, run_scene
    o = oracle_from_scene(sc)
    run_scene(o, sc, 45)
    sim = harness.build_solver(sc, "cuda:0", mode="fast", rebin_interval=-1000000)
    harness.run(sim, 45, fused=True)
    st = sim.solver.stats()
    assert st["n_fallback_particles"] > 1000 and st["n_dropped"] == 0
    assert rel(sim.state.particle_x.cpu().numpy(), o.x) < 2e-5
    assert rel(sim.state.particle_v.cpu().numpy(), o.v) < 2e-4
    assert rel(sim.state.particle_F_trial.cpu().numpy(), o.F_trial) < 2e-4


@pytest.mark.parametrize("mode", MODES)
@pytest.mark.parametrize("fuse_grid", ["0", "1"])
def test_nodes_exactly_on_a_cuboid_face(mode, fuse_grid, oracle_lib, monkeypatch):
    """A velocity cuboid (set_velocity_on_cuboid, mpm_solver.py:950-981) whose faces coincide with grid-node planes:
    14 * 0.05f - 1.0f is -0.3f when the product is rounded and -0.29999999 under an FMA, so `|offset| < size` flips with
    the compiler's contraction choice.  Both back ends round the product (like the oracle): every node must be
    classified identically, then the release window (reset=1: 15 substeps of zero grid velocity) and free motion."""
    monkeypatch.setenv("MPMHIP_FUSE_GRID", fuse_grid)
    sc = scenes.sheet(n=10, n_grid=40, collider_subdiv=1, span=(0.6, 1.4), y=1.2, sphere_r=0.2, sphere_c=(1.0, 0.97, 1.0), name="edge")
    sc.bcs = list(sc.bcs) + [("velocity_cuboid", {"point": [1.1, 1.2, 1.0], "size": [0.05, 0.3, 0.3], "velocity": [-0.8, 0.0, 0.0],
                                                   "start_time": 0.0, "end_time": 0.0011, "reset": 1})]
    sc.dt = 1e-4
    for n in (1, 8, 20):   # inside the window; across its end; inside the reset window
        o, sim = _pair(sc, n, mode, fused=False)
        G = sc.n_grid
        og = np.asarray(o.grid_v_out).reshape(G, G, G, 3)
        live = np.asarray(o.grid_m).reshape(G, G, G) > 0
        _, _, vo = sim.solver.export_grid()
        d = np.abs(vo.cpu().numpy() - og).max(-1) * live
        assert (d > 1e-4).sum() == 0, f"{(d > 1e-4).sum()} nodes classified differently after {n} substeps"
        assert rel(sim.state.particle_x.cpu().numpy(), o.x) < 1e-5
        assert np.abs(sim.state.particle_v.cpu().numpy() - o.v).max() < 1e-4 * max(np.abs(o.v).max(), 0.1)


@pytest.mark.parametrize("subdiv", [2, 4])
@pytest.mark.parametrize("split", ["0", "1"])
def test_body_face_splat_layouts(subdiv, split, oracle_lib, monkeypatch):
    """The body-face splat (mpm_solver.py:829-880) in its four forms: bins of a few faces ((face, node) lanes; both passes in the p2g
    launch = ONE pass through the seven-channel tile, MPMHIP_SPLIT_SPLAT=0, or pass 0 in front of the stress kernel and pass 1 in p2g)
    and bins of hundreds of faces (lane = face with the DPP pre-reduction, two passes) -- a 320-face and a 5,120-face sphere under
    the same sheet, against the oracle, and the two launch layouts against each other."""
    monkeypatch.setenv("MPMHIP_SPLIT_SPLAT", split)
    sc = scenes.sheet(n=40, n_grid=48, collider_subdiv=subdiv, span=(0.6, 1.4), y=1.215, name=f"splat-{subdiv}")
    sc.dt = 1e-4
    o, sim = _pair(sc, 60, "fast", fused=True)
    st = sim.solver.stats()
    assert st["n_collider_nodes"] > 100 and st["n_dropped"] == 0
    x, v = sim.state.particle_x.cpu().numpy(), sim.state.particle_v.cpu().numpy()
    assert rel(x, o.x) < 1e-5
    assert np.abs(v - o.v).max() < 2e-4 * max(np.abs(o.v).max(), 0.1)
    moved = np.abs(o.v[:, 1] + 9.8 * 60 * sc.dt) > 1e-3     # particles the sphere has slowed down: the collider is felt
    assert moved.sum() > 50
