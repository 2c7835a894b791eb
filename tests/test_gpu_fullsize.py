"""BASELINE.json's configurations at FULL size against the CPU oracle, in the driver-run suite.

S1 cube-8k (config 1): 100 substeps, serial oracle.  S2 garment-120k-iso (config 2) and S3 garment-120k-aniso (config 3,
collider + mover + swaying body): 50 substeps, OpenMP oracle on all host threads.  S4 sheet-500k (config 4, the headline
workload): 20 substeps and the full 1000-substep protocol, OpenMP oracle.  The oracle is pinned by the reference's own source (tests/test_ref_golden.py);
here it carries that to the sizes the reference fixtures cannot reach.  Bounds: x and v within 1e-4 (north star); for S3
the cloth rests on the return mapping's R22 = 1 discontinuity from the first substep on, so its v bound is the reference's
own sensitivity there (1.5 x the self-distance, in m/s, of the reference's garment sequence, tests/golden/ref_seq_garment.npz)."""
import os

import numpy as np
import pytest

import refgolden as rg
from mpmavatar_amd import harness, scenes

pytestmark = pytest.mark.gpu


def _pair(name, n, omp):
    from oracle.scene_adapter import omp_threads, oracle_from_scene, run_scene
    sc = scenes.REGISTRY[name]()
    o = oracle_from_scene(sc, omp=omp, n_threads=omp_threads() if omp else 1)
    run_scene(o, sc, n)
    sim = harness.build_solver(scenes.REGISTRY[name](), "cuda:0", mode="fast")
    harness.run(sim, n, fused=True)
    st = sim.solver.stats()
    assert st["n_dropped"] == 0
    return sc, o, sim.state.particle_x.cpu().numpy(), sim.state.particle_v.cpu().numpy()


def test_s1_cube_8k_100_substeps(oracle_lib):
    sc, o, x, v = _pair("cube-8k", 100, omp=False)
    assert sc.n_particles == 8000 and sc.n_grid == 64
    assert rg.rel(x, o.x) < 1e-4 and rg.rel(v, o.v) < 1e-4


def test_s2_garment_120k_isotropic_50_substeps(oracle_lib):
    sc, o, x, v = _pair("garment-120k-iso", 50, omp=True)
    assert sc.n_particles == 119600 and sc.n_grid == 128
    assert rg.rel(x, o.x) < 1e-4 and rg.rel(v, o.v) < 1e-4


def test_s3_garment_120k_anisotropic_with_collider_50_substeps(oracle_lib):
    sc, o, x, v = _pair("garment-120k-aniso", 50, omp=True)
    assert sc.n_elements == 79600 and sc.n_vertices == 40000 and sc.mesh_faces is not None and sc.num_joint_v > 0
    z = rg.load("ref_seq_garment")
    # The kick a branch flip at R22 = 1 gives a vertex is set by the material (gamma, kappa, dt), not by how fast the body
    # moves, and this scene's body starts slowly (0.06 m/s in the first frame against 0.3 m/s in the fixture's), so the
    # envelope is taken in m/s: how far the reference's garment run moves away from itself when svd3 / qr3 are fp32-accurate.
    envelope = max(float(np.abs(z[f"alt_s{c}_particle_v"] - z[f"s{c}_particle_v"]).max()) for c in (40, 80))
    assert rg.rel(x, o.x) < 1e-4
    dv = float(np.abs(v - o.v).max())
    assert dv < 1.5 * envelope, (dv, envelope)


def _follow(name, checkpoints):
    """HIP (fused calls between the checkpoints) against the OpenMP oracle, and the oracle against ITSELF with another
    thread count -- i.e. another summation order of its atomic adds, nothing else.  The second distance is the scene's own
    sensitivity at full size (cloth with shear friction sits on the return mapping's R22 = 1 discontinuity, mpm_utils.py:196-204):
    the envelope a correct implementation can be held to.  Rows: (substep, rel dx, per-particle rel dx, |dv| max, oracle self
    |dv| max, 99.9th percentile of |dv|, of the oracle's self |dv|, max |v|)."""
    from oracle.scene_adapter import omp_threads, oracle_from_scene, run_scene
    sc = scenes.REGISTRY[name]()
    oa = oracle_from_scene(sc, omp=True, n_threads=omp_threads())
    ob = oracle_from_scene(scenes.REGISTRY[name](), omp=True, n_threads=max(omp_threads() // 3, 2))
    sim = harness.build_solver(scenes.REGISTRY[name](), "cuda:0", mode="fast")
    rows, done = [], 0
    for cp in checkpoints:
        run_scene(oa, sc, cp - done, k0=done)
        run_scene(ob, sc, cp - done, k0=done)
        harness.run(sim, cp - done, fused=True)
        done = cp
        x, v = sim.state.particle_x.cpu().numpy(), sim.state.particle_v.cpu().numpy()
        dv, ds = np.linalg.norm(v - oa.v, axis=1), np.linalg.norm(ob.v - oa.v, axis=1)
        rows.append((cp, rg.rel(x, oa.x), rg.rel_pp(x, oa.x), float(dv.max()), float(ds.max()), float(np.quantile(dv, 0.999)),
                     float(np.quantile(ds, 0.999)), float(np.abs(oa.v).max()), rg.rel(ob.x, oa.x)))
    st = sim.solver.stats()
    assert st["n_dropped"] == 0
    return sc, rows


def _check(rows, what, max_factor=2.0, p_factor=2.0):
    env_max = max(r[4] for r in rows)
    env_p999 = max(r[6] for r in rows)
    for cp, dx, ppx, dv, ds, dvq, dsq, vmax, dxs in rows:
        print(f"{what} substep {cp}: x {dx:.1e}; |dv| max {dv:.2e} (oracle vs itself {ds:.2e}), p99.9 {dvq:.2e} ({dsq:.2e}), top speed {vmax:.2f}")
        assert dx < 1e-4 and ppx < 1e-4, f"{what} substep {cp}: x {dx:.2e} (per particle {ppx:.2e})"
        # v: 1e-4 of the top speed (north star), or -- where the oracle itself does not hold that against a change of its
        # summation order -- 2 x its own distance from itself (round 4: was 3 x.  Measured ratios over four runs: maximum 0.6 ... 1.84, 99.9th
        # percentile 0.9 ... 1.3 -- both are statistics of amplified rounding and move by +-50 % from run to run, the HIP path's flush
        # atomics being unordered; 1.5 x would fail one run in a few) (max over the checkpoints: both are maxima over 1e5 particles)
        bound = max(1e-4 * max(vmax, 1e-3), max_factor * env_max)
        assert dv < bound, f"{what} substep {cp}: |dv| {dv:.2e} m/s, oracle vs itself {ds:.2e} (bound {bound:.2e})"
        assert dvq < max(1e-4 * max(vmax, 1e-3), p_factor * env_p999), f"{what} substep {cp}: 99.9 % of |dv| within {dvq:.2e}, oracle {dsq:.2e}"


def test_s3_one_frame_of_the_reference_cadence_400_substeps(oracle_lib):
    """One frame as the reference's drivers run it -- 400 substeps, the body advected by mesh_x + k dt mesh_v inside the library
    (train_material_params.py:616-626) -- on the full-size garment with collider, mover and swaying body, in four fused calls of
    100, against the OpenMP oracle at every call's end.  x within 1e-4 (also per particle).  v within 2 x the oracle's distance
    from ITSELF under another summation order (measured here: the cloth QR of the HIP path is the oracle's bit for bit, what is
    left is the rounding of the transfers, and the oracle moves by as much when only the order of its atomic adds changes:
    3.6e-3 of the top speed at substep 100, profiles/r03_full_parity_garment-120k-aniso.json)."""
    sc, rows = _follow("garment-120k-aniso", [100, 200, 300, 400])
    assert sc.n_elements == 79600 and sc.n_vertices == 40000
    _check(rows, "S3")


def test_s4_sheet_500k_1000_substeps_north_star_protocol(oracle_lib):
    """BASELINE.json's protocol on the headline workload in the driver-run suite: 497,762 particles, 256^3, 1000 substeps
    against the OpenMP oracle (x and v after N substeps)."""
    sc, rows = _follow("sheet-500k", [100, 300, 600, 1000])
    assert sc.n_particles == 497762 and sc.n_grid == 256
    # the MAXIMUM of |dv| over 500k particles after 1000 substeps of amplified rounding is an extreme-value statistic: round 3 measured
    # 0.5-1.0 x the oracle's self-distance, round 4 one run with 1.84 x (3.4e-3 against 1.9e-3 m/s) and one with 0.66 x.  The maximum
    # keeps 2.5 x here; the 99.9th percentile -- the robust form of the same statement -- is held to 2 x like everything else.
    _check(rows, "S4", max_factor=2.5)


def test_s4_sheet_500k_20_substeps(oracle_lib):
    sc, o, x, v = _pair("sheet-500k", 20, omp=True)
    assert sc.n_particles == 497762 and sc.n_grid == 256
    assert rg.rel(x, o.x) < 1e-4 and rg.rel(v, o.v) < 1e-4
