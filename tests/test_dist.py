"""Multi-process tests of the slab decomposition (mpmavatar_amd/dist.py).

CPU (gloo, world_size 2 and 3): partition invariants, matching exchange lists, halo-sum property with the float64
twin.  GPU (two processes sharing cuda:0, gloo transport): the full sharded substep against a single context.
"""
import os
import sys

import numpy as np
import pytest

from launch import torchrun

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
WORKER = os.path.join(ROOT, "tests", "dist_worker.py")


def _launch(nproc, *args, timeout=600, extra_env=None):
    env = dict(os.environ, OMP_NUM_THREADS="1", HSA_ENABLE_IPC_MODE_LEGACY="0", **(extra_env or {}))
    r = torchrun(nproc, WORKER, args, env=env, timeout=timeout)   # (tests/launch.py: a lost race for the port is retried, nothing else)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    return r.stdout


@pytest.mark.parametrize("world", [2, 3])
@pytest.mark.parametrize("scene", ["garment", "demo"])
def test_partition_and_halo_sum_cpu(world, scene, oracle_lib):
    _launch(world, "cpu", scene, 0)


@pytest.mark.gpu
@pytest.mark.parametrize("ghost_g2p", ["1", "0"])  # 1: copies gather for themselves; 0: owners send every substep
@pytest.mark.parametrize("scene,steps", [("garment", 40), ("sheet", 40), ("demo", 30), ("cube", 30)])
def test_sharded_matches_single_context(scene, steps, ghost_g2p):
    out = _launch(2, "gpu", scene, steps, extra_env={"MPMHIP_DIST_GHOST_G2P": ghost_g2p})
    assert "max rel dx" in out


@pytest.mark.gpu
def test_sharded_body_posed_per_frame():
    """Scene.mesh_sway (the S3 stand-in's swaying body: new pose and velocity every frame, joints riding on it) through the
    sharded driver's Python-driven transport: runs are split at the frame boundaries like harness.run does."""
    out = _launch(2, "gpu", "sway", 70)
    assert "max rel dx" in out


@pytest.mark.gpu
@pytest.mark.parametrize("scene", ["sheet", "demo"])
def test_sharded_three_ranks(scene):
    """Three ranks (interior rank with two peers: multi-peer pack / add tables; halo sums are no longer a single
    commutative pair, so owner and ghost copy may differ in the last bit until the next re-synchronisation)."""
    out = _launch(3, "gpu", scene, 60)
    assert "max rel dx" in out


@pytest.mark.gpu
def test_sharded_ghost_resync_at_resort():
    """More substeps than the collective re-sort interval (8 in the worker): the copies are overwritten by their
    owners' values before each re-sort and keep gathering for themselves in between."""
    out = _launch(2, "gpu", "sheet", 100)
    assert "max rel dx" in out


@pytest.mark.gpu
@pytest.mark.parametrize("scene,steps", [("fastcube", 200), ("sheet", 60)])
def test_sharded_adaptive_collective_resort(scene, steps):
    """rebin_interval = 0: the ranks re-sort together when the max-reduced early-warning drift flag asks for it (polled
    every 16 substeps).  The thrown cube needs several re-sorts in 200 substeps, the resting sheet only the first one;
    both must match the single context and drop nothing."""
    import re
    out = _launch(2, "gpu", scene, steps, extra_env={"MPMHIP_TEST_REBIN": "0"})
    assert "max rel dx" in out
    n = [int(x) for x in re.findall(r"rank \d+: (\d+) collective re-sorts", out)]
    assert len(n) == 2 and n[0] == n[1]                 # every rank took the same decisions
    assert (n[0] >= 3) if scene == "fastcube" else (n[0] <= 2)


@pytest.mark.gpu
def test_in_library_rccl_transport_single_rank_adaptive():
    """The same policy inside the library's RCCL loop (ncclAllReduce of the flag, read back with a lag), world size 1."""
    env = dict(os.environ, MPMHIP_DIST_TRANSPORT="rccl", MPMHIP_TEST_REBIN="0")
    r = torchrun(1, WORKER, ["gpu", "fastcube", "200"], env=env, timeout=600)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    assert "max rel dx" in r.stdout


@pytest.mark.gpu
@pytest.mark.parametrize("transport", ["rccl", "torch"])
def test_nccl_backend_world_one_runs_every_collective(transport):
    """torch.distributed with the NCCL (= RCCL) backend, as bench.py --gpus N initialises it, at world size 1: every
    collective the sharded driver issues (all-reduce of counts and flags, all-gather of block maps and of the state,
    object broadcast) runs with device tensors.  Found this way: a float64 CPU tensor in an all-reduce that only ran for
    world > 1."""
    env = dict(os.environ, MPMHIP_DIST_TRANSPORT=transport, MPMHIP_TEST_BACKEND="nccl", MPMHIP_TEST_REBIN="0",
               HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = torchrun(1, WORKER, ["gpu", "garment", "40"], env=env, timeout=600)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    assert "max rel dx" in r.stdout and f"({transport}" in r.stdout


@pytest.mark.gpu
def test_in_library_rccl_transport_single_rank():
    """World size 1 through the library's own RCCL communicator (dlopen, ncclCommInitRank, ncclAllGather of the block
    map, empty send/recv groups) -- the multi-rank send/recv itself cannot run on a one-GPU box."""
    env = dict(os.environ, MPMHIP_DIST_TRANSPORT="rccl")
    r = torchrun(1, WORKER, ["gpu", "garment", "40"], env=env, timeout=600)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    assert "max rel dx" in r.stdout


@pytest.mark.gpu
@pytest.mark.parametrize("scene,steps", [("garment", 40), ("sheet", 60), ("demo", 30)])
def test_in_library_rccl_transport_two_ranks(scene, steps):
    """The RCCL send/recv group between REAL peers (dist.hip: fast_rccl_steps): needs two GPUs, skipped on a one-GPU box
    (RCCL refuses two ranks on one device) -- the first multi-GPU machine that sees this repository runs it."""
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("needs >= 2 GPUs")
    env = dict(os.environ, MPMHIP_DIST_TRANSPORT="rccl", HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = torchrun(2, WORKER, ["gpu", scene, str(steps)], env=env, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    assert "max rel dx" in r.stdout


def _mock_rccl_env(**kw):
    """MPMHIP_RCCL_LIB -> tests/mock_rccl (shared-memory stand-in for the ten RCCL entry points the library binds)."""
    sys.path.insert(0, os.path.join(ROOT, "tests", "mock_rccl"))
    from build import build as build_mock
    return dict(MPMHIP_DIST_TRANSPORT="rccl", MPMHIP_RCCL_LIB=build_mock(), **{k: str(v) for k, v in kw.items()})


@pytest.mark.gpu
@pytest.mark.parametrize("halo", ["peer", "peer-unfused", "rccl"])
@pytest.mark.parametrize("world,scene,steps,rebin,ghost_g2p", [
    (2, "garment", 60, 8, 1), (2, "garment", 40, 8, 0), (2, "sheet", 100, 8, 1), (3, "sheet", 60, 8, 1), (3, "demo", 60, 8, 1),
    (2, "demo", 30, 8, 0), (2, "cube", 30, 8, 1), (2, "fastcube", 200, 0, 1), (2, "sheet", 60, 0, 1), (2, "demohold", 60, 8, 1),
    (2, "sway", 70, 8, 1), (4, "demo", 40, 8, 1)])
def test_in_library_loop_with_several_ranks(world, scene, steps, rebin, ghost_g2p, halo):
    """`mpmhip_rccl_steps` -- the loop bench.py --gpus N runs -- with 2 and 3 ranks on ONE GPU: the library binds RCCL by
    dlsym, and MPMHIP_RCCL_LIB points it at a stand-in that moves the same messages through shared memory (real RCCL
    refuses two ranks on one device).  Covers what a one-GPU box otherwise never executes: the all-gather of the block
    maps, the shared-block lists, the halo send/recv group (sizes checked pairwise by the stand-in), ghost
    re-synchronisation at collective re-sorts, the all-reduced drift flag (rebin 0), staged release.  Result: the single
    context's trajectory.
    halo = "peer": the halos go through peer-mapped buffers (HIP IPC between the processes, flags in the receiver's
    fine-grained memory, handshake at set-up) and the substep has NO halo kernels: the pack rides in the p2g launch, g2p adds
    the neighbour's share while it stages its tile (PackArgs / HaloIn in csrc/fast_device.hpp); "peer-unfused": the same buffers with
    the separate pack / add kernels (MPMHIP_DIST_FUSED_HALO=0); "rccl": through the send/recv groups."""
    import re
    out = _launch(world, "gpu", scene, steps, extra_env=_mock_rccl_env(MPMHIP_TEST_REBIN=rebin, MPMHIP_DIST_GHOST_G2P=ghost_g2p,
                                                                        MPMHIP_DIST_HALO=halo.split("-")[0], MPMHIP_VERBOSE=1,
                                                                        MPMHIP_DIST_FUSED_HALO=0 if halo == "peer-unfused" else 1))
    assert "max rel dx" in out and "(rccl, halos: " in out
    assert out.count("halos: peer-mapped" if halo.startswith("peer") else "halos: send/recv") == world, out[-2000:]
    fused = [int(x) for x in re.findall(r"fused halo substeps (\d+)", out)]
    if halo == "peer" and world == 2:
        assert len(fused) == world and min(fused) > 0, out[-2000:]   # the fused path really ran on every rank
        # (three and more ranks on these toy scenes have slabs thinner than two blocks: a block shared with two neighbours keeps
        # the add kernel with its atomics for that interval -- the fallback the library documents)
    elif halo != "peer":
        assert not fused or max(fused) == 0
    n = [int(x) for x in re.findall(r"rank \d+: (\d+) collective re-sorts", out)]
    assert len(n) == world and len(set(n)) == 1          # every rank took the same decisions
    if rebin == 0:
        assert (n[0] >= 3) if scene == "fastcube" else (n[0] <= 2)


@pytest.mark.gpu
@pytest.mark.parametrize("world,halo", [(4, "peer"), (4, "rccl"), (8, "rccl")])
def test_in_library_loop_with_four_and_eight_ranks(world, halo):
    """What `bench.py --gpus 4` / `--gpus 8` runs on a node -- `mpmhip_rccl_steps`, one process per rank -- with 4 and 8 ranks over the
    RCCL stand-in on ONE GPU (the first contact with a real node must not be the first time the loop sees eight
    ranks): interior ranks with two neighbours each, eight communicator members, 7 + 7 peer links, collective re-sorts every 8
    substeps.  Result: the single context's trajectory; every rank takes the same re-sort decisions.
    (Eight ranks run the send/recv halos only: with peer-mapped halos a rank's kernels spin on a flag its neighbour's kernel raises,
    and eight processes time-slicing ONE GPU do not get the neighbour scheduled inside the flag timeout -- "a peer-mapped halo never
    arrived", measured; on a node every rank has its own GPU.  Four ranks sharing the GPU do.)"""
    import re
    out = _launch(world, "gpu", "widesheet8" if world == 8 else "widesheet", 40,
                  extra_env=_mock_rccl_env(MPMHIP_TEST_REBIN=8, MPMHIP_DIST_HALO=halo, MPMHIP_VERBOSE=1))
    assert "max rel dx" in out and "(rccl, halos: " in out
    assert out.count("halos: peer-mapped" if halo == "peer" else "halos: send/recv") == world, out[-2000:]
    n = [int(x) for x in re.findall(r"rank \d+: (\d+) collective re-sorts", out)]
    assert len(n) == world and len(set(n)) == 1
    if halo == "peer":   # slabs of more than four blocks: every shared block has ONE neighbour rank, the fused halo path runs on every rank
        fused = [int(x) for x in re.findall(r"fused halo substeps (\d+)", out)]
        assert len(fused) == world and min(fused) > 0, out[-2000:]


@pytest.mark.gpu
@pytest.mark.parametrize("halo", ["peer", "rccl"])
def test_s4_sheet_500k_full_size_through_the_in_library_loop_two_ranks(halo, oracle_lib):
    """BASELINE.json config 4, the N > 1 leg at FULL size (497,762 particles, 256^3) in the driver-run suite: two
    ranks of `mpmhip_rccl_steps` -- the loop `bench.py --gpus 2` runs (reference loop: mpm_solver.py:229-536) -- over the RCCL stand-in
    on the one GPU, 20 substeps; x and v of every particle, assembled from its owning rank, within 1e-6 of the single context and
    within the north star's 1e-4 of the OpenMP oracle."""
    import re
    out = _launch(2, "gpu", "sheet-500k", 20, timeout=1200,
                  extra_env=_mock_rccl_env(MPMHIP_TEST_REBIN=0, MPMHIP_DIST_HALO=halo, MPMHIP_VERBOSE=1, MPMHIP_TEST_FULL=1))
    m = re.search(r"FULL world=2 steps=20 n_particles=(\d+) n_grid=(\d+): vs single context rel dx (\S+) rel dv (\S+); vs oracle rel dx (\S+) rel dv (\S+)", out)
    assert m, out[-3000:]
    print(m.group(0))
    assert int(m.group(1)) == 497762 and int(m.group(2)) == 256
    assert out.count("halos: peer-mapped" if halo == "peer" else "halos: send/recv") == 2, out[-2000:]


@pytest.mark.gpu
def test_peer_link_failure_on_one_rank_sends_every_rank_back_to_send_recv():
    """The decision for peer-mapped halos is collective: rank 1 reports that its links failed (MPMHIP_LINK_FAULT), and
    all three ranks keep their halos on the send/recv groups -- with the same result."""
    out = _launch(3, "gpu", "sheet", 40, extra_env=_mock_rccl_env(MPMHIP_DIST_HALO="peer", MPMHIP_LINK_FAULT=1))
    assert "max rel dx" in out and out.count("halos: send/recv") == 3, out[-2000:]


@pytest.mark.gpu
def test_in_library_loop_migration_over_the_stand_in():
    import re
    out = _launch(2, "gpu", "crossing", 150, extra_env=_mock_rccl_env(MPMHIP_TEST_MIGRATE=0.1, MPMHIP_TEST_RUN_CHUNK=30))
    assert "max rel dx" in out and "(rccl, halos: peer-mapped" in out
    n = [int(x) for x in re.findall(r"rank \d+: (\d+) re-partitions", out)]
    assert len(n) == 2 and n[0] == n[1] and n[0] >= 1


@pytest.mark.gpu
@pytest.mark.parametrize("world,transport", [(2, "torch"), (2, "rccl"), (3, "rccl"), (4, "rccl")])
def test_on_device_migration_of_traditional_particles(world, transport):
    """material that MIXES across the slab cuts changes rank on the device (mpmavatar_amd.dist.migrate_traditional):
    new quantile cuts from a histogram summed over the ranks, the leavers' records (52 floats + id) through one all-to-all of device
    tensors into free rows of the destination's traditional class, no context rebuilt, no state through the host, no
    all_gather_object.  A sheared cube on 2 / 3 / 4 ranks, a look every 30 substeps: several migration events, no stop-the-world
    re-partition, ownership stays a partition of the particles, and the run stays on the single context's trajectory."""
    import re
    env = {"MPMHIP_TEST_TRAD_MIG": "0", "MPMHIP_TEST_RUN_CHUNK": "30"}
    if transport == "rccl":
        env.update(_mock_rccl_env())
    out = _launch(world, "gpu", "shear", 150, extra_env=env)
    assert "max rel dx" in out
    m = re.findall(r"rank \d+: (\d+) on-device migrations moved (\d+) particles, (\d+) re-partitions, owns (\d+) of (\d+), (\d+) free rows", out)
    assert len(m) == world, out[-3000:]
    ev, moved, rep, owns, total, free = (np.array([int(r[k]) for r in m]) for k in range(6))
    assert (ev >= 2).all() and len(set(ev)) == 1 and (moved > 0).all() and len(set(moved)) == 1 and (rep == 0).all(), m
    assert owns.sum() == total[0] and (free > 0).all(), m                       # still a partition; nobody ran out of rows
    assert owns.max() - owns.min() <= 0.1 * total[0] / world + 8, m             # ... and a balanced one (quantile cuts)


@pytest.mark.gpu
@pytest.mark.parametrize("transport", ["torch", "rccl"])
def test_a_body_that_only_moves_is_not_re_partitioned(transport):
    """the reference's drivers move the body -- and the garment with it -- for hundreds of frames, and round 5
    re-partitioned (stop-the-world, ~1 s) whenever 10 % of the particles had left the x-slab they were cut by.  Ownership is a matter
    of performance only; what costs is the halo, and a cube thrown along x carries its cut with it: with the production criterion
    (halo grown by 1.5 x AND 10 % of the particles outside their slab) it crosses the slab boundary entirely -- most of its particles
    outside "their" slab -- without one re-partition, without one more byte of halo, on the single context's trajectory."""
    import re
    env = {"MPMHIP_TEST_MIGRATE": "0.1", "MPMHIP_TEST_HALO_FACTOR": "1.5", "MPMHIP_TEST_RUN_CHUNK": "30"}
    if transport == "rccl":
        env.update(_mock_rccl_env())
    out = _launch(2, "gpu", "crossing", 150, extra_env=env)
    assert "max rel dx" in out
    m = re.findall(r"rank \d+: (\d+) re-partitions, (\d+) % outside their slab, halo x([0-9.]+)", out)
    assert len(m) == 2, out[-2000:]
    assert all(int(n) == 0 for n, _, _ in m) and all(int(fr) >= 10 for _, fr, _ in m) and all(float(h) < 1.5 for _, _, h in m), m


@pytest.mark.gpu
@pytest.mark.parametrize("bad", [0, 2])
def test_a_rank_whose_local_build_fails_takes_every_rank_out_cleanly(bad):
    """build_sharded votes on the local part of the build BEFORE its first collective.  Rank `bad` fails there (injected);
    all three ranks raise the same collective RuntimeError -- the failing one with its cause, the others with "another rank" -- nobody
    waits in a collective the failed rank never enters, and the process group works afterwards."""
    out = _launch(3, "gpu", "sheet", 10, timeout=300, extra_env={"MPMHIP_TEST_FAIL_BUILD_RANK": str(bad)})
    assert out.count("collective build error") == 3, out[-2000:]
    assert out.count("on this rank") == 1 and out.count("another rank") == 2, out[-2000:]


@pytest.mark.gpu
def test_one_tile_decision_for_all_ranks_from_the_bound_masses():
    """MPMHIP_P2G_TILE_AUTO in a sharded run.  Rank 1's particles are made 1e7 times heavier AFTER the build
    (reset_density(update_mass=True)): rank 0's own masses still span a factor of one, the scene's 1e7.  Both ranks must run the fp64
    tile from the next import on -- the span is measured on the bound mass tensors and all-reduced (mpmhip_dist_set_mass_span),
    not read off the scene description, and AUTO is never forced to the fixed-point tile."""
    out = _launch(2, "gpu", "cube", 4, extra_env={"MPMHIP_TEST_HEAVY_RANK": "1"})
    assert out.count("p2g tile in use 2") == 2, out[-2000:]


@pytest.mark.gpu
def test_sharded_staged_sand_release():
    """run_demo.py:524 in the sharded driver: each rank holds its share (a suffix of its owned traditional particles) of the
    sand the mover still pins, and lets go of it on the global schedule.  Must match the single context, which is checked
    against the oracle in test_gpu_parity.py::test_staged_sand_release."""
    out = _launch(2, "gpu", "demohold", 60)
    assert "max rel dx" in out


@pytest.mark.gpu
def test_sharded_particle_migration():
    """A cube thrown along x: more than 10 % of the particles leave the slab of the rank that owns them, the ranks gather
    their state, cut new slabs at the current positions and rebuild their shards (mpmavatar_amd.dist.repartition) -- and the
    run continues on the same trajectory as a single context."""
    import re
    out = _launch(2, "gpu", "crossing", 150, extra_env={"MPMHIP_TEST_MIGRATE": "0.1", "MPMHIP_TEST_RUN_CHUNK": "30"})
    assert "max rel dx" in out
    n = [int(x) for x in re.findall(r"rank \d+: (\d+) re-partitions", out)]
    assert len(n) == 2 and n[0] == n[1] >= 1, out[-2000:]
