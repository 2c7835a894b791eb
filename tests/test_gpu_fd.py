"""Finite-difference material fitting on the device (mpmavatar_amd/fd.py, train_material_params.py:575-671): the four
FD simulations on four solver contexts / streams / host threads against the reference's sequential order, the loss of a
captured sequence at its own parameters, and a few Adam steps."""
import numpy as np
import pytest
import torch

from mpmavatar_amd import fd, scenes

pytestmark = pytest.mark.gpu


def _problem(concurrent, init=(1.0, 1.0, 1.0)):
    sc = scenes.garment_cylinder(n_theta=24, n_h=12, n_grid=48, aniso=True, collider_subdiv=2)
    frames = fd.synthetic_problem(sc, n_frames=3, frame_dt=30e-4)
    return fd.MaterialFD(sc, frames, init=init, lrs=(0.05, 0.05, 0.005), iterations=20, frame_dt=30e-4, substeps=30, scale=0.8,
                         shift=(0.2, 0.1, 0.3), concurrent=concurrent)


def test_concurrent_variants_match_the_sequential_order():
    a, b = _problem(True), _problem(False)
    fd.capture(a, 1.0, 1.0, 1.0)
    for f, g, fr in zip(b._frames, a._frames, a.frames):
        f["target"] = g["target"].clone()
    la, lb = a.losses(1.3, 0.8, 1.0), b.losses(1.3, 0.8, 1.0)
    assert len(la) == 4 and all(np.isfinite(la)) and la[0] > 0
    np.testing.assert_allclose(la, lb, rtol=2e-3)          # cloth velocities decorrelate at 1e-3; same physics
    assert len({round(x / la[0], 6) for x in la}) > 1      # the nudged parameters really changed the runs
    assert a.substeps_done == 4 * 3 * 30
    a.close(); b.close()


def test_loss_vanishes_at_the_captured_parameters_and_training_reduces_it():
    # a few milliseconds of motion say little about density or stiffness; the rest-pose scale H acts at once
    m = _problem(True, init=(1.0, 1.0, 1.04))
    fd.capture(m, 1.0, 1.0, 1.0)
    at_truth = m.losses(1.0, 1.0, 1.0)[0]
    first = m.train_one_step()
    assert first["loss"] > 0 and at_truth < 1e-3 * first["loss"]
    assert first["grad"]["H"] > 0.0                        # too long a rest pose: the loss grows with H
    for _ in range(5):
        out = m.train_one_step()
    assert abs(m.torch_param["H"].item() - 1.0) < 0.04 - 0.015   # Adam moves ~lr = 0.005 per step towards the captured value
    assert out["loss"] < first["loss"] and m.best_params["loss"] < first["loss"]
    assert m.last_params["step"] == 5 and m.step == 6
    m.close()


@pytest.mark.parametrize("world", [2, 4])
def test_fd_variants_sharded_over_ranks(world):
    """The four finite-difference simulations are the natural multi-GPU shard of the reference's training step
    (train_material_params.py:583: independent runs): ranks simulate their slice, all-gather four floats, apply the same
    update.  Here the ranks share the GPU (gloo); on a node each takes its own."""
    import os
    import socket
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(root, "tests", "fd_worker.py")]
    r = subprocess.run(cmd, cwd=root, env=dict(os.environ, OMP_NUM_THREADS="1"), capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    assert f"fd sharded over {world} ranks" in r.stdout
