"""Host side of the finite-difference material fitting (mpmavatar_amd/fd.py; train_material_params.py:650-712): the
difference quotients, the Adam / cosine-annealing update, clamping and the best / last bookkeeping, checked against a
plain re-computation with torch.optim on made-up losses (no GPU)."""
import numpy as np
import torch

from mpmavatar_amd import fd


def _mk(**kw):
    return fd.MaterialFD(None, [], build=False, **kw)


def test_difference_quotients_and_adam_update_match_a_plain_torch_run():
    m = _mk(init=(1.0, 2.0, 1.0), lrs=(0.05, 0.02, 0.005), iterations=50)
    ref = {k: torch.tensor(v, dtype=torch.float32, requires_grad=True) for k, v in zip("DEH", (1.0, 2.0, 1.0))}
    opt = torch.optim.Adam([{"params": [ref[k]], "lr": lr} for k, lr in zip("DEH", (0.05, 0.02, 0.005))], lr=0.05)
    sch = torch.optim.lr_scheduler.CosineAnnealingLR(opt, 50 * 0.5 * np.pi / np.arccos(0.4), eta_min=0.0)
    rng = np.random.default_rng(0)
    for it in range(6):
        L = rng.uniform(1e-4, 2e-4, 4).tolist()
        out = m.apply_losses(L)
        opt.zero_grad()
        ref["D"].grad = torch.tensor((L[1] - L[0]) / 0.05).float()
        ref["E"].grad = torch.tensor((L[2] - L[0]) / 0.05).float()
        ref["H"].grad = torch.tensor((L[3] - L[0]) / 0.005).float()
        opt.step(); sch.step()
        for k in "DEH":
            assert m.torch_param[k].item() == ref[k].item()
        assert out["loss"] == L[0] and m.last_params["step"] == it
        assert m.last_params["E"] == ref["E"].item() * 100   # checkpoints store Young's modulus, the parameter is E / 100
    assert m.step == 6


def test_parameters_are_clamped_to_their_ranges_and_best_tracks_the_lowest_loss():
    m = _mk(init=(1.0, 1.0, 1.0), ranges=((0.98, 1.02), (0.5, 1.01), (0.999, 1.001)), lrs=(0.5, 0.5, 0.5))
    m.apply_losses([0.5, 0.9, 0.1, 0.9])     # dL/dD > 0, dL/dE < 0, dL/dH > 0: Adam's first step moves by lr
    assert m.torch_param["D"].item() == np.float32(0.98)
    assert m.torch_param["E"].item() == np.float32(1.01)
    assert m.torch_param["H"].item() == np.float32(0.999)
    assert m.best_params["loss"] == 0.5 and m.best_params["step"] == 0
    m.apply_losses([0.7, 0.7, 0.7, 0.7])
    assert m.best_params["loss"] == 0.5 and m.last_params["loss"] == 0.7 and m.last_params["step"] == 1
    m.apply_losses([0.2, 0.2, 0.2, 0.2])
    assert m.best_params["loss"] == 0.2 and m.best_params["step"] == 2


def test_variant_slices_cover_the_four_runs_once():
    for world in (1, 2, 3, 4, 8):
        got = [i for r in range(world) for i in fd.variant_slice(r, world)]
        assert got == [0, 1, 2, 3]
    assert list(fd.variant_slice(1, 2)) == [2, 3]
    assert fd.DELTAS == ((0.0, 0.0, 0.0), (0.05, 0.0, 0.0), (0.0, 0.05, 0.0), (0.0, 0.0, 0.005))


def test_sharded_training_step_matches_the_single_process_gloo():
    """world 2 (CPU, gloo): each rank evaluates its two variants, all-gather of the losses, identical Adam update."""
    import os
    from launch import torchrun
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = torchrun(2, os.path.join(root, "tests", "fd_host_worker.py"), env=dict(os.environ, OMP_NUM_THREADS="1"), timeout=600, cwd=root)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
