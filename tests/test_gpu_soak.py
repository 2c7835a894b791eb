"""The adaptive re-sort policy over thousands of fused substeps at full size: the early warning (10 substeps of look-ahead,
flag and progress through host-mapped memory, host at most six substeps ahead) must bring every re-sort in time -- nothing
dropped, nobody on the out-of-margin path -- while the scenes move at metres per second (tools/gpu/soak.py is the long form)."""
import pytest
import torch

from mpmavatar_amd import harness, scenes

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name,substeps", [("sheet-500k", 3000), ("block-512k", 5000), ("demo-250", 3000), ("garment-120k-aniso", 3000)])
def test_resorts_arrive_in_time(name, substeps):
    sim = harness.build_solver(scenes.REGISTRY[name](), "cuda:0")
    harness.run(sim, substeps, fused=True)
    st = sim.solver.stats()
    x, v = sim.state.particle_x, sim.state.particle_v
    assert bool(torch.isfinite(x).all() and torch.isfinite(v).all())
    assert float(x.min()) >= 0.0 and float(x.max()) <= sim.scene.grid_lim
    assert st["n_dropped"] == 0
    # cumulative particle-substeps on the slow path: 0 in every recorded run (profiles/archive/r02_soak.txt); the bound leaves room
    # for a handful of strongly accelerated particles (the path is correct, only slow) out of ~1e9 particle-substeps
    assert st["n_fallback_particles"] <= 1000, st
    assert st["rebins"] >= 2                            # the scene did move through its tiles
