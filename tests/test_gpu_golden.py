"""The HIP solver (both modes, through the C ABI) against the committed golden fixtures."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
import importlib.util
_spec = importlib.util.spec_from_file_location("make_golden", os.path.join(HERE, "golden", "make_golden.py"))
make_golden = importlib.util.module_from_spec(_spec)
_spec.loader.exec_module(make_golden)
from test_golden import TOL, rel  # noqa: E402


@pytest.mark.parametrize("mode", ["fast", "baseline"])
@pytest.mark.parametrize("name", sorted(make_golden.CASES))
def test_hip_matches_golden(name, mode):
    from mpmavatar_amd import harness
    g = np.load(os.path.join(HERE, "golden", name + ".npz"))
    mk, n = make_golden.CASES[name]
    sc = mk()
    sim = harness.build_solver(sc, "cuda:0", mode=mode)
    harness.run(sim, n, fused=(mode == "fast"))
    x = sim.state.particle_x.cpu().numpy()
    v = sim.state.particle_v.cpu().numpy()
    tx, tv = TOL[name]
    assert np.isfinite(x).all()
    assert rel(x, g["x"]) < tx
    assert rel(v, g["v"]) < tv
    st = sim.solver.stats()
    assert st.get("n_fallback_particles", 0) >= 0
