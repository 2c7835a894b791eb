"""p2g and g2p of a cloth substep as ONE launch behind a phase gate (round 6: k_p2g_g2p, csrc/p2g.hip; PhaseGate, csrc/fast_device.hpp).

The gather workgroups ride in the scatter's launch and wait -- per wavefront, on a flag the last scattering workgroup raises -- until
every contribution of the substep is in; accumulators, collider and mover channels are then read at agent scope.  Same device code for
the two halves as the two launches (`mpmhip_set_debug_flags(32)` selects those), so: merged = two launches = the oracle, on scenes
with a body collider, a mover and re-sorts in the window; and the merged form must really have run.  (Particles outside their tile
margin inside the merged launch: tests/test_gpu_parity.py::test_out_of_margin_paths[sheet / garment] run it, it is the default form.)"""
import numpy as np
import pytest

from mpmavatar_amd import harness, scenes

pytestmark = pytest.mark.gpu


def _run(sc, n, unmerged):
    sim = harness.build_solver(sc, "cuda:0", mode="fast")
    if unmerged:
        sim.solver._call("mpmhip_set_debug_flags", 32)
    harness.run(sim, n, fused=True)
    st = sim.solver.stats()
    assert st["n_dropped"] == 0
    return sim.state.particle_x.cpu().numpy(), sim.state.particle_v.cpu().numpy(), st


@pytest.mark.parametrize("make,n", [(scenes.small_sheet, 120), (scenes.small_garment, 80),
                                    (lambda: scenes.sheet(n=96, n_grid=128, collider_subdiv=3, n_steps=100, span=(0.4, 1.6), y=1.22, sphere_r=0.2,
                                                          sphere_c=(1.0, 0.98, 1.0), name="sheet-96x96"), 150)])
def test_merged_launch_equals_two_launches_equals_oracle(make, n, oracle_lib):
    from oracle.scene_adapter import oracle_from_scene, run_scene
    sc = make()
    xm, vm, sm = _run(make(), n, unmerged=False)
    xu, vu, su = _run(make(), n, unmerged=True)
    assert sm["merged_launches"] == n and su["merged_launches"] == 0      # the form under test ran for every substep, the other never
    rel = lambda a, b: float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-3))
    # (the two forms add the same tile sums into the grid in different orders: fp32 atomics, last-bit differences)
    assert rel(xm, xu) < 1e-6 and rel(vm, vu) < (1e-4 if sc.gamma > 0 else 1e-5), (rel(xm, xu), rel(vm, vu))
    o = oracle_from_scene(sc)
    run_scene(o, sc, n)
    assert rel(xm, o.x) < 1e-5 and rel(vm, o.v) < (1e-3 if sc.gamma > 0 else 1e-4), (rel(xm, o.x), rel(vm, o.v))
