"""The oracle's ACTIVE BOX option (oracle/mpm_oracle.h; a speed option of the long ensemble runs of tests/test_gpu_fullsize.py, not part
of the restated algorithm): every grid-wide pass of a substep visits only the bounding box of the particles' stencils.  Particle results
must be bit-identical to the dense passes in the serial build, whatever the scene holds (collider faces outside the box, mover, BCs,
staged release, a body that moves), and the box must follow the particles."""
import numpy as np
import pytest

from mpmavatar_amd import scenes
from oracle.scene_adapter import oracle_from_scene, run_scene

SCENES = {
    "garment": (scenes.small_garment, 30),                                    # collider + mover + swaying body
    "sheet": (scenes.small_sheet, 40),                                        # sheet falling on a sphere
    "demo": (lambda: scenes.demo_mix(n_grid=48, n_sheet=16, sand=(16, 3, 8), hold=(10, 5, 64)), 40),   # cloth + sand, floor, staged release
    "cube": (scenes.small_cube, 30),
}
FIELDS = ("x", "v", "C", "F", "F_trial", "stress", "d")


@pytest.mark.parametrize("name", sorted(SCENES))
def test_box_passes_give_the_dense_passes_particles_bit_for_bit(name, oracle_lib):
    make, n = SCENES[name]
    dense, box = oracle_from_scene(make()), oracle_from_scene(make())
    box.sim.box_mode = 1
    sc = make()
    for k0 in range(0, n, 10):
        run_scene(dense, sc, 10, k0=k0)
        run_scene(box, sc, 10, k0=k0)
        for f in FIELDS:
            assert np.array_equal(getattr(dense, f), getattr(box, f)), (name, k0, f)
    lo, hi = np.array(box.sim.box_lo[:]), np.array(box.sim.box_hi[:])
    assert (lo >= 0).all() and (hi < sc.n_grid).all() and (hi > lo).all()
    assert np.prod(hi - lo + 1) < sc.n_grid ** 3                               # ... and it IS a box, not the grid
    # the box holds every particle's stencil at the substep it was taken for (positions have moved one substep since)
    base = (box.x * box.sim.inv_dx - 0.5).astype(np.int32)
    assert (base.min(0) >= lo - 1).all() and (base.max(0) + 2 <= hi + 1).all()


def test_openmp_build_with_the_box_stays_within_rounding_of_the_dense_run(oracle_lib):
    sc = scenes.small_garment()
    dense, box = oracle_from_scene(sc, omp=True, n_threads=3), oracle_from_scene(scenes.small_garment(), omp=True, n_threads=3)
    box.sim.box_mode = 1
    run_scene(dense, sc, 20)
    run_scene(box, sc, 20)
    assert np.abs(dense.x - box.x).max() < 1e-6 and np.abs(dense.v - box.v).max() < 1e-4 * max(np.abs(dense.v).max(), 1e-3)
