"""Generate the committed golden fixtures (tests/golden/*.npz).

The reference (NVIDIA-Warp DSL) cannot be imported or run in this image and ships no fixtures, so the vectors are
produced by the independent float64 NumPy twin (oracle/twin.py): final particle state of small seeded scenes after a
fixed number of substeps, plus single-kernel vectors.  The fp32 oracle (tests/test_golden.py) and the HIP path
(tests/test_gpu_golden.py) are both compared against these files.  Re-run:  python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from mpmavatar_amd import scenes  # noqa: E402
from oracle.scene_adapter import run_scene  # noqa: E402
from oracle.twin import TwinMPM  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))

CASES = {
    "cube_jelly_100": (lambda: scenes.small_cube(material="jelly"), 100),
    "cube_sand_100": (lambda: scenes.small_cube(material="sand", params={"friction_angle": 40.0}), 100),
    "sheet_1": (lambda: scenes.small_sheet(), 1),
    "sheet_100": (lambda: scenes.small_sheet(), 100),
    "garment_60": (lambda: scenes.small_garment(), 60),
    "demo_60": (lambda: scenes.demo_mix(n_grid=48, n_sheet=16, sand=(16, 3, 8)), 60),
}


def build(name):
    mk, n = CASES[name]
    sc = mk()
    t = TwinMPM(sc)
    run_scene(t, sc, n)
    f32 = lambda a: np.asarray(a, np.float32)
    out = dict(n_steps=np.int32(n), x=f32(t.x), v=f32(t.v), C=f32(t.C), d=f32(t.d), F_trial=f32(t.F_trial))
    if n == 1:
        act = np.nonzero(t.grid_m > 0)[0].astype(np.int32)
        out.update(grid_nodes=act, grid_m=f32(t.grid_m[act]), grid_v_out=f32(t.grid_v_out[act]))
    return out


if __name__ == "__main__":
    for name in CASES:
        np.savez_compressed(os.path.join(HERE, name + ".npz"), **build(name))
        print(name, os.path.getsize(os.path.join(HERE, name + ".npz")) // 1024, "KiB")
