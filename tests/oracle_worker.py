"""One member of the oracle ensemble of tests/test_gpu_fullsize.py (TEST INFRASTRUCTURE): the OpenMP oracle on a registry scene with
a given thread count, velocities and positions written at the checkpoints.  Run as a process of its own so that the K members
(which differ in nothing but the order of their atomic adds) advance side by side on the host's cores while the GPU runs the HIP path.

    python tests/oracle_worker.py <scene> <gamma0: 0|1> <threads>[f] <out.npz> <checkpoint> [<checkpoint> ...]

A checkpoint c means: state after c substeps AND after c + 1 (the HIP side spends one substep on the one-substep-map probe there,
see _follow), stored as v_<c>, x_<c>, v_<c>p, x_<c>p."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    name, gamma0, out = sys.argv[1], sys.argv[2] == "1", sys.argv[4]
    fma, threads = sys.argv[3].endswith("f"), int(sys.argv[3].rstrip("f"))   # "<threads>" or "<threads>f" = the FMA-contracted build
    cps = [int(a) for a in sys.argv[5:]]
    from mpmavatar_amd import scenes
    from oracle.scene_adapter import oracle_from_scene, run_scene
    sc = scenes.REGISTRY[name]()
    if gamma0:
        sc.gamma = 0.0
    o = oracle_from_scene(sc, omp=True, n_threads=threads, fma=fma)
    o.sim.box_mode = 1   # grid-wide passes over the particles' bounding box only: same particles bit for bit (tests/test_oracle_box.py), 40 % less time
    res, done, t0 = {}, 0, time.time()
    for cp in cps:
        run_scene(o, sc, cp - done, k0=done)
        res[f"x_{cp}"], res[f"v_{cp}"] = o.x.copy(), o.v.copy()
        run_scene(o, sc, 1, k0=cp)
        done = cp + 1
    res["seconds"] = np.array(time.time() - t0)
    np.savez(out, **res)


if __name__ == "__main__":
    main()
