#!/usr/bin/env python
"""One-substep consistency of the HIP path and the oracle from IDENTICAL inputs at full size: how many elements
take a different return-mapping branch, and how far apart are the particle velocities inside / outside their neighbourhood?
Same code as tests/test_gpu_branch_flips.py, on garment-120k-aniso (and sheet-500k with an argument), OpenMP oracle.
    python tools/gpu/branch_flips.py [scene] [k0] [reps] [stride]"""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
import numpy as np
from mpmavatar_amd import harness, scenes
from oracle.scene_adapter import oracle_from_scene, omp_threads, run_scene
from test_gpu_branch_flips import one_substep_from_identical_inputs, sync_oracle_to_hip

name = sys.argv[1] if len(sys.argv) > 1 else "garment-120k-aniso"
k0, reps, stride = (int(sys.argv[i]) if len(sys.argv) > i else d for i, d in ((2, 40), (3, 30), (4, 10)))
sc = scenes.REGISTRY[name]()
sim = harness.build_solver(sc, "cuda:0", mode="fast")
o = oracle_from_scene(sc, omp=True, n_threads=omp_threads())
free = oracle_from_scene(scenes.REGISTRY[name](), omp=True, n_threads=omp_threads())   # runs freely beside, for the accumulated distance
harness.run(sim, k0, fused=True)
run_scene(free, sc, k0)
k = k0
rows = []
for r in range(reps):
    n_f, out, ins, vmax, n_ring = one_substep_from_identical_inputs(sc, sim, o, k)
    run_scene(free, sc, 1, k0=k)
    k += 1
    acc = float(np.linalg.norm(sim.state.particle_v.cpu().numpy() - free.v, axis=1).max())
    rows.append({"substep": k, "flipped": n_f, "ring": n_ring, "dv_outside": out / vmax, "dv_inside": ins / vmax, "accumulated_dv": acc / vmax})
    print(json.dumps(rows[-1]), flush=True)
    if stride > 1:
        harness.run(sim, stride - 1, fused=True)
        run_scene(free, sc, stride - 1, k0=k)
        k += stride - 1
os.makedirs("gpurun_out", exist_ok=True)
json.dump({"scene": name, "rows": rows}, open(f"gpurun_out/r04_branch_flips_{name}.json", "w"), indent=1)
