#!/usr/bin/env python
"""Throughput of the finite-difference training step (4 simulations per step, train_material_params.py:583) with the
variants run one after the other on one solver context (the reference's order) and concurrently on four contexts /
streams / host threads.    python tools/gpu/fd_bench.py [n_theta n_h n_grid frames substeps]"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from mpmavatar_amd import fd, scenes
fd.request_hw_queues()   # before the first HIP call of the process (the sequential leg comes first and would initialise the device)

a = [int(x) for x in sys.argv[1:]]
n_theta, n_h, n_grid, n_frames, substeps = (a + [200, 200, 128, 2, 400][len(a):])[:5]
sc = scenes.garment_cylinder(n_theta=n_theta, n_h=n_h, n_grid=n_grid, aniso=True)
frames = fd.synthetic_problem(sc, n_frames=n_frames, frame_dt=substeps * 1e-4)
out = {"scene": sc.name, "n_particles": sc.n_particles, "n_grid": n_grid, "frames": n_frames, "substeps_per_frame": substeps}
for label, conc in (("sequential", False), ("concurrent", True)):
    m = fd.MaterialFD(sc, frames, frame_dt=substeps * 1e-4, substeps=substeps, concurrent=conc)
    fd.capture(m, 1.0, 1.0, 1.0)
    m.losses(1.2, 1.0, 1.0)   # warm-up: first sorts, allocations
    torch.cuda.synchronize(); t0 = time.perf_counter(); n0 = m.substeps_done
    for _ in range(2):
        r = m.train_one_step()
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    out[label] = {"substeps_per_s": (m.substeps_done - n0) / dt, "s_per_training_step": dt / 2, "loss": r["loss"]}
    print(label, out[label], flush=True)
    m.close(); del m
out["speedup"] = out["concurrent"]["substeps_per_s"] / out["sequential"]["substeps_per_s"]
os.makedirs("gpurun_out", exist_ok=True)
json.dump(out, open("gpurun_out/fd_bench.json", "w"), indent=1)
print(json.dumps(out))
