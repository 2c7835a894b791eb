#!/usr/bin/env python
"""Compute floor of one rank of an N-way slab decomposition, measured on ONE GPU: the shard of a middle rank (owned
particles + its ghost copies) run as an ordinary single-GPU scene -- no exchange at all.  An upper bound for the strong
scaling bench.py --gpus N can show (the exchange and the two halo launches come on top).
    python tools/gpu/shard_floor.py [scene]"""
import gc, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from mpmavatar_amd import dist as mdist, harness, scenes

name = sys.argv[1] if len(sys.argv) > 1 else "sheet-500k"
sc = scenes.REGISTRY[name]()
base = None
for world in (1, 2, 4, 8):
    shard = mdist.partition(sc, world)[world // 2]
    lsc = shard.scene
    sim = harness.build_solver(lsc, "cuda:0")
    harness.run(sim, 40, fused=True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    harness.run(sim, 400, fused=True)
    torch.cuda.synchronize()
    us = 1e6 * (time.perf_counter() - t0) / 400
    base = base or us
    sim.solver.enable_profiling(True, fused=True); sim.solver.time_profile.clear()
    harness.run(sim, 100, fused=True)
    sim.solver.enable_profiling(False)
    tp = {k: round(1e3 * sum(v) / len(v), 1) for k, v in sim.solver.time_profile.items() if k in ("compute_stress_from_F_trial", "p2g", "g2p_v")}
    print(f"{name}: rank {world // 2} of {world}: {lsc.n_particles} local particles ({shard.own_e.size + shard.own_t.size + shard.own_v.size} owned), "
          f"{us:.1f} us/substep alone -> at best {base / us:.2f}x of one GPU; launches (event-bracketed) {tp}", flush=True)
    del sim; gc.collect(); torch.cuda.synchronize()
