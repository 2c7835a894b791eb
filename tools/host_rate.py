#!/usr/bin/env python
"""How long does the host need to ENQUEUE a fused run of substeps (mpmhip_steps) compared with the GPU executing it?
usage: python tools/host_rate.py [scene] [steps]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mpmavatar_amd import harness, scenes

name = sys.argv[1] if len(sys.argv) > 1 else "sheet-500k"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 400
sim = harness.build_solver(scenes.REGISTRY[name](), "cuda:0")
harness.run(sim, 64, fused=True)
torch.cuda.synchronize()
for rep in range(3):
    t0 = time.perf_counter()
    harness.run(sim, n, fused=True)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"{name}: enqueue {1e6*(t1-t0)/n:.1f} us/substep, complete {1e6*(t2-t0)/n:.1f} us/substep", flush=True)
