#!/usr/bin/env python
"""Where does a SHORT timed window (bench.py --steps 20) lose time?  Host-side stage times of `sync; run(K); sync`
and the GPU-side span of the same window (HIP events).   python tools/gpu/window_overhead.py [scene] [K]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from mpmavatar_amd import harness, scenes

name = sys.argv[1] if len(sys.argv) > 1 else "sheet-500k"
K = int(sys.argv[2]) if len(sys.argv) > 2 else 20
sim = harness.build_solver(scenes.REGISTRY[name](), "cuda:0")
sv = sim.solver
harness.run(sim, 40, fused=True)
torch.cuda.synchronize()
orig = sv._call
stamps = {}
def spy(fn, *a):
    if fn == "mpmhip_steps":
        stamps["enter"] = time.perf_counter()
        r = orig(fn, *a)
        stamps["leave"] = time.perf_counter()
        return r
    return orig(fn, *a)
sv._call = spy
for rep in range(8):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    harness.run(sim, K, fused=True)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"{name} K={K}: python before mpmhip_steps {1e6*(stamps['enter']-t0):6.1f} us | inside mpmhip_steps {1e6*(stamps['leave']-stamps['enter']):7.1f} us | "
          f"after {1e6*(t1-stamps['leave']):5.1f} us | final sync {1e6*(t2-t1):6.1f} us | total {1e6*(t2-t0):7.1f} us = {1e6*(t2-t0)/K:.1f} us/substep", flush=True)
