#!/usr/bin/env python
"""A/B of run-time kernel switches (MPMHIP_DBG bits that keep the results right) on the fused loop, early and late.
    python tools/gpu/ab_flags.py <scene> <flagsA> <flagsB> [substeps before the late measurement]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import time, torch
from mpmavatar_amd import harness, scenes
scene, fa, fb = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
late = int(sys.argv[4]) if len(sys.argv) > 4 else 2200
for upto in (100, late):
    for flags in (fa, fb, fa, fb):
        sim = harness.build_solver(scenes.REGISTRY[scene](), "cuda:0")
        sv = sim.solver
        harness.run(sim, upto, fused=True)
        sv._call("mpmhip_set_debug_flags", flags)
        harness.run(sim, 20, fused=True)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        harness.run(sim, 200, fused=True)
        torch.cuda.synchronize(); el = (time.perf_counter() - t0) / 200
        sv.enable_profiling(True, fused=True); sv.time_profile.clear()
        harness.run(sim, 100, fused=True)
        sv.enable_profiling(False)
        tp = {k: round(1e3 * sum(v) / len(v), 1) for k, v in sv.time_profile.items() if k in ("compute_stress_from_F_trial", "p2g", "g2p_v")}
        print(f"{scene} after {upto} flags {flags}: {el*1e6:.1f} us/substep  {tp}", flush=True)
        del sim
