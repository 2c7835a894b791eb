#!/usr/bin/env python
"""Throughput under the REFERENCE'S call pattern -- one Python-level MPMWARP.p2g2p(...) per substep with the mesh advected
on the torch side (train_material_params.py:621-626) -- against the fused mpmhip_steps loop.
    python tools/gpu/call_pattern.py [scene ...]"""
import gc, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from mpmavatar_amd import harness, scenes

REG = dict(scenes.REGISTRY)
# the S3 garment on a body in uniform motion: the harness poses a swaying body with NumPy on the host every substep, which
# would measure the harness, not the call
REG["garment-120k-aniso-uniform"] = lambda: scenes.garment_cylinder(aniso=True)
for name in (sys.argv[1:] or ["sheet-500k", "garment-120k-aniso-uniform", "demo-250", "cube-8k"]):
    out = {}
    for fused in (False, True):
        sim = harness.build_solver(REG[name](), "cuda:0")
        harness.run(sim, 60, fused=fused)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        harness.run(sim, 300, fused=fused)
        torch.cuda.synchronize()
        out[fused] = 300 / (time.perf_counter() - t0)
        del sim
        gc.collect()                 # the context's hipFree calls would otherwise land inside a later timing
        torch.cuda.synchronize()
    print(f"{name}: one p2g2p call per substep {out[False]:.0f} substeps/s, fused loop {out[True]:.0f} substeps/s", flush=True)
