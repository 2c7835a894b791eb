#!/usr/bin/env python
"""Lane fragmentation of p2g's pre-reduction early and late in the headline run: lanes that issue LDS atomics per lane
that holds a particle (device counter, MPMHIP_DBG bit 512), and the fused-loop kernel times.
    python tools/gpu/late_frag.py [scene] [rebin_interval]      (rebin_interval < 0: exactly every -n substeps)"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from mpmavatar_amd import harness, scenes
scene = sys.argv[1] if len(sys.argv) > 1 else "sheet-500k"
ri = int(sys.argv[2]) if len(sys.argv) > 2 else 0
sim = harness.build_solver(scenes.REGISTRY[scene](), "cuda:0", rebin_interval=ri)
sv = sim.solver
done = 0
def counter(i):
    v = C.c_int64()
    sv._call("mpmhip_debug_counter", i, C.byref(v))
    return v.value
for upto in (60, 1000, 2200, 3000):
    harness.run(sim, upto - done, fused=True); done = upto
    a0, b0 = counter(8), counter(9)
    sv._call("mpmhip_set_debug_flags", 512)
    harness.run(sim, 60, fused=True); done += 60
    sv._call("mpmhip_set_debug_flags", 0)
    a1, b1 = counter(8), counter(9)
    r0 = sv.stats()["rebins"]
    sv.enable_profiling(True, fused=True); sv.time_profile.clear()
    harness.run(sim, 120, fused=True); done += 120
    sv.enable_profiling(False)
    tp = {k: round(1e3 * sum(v) / len(v), 1) for k, v in sv.time_profile.items() if k in ("compute_stress_from_F_trial", "p2g", "g2p_v")}
    rb = sv.time_profile.get("rebin", [])
    st = sv.stats()
    print(f"{scene} interval={ri} after {upto}: atomic lanes / particle lanes = {(a1-a0)/max(b1-b0,1):.3f}  {tp} "
          f"re-sorts in the 120-substep window {st['rebins']-r0} ({[round(x*1e3) for x in rb[-1:]]} us each) fallback {st['n_fallback_particles']}", flush=True)
