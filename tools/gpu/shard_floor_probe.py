"""Root-cause probe for the 416 us/substep `shard_floor` of GPUTEST_r04: the shard scene of rank 0 of a two-slab
cube-8k, run alone as a single-GPU scene exactly the way bench.py (round 4) did -- build, 20 warm-up substeps, then one timed call
of 100 -- but with the timed region cut into calls of 10 substeps and the solver's statistics printed beside each, three builds
in a row, and then the same with a second process idling on the GPU."""
import json
import sys
import time

import torch

sys.path.insert(0, ".")
from mpmavatar_amd import dist as mdist, harness, scenes  # noqa: E402


def probe(tag, sc):
    t0 = time.perf_counter()
    sim = harness.build_solver(sc, "cuda:0", mode="fast")
    harness.run(sim, 20, fused=True)
    torch.cuda.synchronize()
    t_build = time.perf_counter() - t0
    rows = []
    for k in range(20):
        st0 = sim.solver.stats()
        t = time.perf_counter()
        harness.run(sim, 10, fused=True)
        torch.cuda.synchronize()
        us = 1e5 * (time.perf_counter() - t)
        st = sim.solver.stats()
        rows.append((round(us, 1), st["rebins"] - st0["rebins"], st["n_fallback_particles"]))
    # one call of 100 like bench.py r4
    t = time.perf_counter()
    harness.run(sim, 100, fused=True)
    torch.cuda.synchronize()
    one = 1e4 * (time.perf_counter() - t)
    print(json.dumps({"tag": tag, "n_particles": int(sc.n_particles), "build_plus_warmup_s": round(t_build, 3),
                      "us_per_substep_by_10": rows, "us_per_substep_one_call_of_100": round(one, 1),
                      "g2p2g_launches": sim.solver.stats().get("g2p2g_launches")}), flush=True)


if __name__ == "__main__":
    full = scenes.REGISTRY["cube-8k"]()
    shard = mdist.partition(full, 2)[0].scene
    for i in range(3):
        probe(f"full-{i}", full)
    if shard is not None:
        for i in range(3):
            probe(f"shard0of2-{i}", shard)
