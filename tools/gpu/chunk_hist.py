import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
os.environ["MPMHIP_VERBOSE"] = "1"
from mpmavatar_amd import harness, scenes
for name, n in (("sheet-500k", 2400), ("garment-120k-aniso", 400), ("demo-250", 2000)):
    sim = harness.build_solver(scenes.REGISTRY[name](), "cuda:0")
    harness.run(sim, n, fused=True)
    sim.solver.close()
