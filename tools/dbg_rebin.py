import sys; sys.path.insert(0,'.')
import torch
from mpmavatar_amd import harness, scenes
sc = scenes.sheet()
mode = sys.argv[1] if len(sys.argv) > 1 else "fast"
sim = harness.build_solver(sc, "cuda:0", mode=mode)
for i in range(16):
    harness.run(sim, 64, fused=True)
    torch.cuda.synchronize()
    st = sim.solver.stats()
    x = sim.state.particle_x
    v = sim.state.particle_v
    print(i, (i+1)*64, {k: st[k] for k in ("rebins","n_fallback_particles","n_active_blocks","n_active_nodes")},
          "y mean %.5f min %.5f  vy mean %.4f  finite %s" % (x[:,1].mean().item(), x[:,1].min().item(), v[:,1].mean().item(), bool(torch.isfinite(x).all())), flush=True)
