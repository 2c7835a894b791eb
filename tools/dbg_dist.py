import os, sys, faulthandler
faulthandler.enable()
sys.path.insert(0, '.')
import numpy as np, torch, torch.distributed as dist
from mpmavatar_amd import dist as mdist, scenes
dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
sc = scenes.small_garment()
P = lambda *a: print(f"[{rank}]", *a, flush=True)
ss = mdist.build_sharded(sc, "cuda:0", rank, world, rebin_interval=8)
torch.cuda.synchronize(); P("built", ss.shard.scene.n_particles)
mdist.rebin_all(ss); torch.cuda.synchronize(); P("rebin ok, peers", [(p['rank'], p['n_halo'], p['n_gs'], p['n_gr']) for p in ss.peers])
sv = ss.sim.solver; sim = ss.sim
dp = lambda t: None if t is None or t.numel() == 0 else t.data_ptr()
sv._call("mpmhip_dist_step_begin", float(sc.dt), dp(sim.mesh_x0), dp(sim.mesh_v), 0.0, None, 0, dp(sim.joint_verts_v) or sv._dummy_ptr(), dp(sim.joint_faces_v) or sv._dummy_ptr())
torch.cuda.synchronize(); P("begin ok")
mdist._exchange(ss, "halo"); torch.cuda.synchronize(); P("halo ok")
sv._call("mpmhip_dist_step_mid"); torch.cuda.synchronize(); P("mid ok")
mdist._exchange(ss, "ghost"); torch.cuda.synchronize(); P("ghost ok")
sv._call("mpmhip_dist_step_end"); torch.cuda.synchronize(); P("end ok")
dist.destroy_process_group()
