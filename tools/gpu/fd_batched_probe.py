"""The batched FD leg alone (for rocprofv3 --kernel-trace --stats): python tools/gpu/fd_batched_probe.py [batched|concurrent|sequential]"""
import sys, time
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from mpmavatar_amd import fd, scenes
mode = sys.argv[1] if len(sys.argv) > 1 else "batched"
if mode == "concurrent":
    fd.request_hw_queues()
sc = scenes.garment_cylinder(n_theta=200, n_h=200, n_grid=128, aniso=True)
frames = fd.synthetic_problem(sc, n_frames=2, frame_dt=400e-4)
m = fd.MaterialFD(sc, frames, frame_dt=400e-4, substeps=400, concurrent=mode == "concurrent", batched=mode == "batched")
fd.capture(m, 1.0, 1.0, 1.0)
m.losses(1.2, 1.0, 1.0)
torch.cuda.synchronize(); t0 = time.perf_counter(); n0 = m.substeps_done
m.losses(1.1, 1.0, 1.0)
torch.cuda.synchronize(); dt = time.perf_counter() - t0
print(mode, (m.substeps_done - n0) / dt, "substeps/s")
