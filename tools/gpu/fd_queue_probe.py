#!/usr/bin/env python
"""Does a stream created with hipExtStreamCreateWithCUMask (all CUs) get a hardware queue of its own whatever GPU_MAX_HW_QUEUES says?
Concurrent FD step (four contexts) with (a) torch streams after a LATE request (runtime already up: shared queues), (b) CU-masked
streams in the same late situation, (c) torch streams with the variable exported in time.    python tools/gpu/fd_queue_probe.py a|b|c"""
import ctypes as C, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
mode = sys.argv[1]
if mode == "c":
    os.environ["GPU_MAX_HW_QUEUES"] = "8"
import torch
torch.cuda.is_available(); torch.zeros(1, device="cuda")     # the runtime is up (and has read the variable) from here on
from mpmavatar_amd import fd, scenes
if mode == "b":
    hip = C.CDLL("libamdhip64.so")
    def masked(device=None):
        s = C.c_void_p()
        arr = (C.c_uint32 * 8)(*([0xffffffff] * 8))
        assert hip.hipExtStreamCreateWithCUMask(C.byref(s), 8, arr) == 0
        return torch.cuda.ExternalStream(s.value)
    torch.cuda.Stream = masked          # (probe only: MaterialFD builds its streams with torch.cuda.Stream(device))
sc = scenes.garment_cylinder(n_theta=200, n_h=200, n_grid=128, aniso=True)
frames = fd.synthetic_problem(sc, n_frames=2, frame_dt=400 * 1e-4)
import warnings
with warnings.catch_warnings():
    warnings.simplefilter("ignore")
    m = fd.MaterialFD(sc, frames, frame_dt=400 * 1e-4, substeps=400, concurrent=True)
fd.capture(m, 1.0, 1.0, 1.0)
m.losses(1.2, 1.0, 1.0)
torch.cuda.synchronize(); t0 = time.perf_counter(); n0 = m.substeps_done
for _ in range(2):
    m.train_one_step()
torch.cuda.synchronize(); dt = time.perf_counter() - t0
print(json.dumps({"mode": mode, "GPU_MAX_HW_QUEUES": os.environ.get("GPU_MAX_HW_QUEUES"), "substeps_per_s": (m.substeps_done - n0) / dt}), flush=True)
