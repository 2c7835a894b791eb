"""Finite-difference material fitting over the solver.

Restates the training step of the reference's physics driver, ``Trainer.train_one_step``
(/root/reference/train_material_params.py:575-671): the garment is simulated four times per step -- at the current
parameters (density D, Young's modulus 100*E, rest-pose height scale H) and with D, E, H nudged by 0.05 / 0.05 / 0.005
-- the loss of each run is the mean over frames of the MSE between simulated and captured cloth vertices in world space,
the three difference quotients are written into ``.grad`` and Adam + cosine annealing + clamping update the parameters
(:657-670).  Checkpoints are the reference's ``best_param_*.npz`` / ``last_param_*.npz`` (:725-728, io_formats).

What is MI355X-specific: the four runs are independent, and a garment-sized substep (~120k particles) leaves most of the
GPU idle (two or three launches of ~10 us each, far below one round of workgroups).  With ``concurrent=True`` every
variant owns a solver context on its own HIP stream and a host thread (ctypes releases the GIL inside the library), so
the four simulations share the GPU instead of queueing behind each other; ``concurrent=False`` is the reference's
sequential order on one context.  Both give the same losses.  Across GPUs the variants are the natural shard: no
exchange at all (``variant_slice``).
"""
from __future__ import annotations

import dataclasses
import math
from concurrent.futures import ThreadPoolExecutor
from typing import List, Optional, Sequence

import numpy as np
import torch

from . import garment, harness
from .scenes import Scene

# (dD, dE, dH) of the four simulations, train_material_params.py:583
DELTAS = ((0.0, 0.0, 0.0), (0.05, 0.0, 0.0), (0.0, 0.05, 0.0), (0.0, 0.0, 0.005))


@dataclasses.dataclass
class Frame:
    """One captured frame in simulation space: body mesh at the frame start and its velocity (advected inside the frame
    as mesh_x + k*dt*mesh_v, :622), velocities of the attached garment vertices / faces, and the captured cloth
    vertices of the NEXT frame in world space (the loss target, :631)."""
    mesh_x: np.ndarray
    mesh_v: np.ndarray
    joint_verts_v: Optional[np.ndarray]
    joint_faces_v: Optional[np.ndarray]
    target: np.ndarray


def variant_slice(rank: int, world: int, n: int = len(DELTAS)) -> range:
    """Variants simulated by ``rank`` when the FD runs are spread over ``world`` processes (contiguous, balanced)."""
    lo, hi = (rank * n) // world, ((rank + 1) * n) // world
    return range(lo, hi)


_HWQ_SET_BY_US = None   # None: request_hw_queues() has not set the variable; else: whether its setting came in time


def hip_runtime_started() -> bool:
    """Has anything in this process initialised the HIP / HSA runtime yet?  The runtime opens /dev/kfd when it initialises -- at the first
    HIP call of any kind, torch.cuda.is_available() and device_count() included, which torch.cuda.is_initialized() does not report
     -- so an open descriptor on /dev/kfd is the test."""
    import os
    try:
        for fd in os.listdir("/proc/self/fd"):
            try:
                if os.readlink(f"/proc/self/fd/{fd}") == "/dev/kfd":
                    return True
            except OSError:
                continue
    except OSError:
        return True      # cannot tell (no /proc): do not claim the setting will work
    return False


def request_hw_queues(n: int = 8) -> bool:
    """One hardware queue per HIP stream for the concurrent FD step.  The ROCm runtime multiplexes all HIP streams of a process onto
    GPU_MAX_HW_QUEUES = 4 hardware queues; four solver contexts on four streams beside torch's own then share queues and take turns
    (36.0 k substeps/s at the S3 size with the default, 45.3 k with 8, nothing more with 16: profiles/HISTORY.md).  The runtime reads the
    variable when it initialises, i.e. at the FIRST HIP call of the process -- torch.cuda.is_available() is one -- so this must run (or
    the variable be exported) before anything touched the device.  A value the user exported wins.  Returns True when the setting
    takes effect; when the runtime is already up (hip_runtime_started) it sets nothing, WARNS and returns False: the FD step still
    runs, on shared queues, and the caller has been told -- no silent -20 %."""
    import os
    import warnings
    global _HWQ_SET_BY_US
    if _HWQ_SET_BY_US is not None:
        return _HWQ_SET_BY_US            # an earlier call of this function decided
    if "GPU_MAX_HW_QUEUES" in os.environ:
        return True                      # exported by the user (before the process started: in time by construction)
    if hip_runtime_started():
        warnings.warn("mpmavatar_amd.fd: the HIP runtime of this process is already initialised (/dev/kfd is open: some torch.cuda / HIP "
                      "call ran, torch.cuda.is_available() counts), GPU_MAX_HW_QUEUES cannot be raised any more; the concurrent "
                      "finite-difference contexts will share hardware queues (about -20 % throughput).  Call "
                      "mpmavatar_amd.fd.request_hw_queues() or export GPU_MAX_HW_QUEUES=8 before the first CUDA/HIP call.", stacklevel=2)
        _HWQ_SET_BY_US = False
        return False
    os.environ["GPU_MAX_HW_QUEUES"] = str(int(n))
    _HWQ_SET_BY_US = True
    return True


class MaterialFD:
    """train_material_params.py:575-728 on top of the shim.  ``scene`` supplies the garment (vertices, faces, collider
    mesh, joints, fixed nu / gamma / kappa); ``frames`` the driving motion and targets."""

    def __init__(self, scene: Scene, frames: Sequence[Frame], *, init=(1.0, 1.0, 1.0), ranges=((0.1, 10.0), (0.1, 10.0), (0.5, 1.5)),
                 lrs=(0.05, 0.05, 0.005), iterations=100, frame_dt=1.0 / 25, substeps=400, scale=1.0, shift=(0.0, 0.0, 0.0),
                 device="cuda:0", concurrent=True, mode=None, variants: Optional[Sequence[int]] = None, build=True):
        self.sc, self.frames = scene, list(frames)
        self.device = torch.device(device)
        self.iterations = int(iterations)
        self.substeps = int(substeps)
        self.substep_size = frame_dt / substeps                       # :579-581
        self.scale = float(scale)
        self.shift = torch.tensor(np.asarray(shift, np.float32).reshape(1, 3), device=self.device if build else "cpu")
        self.param_ranges = {"D": list(ranges[0]), "E": list(ranges[1]), "H": list(ranges[2])}
        self.torch_param = {k: torch.tensor(float(v), dtype=torch.float32, requires_grad=True) for k, v in zip("DEH", init)}
        # :190-191
        self.optimizer = torch.optim.Adam([{"params": [self.torch_param[k]], "lr": lr} for k, lr in zip("DEH", lrs)], lr=lrs[0])
        self.scheduler = torch.optim.lr_scheduler.CosineAnnealingLR(self.optimizer, self.iterations * 0.5 * np.pi / np.arccos(0.4), eta_min=0.0)
        self.best_params = {"D": float(init[0]), "E": float(init[1]) * 100, "H": float(init[2]), "loss": 1.0, "step": -1}
        self.last_params = dict(self.best_params)
        self.step = 0
        self.variants = list(range(len(DELTAS))) if variants is None else list(variants)
        self.substeps_done = 0
        if not build:   # optimiser / bookkeeping only (host-side tests)
            self.concurrent, self.sims, self.pool = False, [], None
            return
        # concurrent: one context, HIP stream and host thread per variant (one batched launch per phase over the four contexts was built
        # in round 5, bit-identical and slower -- 31.9 k against 46.2 k substeps/s -- and removed again: profiles/r05_experiments.md 7)
        self.concurrent = bool(concurrent) and len(self.variants) > 1
        if self.concurrent:
            request_hw_queues()          # (opt-in by use: only the concurrent contexts want it; no import side effect)
        n_ctx = len(self.variants) if self.concurrent else 1
        self.streams = [torch.cuda.Stream(self.device) for _ in range(n_ctx)] if self.concurrent else [None] * n_ctx
        self.sims = []
        for s in self.streams:
            with torch.cuda.stream(s) if s is not None else _null():
                self.sims.append(harness.build_solver(scene, str(self.device), mode=mode))
        self.pool = ThreadPoolExecutor(max_workers=n_ctx) if self.concurrent else None
        t = lambda a: None if a is None else torch.as_tensor(np.ascontiguousarray(a), dtype=torch.float32, device=self.device)
        self._frames = [dict(mesh_x=t(f.mesh_x), mesh_v=t(f.mesh_v), jv=t(f.joint_verts_v), jf=t(f.joint_faces_v), target=t(f.target))
                        for f in self.frames]
        self._x0, self._d0, self._v0 = t(scene.x), t(scene.d), t(scene.v)
        self._verts0 = scene.x[scene.n_elements + scene.n_traditional:].astype(np.float32)

    # ---- one simulation ------------------------------------------------------------------------------------------
    def sim2wld(self, p):
        return (p - self.shift) / self.scale   # :372-373

    def simulate(self, sim, D: float, E: float, H: float, record: Optional[List[np.ndarray]] = None) -> float:
        """Reset the state to the first frame, set density / stiffness / rest pose, run every frame's substeps as one
        fused call and accumulate the vertex loss (:584-641).  E is the trained parameter: Young's modulus is 100*E."""
        sc, dev = self.sc, self.device
        st, md, sv = sim.state, sim.model, sim.solver
        n_p = sc.n_particles
        scaled = self._verts0 * np.array([[1.0, H, 1.0]], np.float32)                       # :587
        R_inv = torch.as_tensor(garment.compute_rest_dir_inv_from_vf(scaled, sc.faces), device=dev)
        st.reset_state(sc.n_vertices, self._x0.clone(), self._d0.clone(), None, self._v0.clone(), tensor_R_inv=R_inv, device=dev,
                       requires_grad=True)
        st.set_require_grad(True)
        ones = torch.ones(n_p, dtype=torch.float32, device=dev)
        st.reset_density(ones * D, None, dev, update_mass=True)                               # :603-605
        sv.set_E_nu_from_torch(md, ones * (E * 100.0), ones * sc.nu, ones * sc.gamma, ones * sc.kappa, dev)  # :607-609
        sv.prepare_mu_lam(md, st, dev)
        loss = torch.zeros((), dtype=torch.float32, device=dev)
        for f in self._frames:
            sv.p2g2p_n(md, st, self.substep_size, self.substeps, mesh_x=f["mesh_x"], mesh_v=f["mesh_v"], joint_traditional_v=None,
                       joint_verts_v=f["jv"], joint_faces_v=f["jf"])                          # :621-626 as one call
            cloth = self.sim2wld(st.particle_x[sc.n_elements:])
            loss = loss + torch.nn.functional.mse_loss(cloth, f["target"])                    # :630-631
            if record is not None:
                record.append(cloth.detach().cpu().numpy().copy())
        return float((loss / len(self._frames)).item())

    def losses(self, D: float, E: float, H: float) -> List[float]:
        """Losses of this process's variants at (D, E, H) + DELTAS[i]."""
        jobs = [(D + DELTAS[i][0], E + DELTAS[i][1], H + DELTAS[i][2]) for i in self.variants]
        if not self.concurrent:
            out = [self.simulate(self.sims[0], *p) for p in jobs]
        else:
            def work(k):
                with torch.cuda.stream(self.streams[k]):
                    return self.simulate(self.sims[k], *jobs[k])
            out = list(self.pool.map(work, range(len(jobs))))
        self.substeps_done += len(jobs) * len(self._frames) * self.substeps
        return out

    # ---- the training step ---------------------------------------------------------------------------------------
    def apply_losses(self, geo_losses: Sequence[float]) -> dict:
        """:650-712: difference quotients -> .grad -> Adam, cosine schedule, clamp, best/last bookkeeping."""
        g = {"D": (geo_losses[1] - geo_losses[0]) / 0.05, "E": (geo_losses[2] - geo_losses[0]) / 0.05,
             "H": (geo_losses[3] - geo_losses[0]) / 0.005}
        self.optimizer.zero_grad()
        for k in "DEH":
            self.torch_param[k].grad = torch.tensor(g[k]).float()
        self.optimizer.step()
        self.scheduler.step()
        with torch.no_grad():
            for k in "DEH":
                self.torch_param[k].clamp_(min=float(self.param_ranges[k][0]), max=float(self.param_ranges[k][-1]))
        self.last_params = {"D": self.torch_param["D"].item(), "E": self.torch_param["E"].item() * 100, "H": self.torch_param["H"].item(),
                            "loss": geo_losses[0], "step": self.step}
        if geo_losses[0] < self.best_params["loss"]:
            self.best_params = dict(self.last_params)
        self.step += 1
        return {"loss": geo_losses[0], "grad": g, **{k: self.last_params[k] for k in "DEH"}}

    def train_one_step(self) -> dict:
        if len(self.variants) != len(DELTAS):
            raise RuntimeError("train_one_step needs all four variants here; with variant_slice() use train_one_step_sharded()")
        p = self.torch_param
        return self.apply_losses(self.losses(p["D"].item(), p["E"].item(), p["H"].item()))

    def train_one_step_sharded(self, group=None) -> dict:
        """The same step with the four simulations spread over the ranks of a ``torch.distributed`` group (construct
        every rank's MaterialFD with ``variants=variant_slice(rank, world)``): the runs are independent, the only
        exchange is an all-gather of the four losses, after which every rank applies the identical Adam update -- so the
        parameters stay bit-identical on all ranks without a broadcast.  1, 2 or 4 ranks."""
        import torch.distributed as dist
        world, rank = dist.get_world_size(group), dist.get_rank(group)
        if list(self.variants) != list(variant_slice(rank, world)):
            raise RuntimeError(f"rank {rank} of {world} must hold variants {list(variant_slice(rank, world))}, not {self.variants}")
        p = self.torch_param
        mine = self.losses(p["D"].item(), p["E"].item(), p["H"].item())
        parts = [None] * world
        dist.all_gather_object(parts, [float(x) for x in mine], group=group)
        return self.apply_losses([x for part in parts for x in part])

    def close(self):
        if self.pool is not None:
            self.pool.shutdown()


class _null:
    def __enter__(self): return None
    def __exit__(self, *a): return False


# ---- synthetic captured sequence ------------------------------------------------------------------------------------
def synthetic_problem(scene: Scene, n_frames=4, frame_dt=1.0 / 25, sway=0.4, period=0.5):
    """Driving motion for ``scene`` (a garment_cylinder-like scene with a body mesh and attached rows): the body sways
    along x with velocity sway*sin(2 pi t / period), constant inside a frame like the captured SMPL-X velocities
    (:617-620).  Targets are filled in by ``capture``."""
    frames, x = [], scene.mesh_vertices.astype(np.float32).copy()
    njv, njf = scene.num_joint_v, scene.num_joint_f
    for i in range(n_frames):
        vel = np.float32(sway * math.sin(2 * math.pi * (i + 0.5) * frame_dt / period))
        mesh_v = np.tile(np.array([[vel, 0.0, 0.0]], np.float32), (x.shape[0], 1))
        jv = np.tile(np.array([[vel, 0.0, 0.0]], np.float32), (njv, 1)) if njv else None
        jf = np.tile(np.array([[vel, 0.0, 0.0]], np.float32), (njf, 1)) if njv else None
        frames.append(Frame(x.copy(), mesh_v, jv, jf, np.zeros((scene.n_vertices + scene.n_traditional, 3), np.float32)))
        x = x + np.float32(frame_dt) * mesh_v
    return frames


def capture(fd: MaterialFD, D: float, E: float, H: float):
    """Simulate once at the 'true' parameters and store the cloth vertices as the frames' targets."""
    rec: List[np.ndarray] = []
    stream = fd.streams[0]
    with torch.cuda.stream(stream) if stream is not None else _null():
        fd.simulate(fd.sims[0], D, E, H, record=rec)
    for f, fr, r in zip(fd._frames, fd.frames, rec):
        fr.target = r
        f["target"] = torch.as_tensor(r, device=fd.device)
    torch.cuda.synchronize(fd.device)
