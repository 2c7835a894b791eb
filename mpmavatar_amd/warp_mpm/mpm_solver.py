"""``MPMWARP`` -- the reference's solver class (/root/reference/warp_mpm/mpm_solver.py:13-1417) as a thin
Python shim over libmpmhip.so.  Same constructor, setters, collider / boundary-condition registration and
``p2g2p`` substep; every device operation goes through the C ABI in include/mpmhip.h.

Differences that are visible to a caller (all documented in INTEGRATION.md):
  * state/model fields are torch tensors (no ``wp.array``); ``mesh.id`` is an opaque int;
  * mesh / joint tensors handed to ``p2g2p`` stay on the device (the reference bounces them through
    ``.cpu().numpy()`` every substep, mpm_solver.py:282-302,422-423);
  * ``ScopedTimer`` syncs are off unless ``enable_profiling(True)``; ``time_profile`` keeps the same keys;
  * ``p2g2p_n`` fuses the caller's substep loop (train_material_params.py:622-626) into one C call;
  * there is no CPU execution path: without a visible MI355X the constructor raises ``MPMHipError``.
"""
from __future__ import annotations

import ctypes as C
import math
import os
import sys
from typing import Optional

import numpy as np
import torch

from .. import _lib as L
from .mpm_data_structure import MPMModelStruct, MPMStateStruct

_MATERIALS = {"jelly": 0, "metal": 1, "sand": 2, "foam": 3, "snow": 4, "plasticine": 5, "neo-hookean": 6, "cloth": 7}


class _BodyMesh:
    """Stand-in for ``wp.Mesh`` (mpm_solver.py:45-51): callers only read ``.id`` (run_demo.py:378)."""

    def __init__(self, handle, n_v, n_f):
        self.id, self.num_v, self.num_f = handle, n_v, n_f


def _f3(v):
    return L.f3(float(v[0]), float(v[1]), float(v[2]))


def _mode_from_env():
    m = os.environ.get("MPMHIP_MODE", "fast").lower()
    return L.MODE_BASELINE if m in ("baseline", "1") else L.MODE_FAST


class MPMWARP(object):
    # mpm_solver.py:14-51
    def __init__(self, n_particles, n_elements, n_vertices, n_grid=100, grid_lim=1.0, mesh_vertices=None,
                 mesh_faces=None, num_joint_t=0, num_joint_v=0, num_joint_f=0, device="cuda:0", mode=None,
                 rebin_interval=0, p2g_tile="auto"):
        # p2g_tile (no reference counterpart): "auto" | "fixed" | "f64", mpmhip_config.p2g_tile -- the one numerics-changing setting
        self._lib = L.load()
        self._p2g_tile = {"auto": 0, "fixed": 1, "fx": 1, "f64": 2}[p2g_tile] if isinstance(p2g_tile, str) else int(p2g_tile)
        self._ctx = None
        self._mode = _mode_from_env() if mode is None else {"fast": L.MODE_FAST, "baseline": L.MODE_BASELINE}.get(mode, mode)
        self._rebin_interval = rebin_interval
        self.initialize(n_particles, n_elements, n_vertices, n_grid, grid_lim, mesh_vertices, mesh_faces, num_joint_t,
                        num_joint_v, num_joint_f, device=device)
        self.time_profile = {}
        self.kernel_profile = {}   # bench.py: kernel durations of the fused loop's launches (no reference counterpart)

    def initialize(self, n_particles, n_elements, n_vertices, n_grid=100, grid_lim=1.0, mesh_vertices=None,
                   mesh_faces=None, num_joint_t=0, num_joint_v=0, num_joint_f=0, device="cuda:0"):
        self.n_particles, self.n_elements, self.n_vertices = n_particles, n_elements, n_vertices
        self.n_no_vertices = n_particles - n_vertices
        self.num_joint_t, self.num_joint_v, self.num_joint_f = num_joint_t, num_joint_v, num_joint_f
        self.n_grid, self.grid_lim = n_grid, grid_lim
        self.device = torch.device(device)
        if self._ctx is not None:
            self._lib.mpmhip_destroy(self._ctx)
            self._ctx = None
        if self.device.type != "cuda":
            raise L.MPMHipError(L.ERR_NO_DEVICE, f"MPMWARP needs a ROCm device, got device={device!r}; libmpmhip has "
                                                 "no CPU path")
        dev_index = self.device.index if self.device.index is not None else 0
        stream = 0
        if torch.cuda.is_available():
            stream = torch.cuda.current_stream(self.device).cuda_stream
        cfg = L.Config(n_particles, n_elements, n_vertices, n_grid, float(grid_lim), num_joint_t, num_joint_v,
                       num_joint_f, dev_index, self._mode, self._rebin_interval, 0, stream, self._p2g_tile, 0)
        ctx = L.vp()
        rc = self._lib.mpmhip_create(C.byref(cfg), C.byref(ctx))
        L.check(self._lib, None, rc)
        self._ctx = ctx
        self._bound_state = None
        self._bound_state_version = -1
        self._bound_model = None
        self._bound_model_version = -1
        self._bound_scalar_version = -1
        self._keep = None
        self._masks = []
        self._profiling = False
        self._read_snaps = {}
        self._host_dt = None
        # bookkeeping lists kept for API familiarity (mpm_solver.py:30-43)
        self.grid_postprocess, self.collider_params, self.modify_bc = [], [], []
        self.mesh_colliders, self.mesh_collider_params = [], []
        self.particle_movers, self.particle_mover_params = [], []
        self.pre_p2g_operations, self.impulse_params = [], []
        self.particle_velocity_modifiers, self.particle_velocity_modifier_params = [], []
        if mesh_vertices is not None and mesh_faces is not None:
            mv = np.ascontiguousarray(np.asarray(mesh_vertices, np.float32).reshape(-1, 3))
            mf = np.ascontiguousarray(np.asarray(mesh_faces, np.int32).reshape(-1))
            self._call("mpmhip_set_body_mesh", mv.shape[0], mf.size // 3, mv.ctypes.data, mf.ctypes.data)
            self.num_mesh_v, self.num_mesh_f = mv.shape[0], mf.size // 3
            self.mesh = _BodyMesh(id(self) & 0x7FFFFFFF, self.num_mesh_v, self.num_mesh_f)

    def close(self):
        """Destroy the solver context now (device memory, streams, the multi-GPU communicator).  Idempotent; the object is
        unusable afterwards.  (Not in the reference, whose Warp arrays die with the Python object.)"""
        if getattr(self, "_ctx", None):
            self._lib.mpmhip_destroy(self._ctx)
            self._ctx = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ------------------------------------------------------------------ plumbing
    def _call(self, name, *args):
        rc = getattr(self._lib, name)(self._ctx, *args)
        if rc != L.OK:
            msg = self._lib.mpmhip_last_error(self._ctx).decode()
            if rc == L.ERR_INVALID and "material" in msg:
                raise TypeError(msg)
            if rc == L.ERR_INVALID and "sticky" in msg:
                raise ValueError(msg)
            raise L.MPMHipError(rc, msg)

    @property
    def time(self):
        return self._lib.mpmhip_get_time(self._ctx)

    @time.setter
    def time(self, t):
        self._lib.mpmhip_set_time(self._ctx, float(t))

    def _ptr(self, t: Optional[torch.Tensor], n_rows=None, name="tensor"):
        if t is None:
            return None
        if not isinstance(t, torch.Tensor):
            t = torch.as_tensor(np.asarray(t), dtype=torch.float32)
        t = t.detach()
        if t.device != self.device or t.dtype != torch.float32 or not t.is_contiguous():
            t = t.to(device=self.device, dtype=torch.float32).contiguous()
        if n_rows is not None and (t.dim() != 2 or t.shape[0] != n_rows or t.shape[1] != 3):
            raise RuntimeError(f"{name}: expected shape ({n_rows}, 3), got {tuple(t.shape)}")
        return t

    def _bind(self, mpm_model: MPMModelStruct, mpm_state: MPMStateStruct):
        st = mpm_state
        if st is not self._bound_state or st._version != self._bound_state_version:
            if st.n_particles != self.n_particles or st.n_elements != self.n_elements or st.n_vertices != self.n_vertices:
                raise RuntimeError("mpm_state sizes do not match the solver's")
            p = L.StatePtrs()
            for name, _ in L.StatePtrs._fields_:
                t = st._raw(name)
                want = torch.int32 if name == "particle_selection" else torch.float32
                if t.device != self.device or t.dtype != want or not t.is_contiguous():
                    raise RuntimeError(f"mpm_state.{name}: expected a contiguous {want} tensor on {self.device}")
                setattr(p, name, t.data_ptr() if t.numel() else None)
            self._call("mpmhip_bind_state", C.byref(p))
            st._attach(self)
            self._bound_state, self._bound_state_version = st, st._version
            self._watch(st)   # tensors the caller took before the first substep are handed-out tensors too
        md = mpm_model
        if md is not self._bound_model or md._version != self._bound_model_version:
            p = L.ModelPtrs()
            for name, _ in L.ModelPtrs._fields_:
                t = md._raw(name)
                if t.device != self.device or t.dtype != torch.float32 or not t.is_contiguous() or t.numel() != self.n_particles:
                    raise RuntimeError(f"mpm_model.{name}: expected a contiguous float32 [n_particles] tensor on {self.device}")
                setattr(p, name, t.data_ptr() if t.numel() else None)
            self._call("mpmhip_bind_model", C.byref(p))
            md._attach(self)
            self._bound_model, self._bound_model_version = md, md._version
            self._watch(md)
        if md._scalar_version != self._bound_scalar_version:
            g = md.gravitational_accelaration
            s = L.ModelScalars(int(md.material), float(md.friction_coeff), float(md.alpha), _f3(g), float(md.hardening),
                               float(md.xi), float(md.plastic_viscosity), float(md.softening), float(md.rpic_damping),
                               float(md.grid_v_damping_scale))
            self._call("mpmhip_set_model_scalars", C.byref(s))
            self._bound_scalar_version = md._scalar_version

    # hooks used by MPMStateStruct / MPMModelStruct
    #
    # The reference hands out zero-copy views of the solver's arrays (``wp.to_torch(mpm_state.particle_x)``): live in both
    # directions.  The fast back end keeps its own sorted copy, so the shim emulates that:
    #   * reading a solver-written field pulls the results into the tensor first;
    #   * every tensor ever handed out stays on a watch list with torch's in-place-modification counter
    #     (``Tensor._version``): before each substep the counters are compared and, if one moved, the state is re-imported
    #     (writes that bypass torch -- raw pointers, another library -- need ``state._touch()``);
    #   * a handed-out tensor the caller still holds on to (``x = state.particle_x`` kept across substeps) is refreshed
    #     after every substep, so that it does not go stale.  A read-and-drop (``.clone()``, ``.cpu()``: what the reference
    #     drivers do once per frame) costs nothing per substep.
    def _watch(self, obj):
        snap = self._read_snaps.setdefault(id(obj), {})
        for name in type(obj)._fields:
            t = obj._t.get(name)
            if isinstance(t, torch.Tensor):
                snap[name] = (t, t._version)
            else:
                snap.pop(name, None)

    def _before_caller_read(self, obj):
        if self._ctx and (obj is self._bound_state or obj is self._bound_model):
            self._call("mpmhip_pull_state")
            self._stale = False
            self._watch(obj)

    def _push_if_modified(self):
        if not self._read_snaps:
            return
        dirty, dirty_synced = False, False
        for obj in (self._bound_state, self._bound_model):
            snap = self._read_snaps.get(id(obj))
            if not snap:
                continue
            for name, (t, v) in list(snap.items()):
                if obj._t.get(name) is not t:      # field was rebound to another tensor: _bind() sees that
                    del snap[name]
                elif t._version != v:
                    dirty = True
                    dirty_synced = dirty_synced or name in type(obj)._synced
        if dirty:
            if getattr(self, "_stale", False):
                if dirty_synced:
                    # The caller wrote in place into a tensor that has NOT seen the substeps run since it was handed out -- a view
                    # (state.particle_x[:n], .view(-1)) kept across substeps: only whole-tensor handles are refreshed after a
                    # substep (_refresh_held counts references to the tensor OBJECT, a view holds the storage).  Importing it would
                    # first write the solver's newer results over that write and lose it silently; the reference's zero-copy
                    # aliases have no such case.  Re-read the field (state.particle_x) before modifying it.
                    raise RuntimeError("in-place write into a stale view of a solver-written field (taken before the last substeps): "
                                       "read the field again from the state (state.particle_x, ...) and modify that tensor")
                # Only fields the solver never writes were edited (particle_selection, particle_vol, E, gamma, ...): nothing can be
                # lost.  Bring the solver-written fields of the caller's tensors up to date first -- the pull does not touch the
                # others -- so that the import below does not put old positions back.
                self._call("mpmhip_pull_state")
                self._stale = False
            self._call("mpmhip_push_state")
            for obj in (self._bound_state, self._bound_model):
                if id(obj) in self._read_snaps:
                    self._watch(obj)

    def _refresh_held(self):
        """After a substep: if the caller still holds a tensor it was handed, write the new results into it now."""
        for obj in (self._bound_state, self._bound_model):
            snap = self._read_snaps.get(id(obj)) if obj is not None else None
            if not snap:
                continue
            for name in type(obj)._synced:
                ent = snap.get(name)
                # references we know of: obj._t, the watch-list tuple, `ent`'s own lookup is the same tuple, getrefcount's
                # argument.  Anything beyond that is the caller's.
                if ent is not None and sys.getrefcount(ent[0]) > 3:
                    self._call("mpmhip_pull_state")
                    self._stale = False
                    return

    def _before_caller_write(self, obj):
        if self._ctx and (obj is self._bound_state or obj is self._bound_model):
            self._call("mpmhip_push_state")   # (writes the solver's newer results back first: the tensors are current now)
            self._stale = False

    # ------------------------------------------------------------------ parameters
    def set_parameters(self, device="cuda:0", **kwargs):  # mpm_solver.py:54-55 (wrong arity in the reference)
        raise TypeError("set_parameters() is broken in the reference (mpm_solver.py:54-55); use set_parameters_dict")

    # mpm_solver.py:57-126
    def set_parameters_dict(self, mpm_model, mpm_state, kwargs={}, device="cuda:0"):
        if "material" in kwargs:
            if kwargs["material"] not in _MATERIALS:
                raise TypeError("Undefined material type")
            mpm_model.material = _MATERIALS[kwargs["material"]]
        if "yield_stress" in kwargs:
            self._before_caller_write(mpm_model)  # results first (hardening updates yield_stress), then the new value
            mpm_model._raw("yield_stress").fill_(float(kwargs["yield_stress"]))
            mpm_model._touch()
        if "hardening" in kwargs:
            mpm_model.hardening = kwargs["hardening"]
        if "xi" in kwargs:
            mpm_model.xi = kwargs["xi"]
        if "friction_angle" in kwargs:
            mpm_model.friction_angle = kwargs["friction_angle"]
            sin_phi = math.sin(mpm_model.friction_angle / 180.0 * 3.14159265)
            mpm_model.friction_coeff = math.tan(mpm_model.friction_angle / 180.0 * 3.14159265)
            mpm_model.alpha = math.sqrt(2.0 / 3.0) * 2.0 * sin_phi / (3.0 - sin_phi)
        if "g" in kwargs:
            mpm_model.gravitational_accelaration = (kwargs["g"][0], kwargs["g"][1], kwargs["g"][2])
        if "density" in kwargs:
            self._before_caller_write(mpm_state)
            mpm_state._raw("particle_density").fill_(float(kwargs["density"]))
            torch.mul(mpm_state._raw("particle_density"), mpm_state._raw("particle_vol"), out=mpm_state._raw("particle_mass"))
            mpm_state._touch()
        for key in ("rpic_damping", "plastic_viscosity", "softening", "grid_v_damping_scale"):
            if key in kwargs:
                setattr(mpm_model, key, kwargs[key])

    # mpm_solver.py:128-187
    def set_E_nu(self, mpm_model, E, nu, gamma, kappa, device="cuda:0"):
        self._before_caller_write(mpm_model)
        for name, val in (("E", E), ("nu", nu), ("gamma", gamma), ("kappa", kappa)):
            dst = mpm_model._raw(name)
            if isinstance(val, float):
                dst.fill_(val)
            else:
                dst.copy_(val.detach().to(device=dst.device, dtype=torch.float32).reshape(-1)[:dst.numel()])
        mpm_model._touch()

    # mpm_solver.py:189-218
    def set_E_nu_from_torch(self, mpm_model, E, nu, gamma, kappa, device="cuda:0"):
        conv = lambda t: t.item() if t.ndim == 0 else t
        self.set_E_nu(mpm_model, conv(E), conv(nu), conv(gamma), conv(kappa), device=device)

    # mpm_solver.py:220-227 -> compute_mu_lam_from_E_nu, mpm_utils.py:402-408
    def prepare_mu_lam(self, mpm_model, mpm_state, device="cuda:0"):
        mpm_model.finalize_mu_lam(self.n_particles, device=device)

    # ------------------------------------------------------------------ the substep
    # mpm_solver.py:229-536
    def p2g2p(self, mpm_model, mpm_state, dt, mesh_x=None, mesh_v=None, joint_traditional_v=None, joint_verts_v=None,
              joint_faces_v=None, device="cuda:0"):
        self._step(mpm_model, mpm_state, dt, 1, mesh_x, mesh_v, joint_traditional_v, joint_verts_v, joint_faces_v)

    step = p2g2p  # BASELINE.json's prose name

    def p2g2p_n(self, mpm_model, mpm_state, dt, n, mesh_x=None, mesh_v=None, joint_traditional_v=None,
                joint_verts_v=None, joint_faces_v=None, device="cuda:0"):
        """``n`` substeps with the caller's mesh advection ``mesh_x + k*dt*mesh_v`` done on the device
        (the loop at train_material_params.py:622-626 / run_demo.py:526-530 as one C call)."""
        self._step(mpm_model, mpm_state, dt, n, mesh_x, mesh_v, joint_traditional_v, joint_verts_v, joint_faces_v,
                   fused=True)

    def _step(self, mpm_model, mpm_state, dt, n, mesh_x, mesh_v, jt, jv, jf, fused=False):
        if dt != self._host_dt:   # MPMWARP.time advances by the Python float, the kernels get fp32 (mpm_solver.py:536)
            self._call("mpmhip_set_host_dt", float(dt))
            self._host_dt = dt
        self._push_if_modified()
        self._bind(mpm_model, mpm_state)
        has_mesh = hasattr(self, "mesh")
        mx = self._ptr(mesh_x, self.num_mesh_v if has_mesh else None, "mesh_x")
        mv = self._ptr(mesh_v, self.num_mesh_v if has_mesh else None, "mesh_v")
        jt = self._ptr(jt)
        jv = self._ptr(jv, self.num_joint_v, "joint_verts_v")
        jf = self._ptr(jf, self.num_joint_f, "joint_faces_v")
        n_jt = 0 if jt is None else int(jt.shape[0])
        self._keep = (mx, mv, jt, jv, jf)
        dp = lambda t: None if (t is None or t.numel() == 0) else t.data_ptr()
        # an empty joint tensor is still "not None" for the reference's mover guard (mpm_solver.py:421)
        jvp = dp(jv) if jv is None or jv.numel() else self._dummy_ptr()
        jfp = dp(jf) if jf is None or jf.numel() else self._dummy_ptr()
        if fused:
            self._call("mpmhip_steps", float(dt), int(n), dp(mx), dp(mv), dp(jt), n_jt, jvp, jfp)
        else:
            self._call("mpmhip_step", float(dt), dp(mx), dp(mv), dp(jt), n_jt, jvp, jfp)
        if self._profiling:
            self._collect_profile()
        self._stale = True   # tensors handed out earlier are behind the solver now (until a read or _refresh_held pulls)
        if self._read_snaps:
            self._refresh_held()



    def _dummy_ptr(self):
        if not hasattr(self, "_dummy"):
            self._dummy = torch.zeros(4, dtype=torch.float32, device=self.device)
        return self._dummy.data_ptr()

    def synchronize(self):
        self._call("mpmhip_synchronize")

    # ------------------------------------------------------------------ profiling (mpm_solver.py:16,538-541)
    def enable_profiling(self, on=True, fused=False):
        """on: fill ``time_profile`` like the reference's ScopedTimers (one launch + one sync per reference phase).
        fused=True: time the launches of the production loop instead (same kernels as an unprofiled run)."""
        self._profiling = bool(on)
        self._lib.mpmhip_profile_enable(self._ctx, (2 if fused else 1) if on else 0)

    def _collect_profile(self):
        name, ms, cnt = C.c_char_p(), C.c_double(), C.c_int64()
        for i in range(self._lib.mpmhip_profile_count(self._ctx)):
            self._lib.mpmhip_profile_get(self._ctx, i, C.byref(name), C.byref(ms), C.byref(cnt))
            if cnt.value > 0:  # one entry per bracketed launch (a fused call of n substeps brackets n launches per phase)
                self.time_profile.setdefault(name.value.decode(), []).extend([ms.value / cnt.value] * cnt.value)
            self._lib.mpmhip_profile_get_kernel(self._ctx, i, C.byref(ms), C.byref(cnt))
            if cnt.value > 0:  # (fused=True) the launch's own start -> stop time, without the event bracket's cost
                self.kernel_profile.setdefault(name.value.decode(), []).extend([ms.value / cnt.value] * cnt.value)
        self._lib.mpmhip_profile_reset(self._ctx)

    def print_time_profile(self):
        print("MPM Time profile:")
        for key, value in self.time_profile.items():
            print(key, sum(value))

    # mpm_solver.py:543-561 (kernel compute_cov_from_F, mpm_utils.py:1108-1132)
    def export_particle_cov_to_torch(self, mpm_state, device="cuda:0"):
        n = self.n_no_vertices
        Ft, cov0 = mpm_state.particle_F_trial, mpm_state.particle_cov   # reading F_trial writes the solver's results back first
        if cov0 is None or cov0.numel() < 6 * n:
            raise RuntimeError("export_particle_cov_to_torch: particle_cov holds fewer than 6 * n_no_vertices values")
        # the caller ASSIGNS particle_cov (state.particle_cov = ...): whatever it is, the kernel gets contiguous fp32 on F_trial's device
        if cov0.dtype != torch.float32 or cov0.device != Ft.device or not cov0.is_contiguous():
            cov0 = cov0.to(device=Ft.device, dtype=torch.float32).contiguous()
        if torch.device(device).type == "cuda" and torch.device(device) != Ft.device and torch.device(device).index is not None:
            raise RuntimeError(f"export_particle_cov_to_torch: the state lives on {Ft.device}, not on {device}")
        new_cov = torch.zeros(n * 6, dtype=torch.float32, device=Ft.device)
        if n:
            dev = Ft.device
            self.synchronize()   # F_trial was written back on the context's stream; the launch below goes on torch's current stream
            rc = self._lib.mpmhip_cov_from_F(dev.index or 0, torch.cuda.current_stream(dev).cuda_stream, Ft.data_ptr(),
                                             cov0.data_ptr(), n, new_cov.data_ptr())
            if rc != 0:
                raise RuntimeError(f"mpmhip_cov_from_F failed ({rc})")
        return new_cov

    # ------------------------------------------------------------------ introspection (not in the reference)
    def export_grid(self):
        """Dense reference-layout (grid_m [G,G,G], grid_v_in [G,G,G,3], grid_v_out [G,G,G,3]) copies."""
        G = self.n_grid
        m = torch.empty(G, G, G, dtype=torch.float32, device=self.device)
        vi = torch.empty(G, G, G, 3, dtype=torch.float32, device=self.device)
        vo = torch.empty(G, G, G, 3, dtype=torch.float32, device=self.device)
        self._call("mpmhip_export_grid", m.data_ptr(), vi.data_ptr(), vo.data_ptr())
        return m, vi, vo

    def stats(self):
        s = L.Stats()
        self._call("mpmhip_get_stats", C.byref(s))
        return {k: getattr(s, k) for k, _ in L.Stats._fields_}

    # ------------------------------------------------------------------ colliders / BCs
    # mpm_solver.py:564-658
    def add_surface_collider(self, point, normal, surface="sticky", friction=0.0, start_time=0.0, end_time=999.0):
        point = list(point)
        normal_scale = 1.0 / math.sqrt(float(sum(x ** 2 for x in normal)))
        normal = list(normal_scale * x for x in normal)
        if surface == "sticky" and friction != 0:
            raise ValueError("friction must be 0 on sticky surfaces.")
        st = {"sticky": 0, "slip": 1, "cut": 11}.get(surface, 2)
        self._call("mpmhip_add_surface_collider", _f3(point), _f3(normal), st, float(friction), float(start_time),
                   float(end_time))
        self.collider_params.append(dict(kind="surface", point=point, normal=normal, surface_type=st, friction=friction,
                                         start_time=start_time, end_time=end_time))
        self.grid_postprocess.append("surface_collider")
        self.modify_bc.append(None)

    # mpm_solver.py:661-802
    def add_particle_mover(self, n_grid):
        if n_grid != self.n_grid:
            raise ValueError("add_particle_mover: n_grid must equal the solver's grid resolution")
        self._call("mpmhip_add_particle_mover")
        self.particle_movers.append("particle_mover")
        self.particle_mover_params.append({})

    # mpm_solver.py:805-919
    def add_mesh_collider(self, mesh_id, n_grid, friction=0.0):
        if not hasattr(self, "mesh") or mesh_id != self.mesh.id:
            raise ValueError("add_mesh_collider: unknown mesh id (pass mpm_solver.mesh.id)")
        if n_grid != self.n_grid:
            raise ValueError("add_mesh_collider: n_grid must equal the solver's grid resolution")
        self._call("mpmhip_add_mesh_collider", float(friction))
        self.mesh_colliders.append("mesh_collider")
        self.mesh_collider_params.append(dict(mesh_id=mesh_id, friction=friction))

    # mpm_solver.py:929-984
    def set_velocity_on_cuboid(self, point, size, velocity, start_time=0.0, end_time=999.0, reset=0):
        self._call("mpmhip_add_velocity_cuboid", _f3(point), _f3(size), _f3(velocity), float(start_time),
                   float(end_time), int(reset))
        self.collider_params.append(dict(kind="cuboid", point=list(point), size=list(size), velocity=list(velocity)))
        self.grid_postprocess.append("velocity_cuboid")
        self.modify_bc.append("modify")

    # mpm_solver.py:986-1053
    def add_bounding_box(self, start_time=0.0, end_time=999.0):
        self._call("mpmhip_add_bounding_box", float(start_time), float(end_time))
        self.collider_params.append(dict(kind="bounding_box"))
        self.grid_postprocess.append("bounding_box")
        self.modify_bc.append(None)

    # mpm_solver.py:1330-1355
    def enforce_grid_velocity_by_mask(self, selection_mask: torch.Tensor):
        m = selection_mask.detach().to(device=self.device, dtype=torch.int32).contiguous()
        if m.numel() != self.n_grid ** 3:
            raise RuntimeError("enforce_grid_velocity_by_mask: mask must have n_grid^3 entries")
        self._masks.append(m)
        self._call("mpmhip_add_grid_mask", m.data_ptr())
        self.collider_params.append(dict(kind="grid_mask"))
        self.grid_postprocess.append("grid_mask")
        self.modify_bc.append(None)

    # ------------------------------------------------------------------ pre-p2g particle operations
    def _new_mask(self):
        m = torch.zeros(self.n_particles, dtype=torch.int32, device=self.device)
        self._masks.append(m)
        return m

    def _select_box(self, mpm_state, point, size, mask):
        if mpm_state is not self._bound_state:
            if self._bound_model is None:
                raise RuntimeError("bind the model first: call p2g2p once or register after set_parameters_dict")
            self._bind(self._bound_model, mpm_state)
        self._call("mpmhip_select_box", _f3(point), _f3(size), mask.data_ptr())

    def _ensure_bound_for_selection(self, mpm_state):
        if self._bound_state is not mpm_state or mpm_state._version != self._bound_state_version:
            p = L.StatePtrs()
            for name, _ in L.StatePtrs._fields_:
                t = mpm_state._raw(name)
                setattr(p, name, t.data_ptr() if t.numel() else None)
            self._call("mpmhip_bind_state", C.byref(p))
            mpm_state._attach(self)
            self._bound_state, self._bound_state_version = mpm_state, mpm_state._version

    # mpm_solver.py:1058-1106
    def add_impulse_on_particles(self, mpm_state, force, dt, point=[1, 1, 1], size=[1, 1, 1], num_dt=1, start_time=0.0,
                                 device="cuda:0"):
        self._ensure_bound_for_selection(mpm_state)
        mask = self._new_mask()
        self._call("mpmhip_select_box", _f3(point), _f3(size), mask.data_ptr())
        self._call("mpmhip_add_impulse", _f3(force), mask.data_ptr(), 1, float(start_time), float(start_time + dt * num_dt))
        self.impulse_params.append(dict(force=list(force), mask=mask))
        self.pre_p2g_operations.append("apply_force")

    # mpm_solver.py:1360-1417
    def add_impulse_on_particles_with_mask(self, mpm_state, force, dt, particle_mask, point=[1, 1, 1], size=[1, 1, 1],
                                           end_time=1, start_time=0.0, device="cuda:0"):
        assert len(particle_mask) == self.n_particles, "mask should have n_particles elements"
        self._ensure_bound_for_selection(mpm_state)
        mask = particle_mask.detach().to(device=self.device, dtype=torch.int32).contiguous()
        self._masks.append(mask)
        # the reference re-runs the box selection over the caller's mask (mpm_solver.py:1389-1394)
        self._call("mpmhip_select_box", _f3(point), _f3(size), mask.data_ptr())
        self._call("mpmhip_add_impulse", _f3(force), mask.data_ptr(), 0, float(start_time), float(end_time))
        self.impulse_params.append(dict(force=list(force), mask=mask))
        self.pre_p2g_operations.append("apply_force")

    # mpm_solver.py:1108-1151
    def enforce_particle_velocity_translation(self, mpm_state, point, size, velocity, start_time, end_time,
                                              device="cuda:0"):
        self._ensure_bound_for_selection(mpm_state)
        mask = self._new_mask()
        self._call("mpmhip_select_box", _f3(point), _f3(size), mask.data_ptr())
        self._call("mpmhip_add_velocity_set", _f3(velocity), mask.data_ptr(), float(start_time), float(end_time))
        self.particle_velocity_modifier_params.append(dict(velocity=list(velocity), mask=mask))
        self.particle_velocity_modifiers.append("modify_particle_v_before_p2g")

    # mpm_solver.py:1156-1257
    def enforce_particle_velocity_rotation(self, mpm_state, point, normal, half_height_and_radius, rotation_scale,
                                           translation_scale, start_time, end_time, device="cuda:0"):
        self._ensure_bound_for_selection(mpm_state)
        ns = 1.0 / math.sqrt(float(normal[0] ** 2 + normal[1] ** 2 + normal[2] ** 2))
        normal = np.array([ns * x for x in normal], np.float64)
        h1 = np.array([1.0, 1.0, 1.0])
        if abs(float(normal @ h1)) < 0.01:
            h1 = np.array([0.72, 0.37, -0.67])
        h1 = h1 - (h1 @ normal) * normal
        h1 = h1 * (1.0 / np.linalg.norm(h1))
        h2 = np.cross(h1, normal)
        mask = self._new_mask()
        self._call("mpmhip_select_cylinder", _f3(point), _f3(normal), float(half_height_and_radius[0]),
                   float(half_height_and_radius[1]), mask.data_ptr())
        self._call("mpmhip_add_velocity_rotation", _f3(point), _f3(normal), _f3(h1), _f3(h2), float(rotation_scale),
                   float(translation_scale), mask.data_ptr(), float(start_time), float(end_time))
        self.particle_velocity_modifier_params.append(dict(mask=mask))
        self.particle_velocity_modifiers.append("modify_particle_v_before_p2g")

    # mpm_solver.py:1261-1287
    def release_particles_sequentially(self, mpm_state, normal, start_position, end_position, num_layers, start_time,
                                       end_time):
        num_layers = 50
        point, size, axis = [0, 0, 0], [0, 0, 0], -1
        for i in range(3):
            if normal[i] == 0:
                point[i] = 1
                size[i] = 1
            else:
                axis = i
                point[i] = end_position
        half_length_portion = abs(start_position - end_position) / num_layers
        end_time_portion = end_time / num_layers
        for i in range(num_layers):
            size[axis] = half_length_portion * (num_layers - i)
            self.enforce_particle_velocity_translation(mpm_state=mpm_state, point=point, size=size, velocity=[0, 0, 0],
                                                       start_time=start_time, end_time=end_time_portion * (i + 1))

    # mpm_solver.py:1289-1328
    def enforce_particle_velocity_by_mask(self, mpm_state, selection_mask: torch.Tensor, velocity, start_time, end_time):
        mask = selection_mask.detach().to(device=self.device, dtype=torch.int32).contiguous()
        if mask.numel() != self.n_particles:
            raise RuntimeError("enforce_particle_velocity_by_mask: mask must have n_particles entries")
        self._masks.append(mask)
        self._call("mpmhip_add_velocity_set", _f3(velocity), mask.data_ptr(), float(start_time), float(end_time))
        self.particle_velocity_modifier_params.append(dict(velocity=list(velocity), mask=mask))
        self.particle_velocity_modifiers.append("modify_particle_v_before_p2g")
