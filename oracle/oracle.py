"""ctypes front-end of the CPU oracle (TEST INFRASTRUCTURE ONLY).

Only tests/, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import this
module.  It wraps ``oracle/mpm_oracle.c`` -- the serial fp32 restatement of the reference substep
``MPMWARP.p2g2p`` (/root/reference/warp_mpm/mpm_solver.py:229-536).  The reference is NVIDIA-Warp DSL and
cannot run in this image; the restatement is pinned by fixtures the reference's own kernel bodies produced over a NumPy stand-in of
the ``warp`` module (header of ``mpm_oracle.h``, PINNING; tests/test_ref_golden.py).

All particle/grid memory is owned by NumPy arrays held on the :class:`OracleMPM` instance, in the
reference's AoS layout, so tests can read/write any field between kernels.
"""
from __future__ import annotations

import ctypes as C
import math
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_BUILD = os.path.join(_HERE, "_build")

MAX_BC, MAX_MC, MAX_MV, MAX_PRE = 16, 4, 2, 64
BC_SURFACE, BC_CUBOID, BC_BBOX, BC_GRIDMASK = 0, 1, 2, 3
PRE_IMPULSE, PRE_IMPULSE_MASK, PRE_VEL_SET, PRE_VEL_ROTATE = 0, 1, 2, 3

MATERIALS = {"jelly": 0, "metal": 1, "sand": 2, "foam": 3, "snow": 4, "plasticine": 5,
             "neo-hookean": 6, "cloth": 7}  # mpm_solver.py:58-76

fp = C.POINTER(C.c_float)
ip = C.POINTER(C.c_int32)
f3 = C.c_float * 3


class _BC(C.Structure):
    _fields_ = [("type", C.c_int32), ("surface_type", C.c_int32), ("reset", C.c_int32), ("pad_", C.c_int32),
                ("point", f3), ("normal", f3), ("size", f3), ("velocity", f3), ("friction", C.c_float),
                ("start_time", C.c_float), ("end_time", C.c_float), ("mask", ip)]


class _Pre(C.Structure):
    _fields_ = [("type", C.c_int32), ("pad_", C.c_int32), ("start_time", C.c_float), ("end_time", C.c_float),
                ("force", f3), ("velocity", f3), ("point", f3), ("normal", f3), ("axis1", f3), ("axis2", f3),
                ("rotation_scale", C.c_float), ("translation_scale", C.c_float), ("mask", ip)]


class _MC(C.Structure):
    _fields_ = [("friction", C.c_float), ("weight", fp), ("v_in", fp), ("v_out", fp), ("normal", fp)]


class _MV(C.Structure):
    _fields_ = [("weight", fp), ("velocity", fp)]


class _Sim(C.Structure):
    _fields_ = [
        ("n_particles", C.c_int32), ("n_elements", C.c_int32), ("n_vertices", C.c_int32), ("n_grid", C.c_int32),
        ("grid_lim", C.c_float), ("dx", C.c_float), ("inv_dx", C.c_float),
        ("x", fp), ("v", fp), ("C", fp), ("F", fp), ("F_trial", fp), ("stress", fp),
        ("d", fp), ("R_inv", fp), ("faces", fp), ("vertex_force", fp),
        ("vol", fp), ("mass", fp), ("density", fp), ("selection", ip),
        ("grid_m", fp), ("grid_v_in", fp), ("grid_v_out", fp),
        ("E", fp), ("nu", fp), ("mu", fp), ("lam", fp), ("gamma", fp), ("kappa", fp), ("yield_stress", fp),
        ("material", C.c_int32), ("friction_coeff", C.c_float), ("alpha", C.c_float), ("g", f3),
        ("hardening", C.c_float), ("xi", C.c_float), ("plastic_viscosity", C.c_float), ("softening", C.c_float),
        ("rpic_damping", C.c_float), ("grid_v_damping_scale", C.c_float),
        ("num_mesh_v", C.c_int32), ("num_mesh_f", C.c_int32),
        ("mesh_points", fp), ("mesh_velocities", fp), ("mesh_indices", ip),
        ("n_mesh_colliders", C.c_int32), ("mesh_colliders", _MC * MAX_MC),
        ("n_movers", C.c_int32), ("movers", _MV * MAX_MV),
        ("num_joint_v", C.c_int32), ("num_joint_f", C.c_int32),
        ("n_bc", C.c_int32), ("bc", _BC * MAX_BC),
        ("n_pre", C.c_int32), ("pre", _Pre * MAX_PRE),
        ("time", C.c_double), ("n_threads", C.c_int32),
        ("box_mode", C.c_int32), ("box_lo", C.c_int32 * 3), ("box_hi", C.c_int32 * 3),
    ]


def build(force: bool = False) -> None:
    """Compile liboracle.so / liboracle_omp.so with gcc (oracle/Makefile)."""
    if force:
        subprocess.check_call(["make", "-C", _HERE, "clean"], stdout=subprocess.DEVNULL)
    subprocess.check_call(["make", "-C", _HERE], stdout=subprocess.DEVNULL)


_libs: dict = {}


def _load(omp: bool, fma: bool = False):
    key = ("omp_fma" if fma else "omp") if omp else "serial"
    if key in _libs:
        return _libs[key]
    path = os.path.join(_BUILD, {"serial": "liboracle.so", "omp": "liboracle_omp.so", "omp_fma": "liboracle_omp_fma.so"}[key])
    src = os.path.join(_HERE, "mpm_oracle.c")
    if not os.path.exists(path) or os.path.getmtime(path) < os.path.getmtime(src):
        build()
    lib = C.CDLL(path)
    lib.orc_sizeof_sim.restype = C.c_int
    assert lib.orc_sizeof_sim() == C.sizeof(_Sim), "ctypes mirror of orc_sim out of date"
    sp = C.POINTER(_Sim)
    for name, args in {
        "orc_zero_grid": [sp], "orc_pre_p2g": [sp, C.c_float], "orc_compute_stress_from_F_trial": [sp, C.c_float], "orc_p2g": [sp, C.c_float],
        "orc_grid_normalization_and_gravity": [sp, C.c_float], "orc_add_damping_via_grid": [sp, C.c_float],
        "orc_mesh_collide": [sp, C.c_int], "orc_particle_move": [sp, C.c_int, fp, C.c_int, fp, fp],
        "orc_apply_bc": [sp, C.c_int, C.c_float], "orc_g2p_v": [sp, C.c_float], "orc_g2p_e": [sp, C.c_float],
        "orc_p2g2p": [sp, C.c_double, fp, fp, fp, C.c_int, fp, fp],
        "orc_p2g2p_n": [sp, C.c_double, C.c_int, fp, fp, fp, C.c_int, fp, fp],
        "orc_svd3": [fp, fp, fp, fp], "orc_qr_signfixed": [fp, fp, fp],
        "orc_anisotropy_return_mapping": [fp, C.c_float, C.c_float, C.c_float, fp],
        "orc_kirchhoff_anisotropy": [fp, fp, C.c_float, C.c_float, C.c_float, C.c_float, C.c_float, fp, fp, fp, fp],
        "orc_stencil": [fp, C.c_float, ip, fp, fp],
    }.items():
        fn = getattr(lib, name)
        fn.argtypes = args
        fn.restype = None
    _libs[key] = lib
    return lib


def _p(a):
    if a is None:
        return None
    assert a.flags["C_CONTIGUOUS"]
    if a.dtype == np.float32:
        return a.ctypes.data_as(fp)
    if a.dtype == np.int32:
        return a.ctypes.data_as(ip)
    raise TypeError(a.dtype)


def _f32(a, shape=None):
    a = np.ascontiguousarray(np.asarray(a, dtype=np.float32))
    if shape is not None:
        a = a.reshape(shape)
    return a


# ---------------------------------------------------------------- pure-function wrappers (KATs)
def svd3(A):
    lib = _load(False)
    A = _f32(A, (3, 3)); U = np.zeros((3, 3), np.float32); s = np.zeros(3, np.float32); V = np.zeros((3, 3), np.float32)
    lib.orc_svd3(_p(A), _p(U), _p(s), _p(V))
    return U, s, V


def qr_signfixed(d):
    lib = _load(False)
    d = _f32(d, (3, 3)); Q = np.zeros((3, 3), np.float32); R = np.zeros((3, 3), np.float32)
    lib.orc_qr_signfixed(_p(d), _p(Q), _p(R))
    return Q, R


def anisotropy_return_mapping(d, gamma, kappa, friction_coeff):
    lib = _load(False)
    d = _f32(d, (3, 3)); o = np.zeros((3, 3), np.float32)
    lib.orc_anisotropy_return_mapping(_p(d), gamma, kappa, friction_coeff, _p(o))
    return o


def kirchhoff_anisotropy(R_inv, d, vol, mu, lam, gamma, kappa):
    lib = _load(False)
    R_inv = _f32(R_inv, (3,)); d = _f32(d, (3, 3))
    S = np.zeros((3, 3), np.float32); f1 = np.zeros(3, np.float32); f2 = np.zeros(3, np.float32); f3_ = np.zeros(3, np.float32)
    lib.orc_kirchhoff_anisotropy(_p(R_inv), _p(d), vol, mu, lam, gamma, kappa, _p(S), _p(f1), _p(f2), _p(f3_))
    return S, f1, f2, f3_


def stencil(x, inv_dx):
    lib = _load(False)
    x = _f32(x, (3,)); base = np.zeros(3, np.int32); w = np.zeros((3, 3), np.float32); dw = np.zeros((3, 3), np.float32)
    lib.orc_stencil(_p(x), inv_dx, _p(base), _p(w), _p(dw))
    return base, w, dw  # w[axis, node]


# ---------------------------------------------------------------- the simulator
class OracleMPM:
    """CPU oracle with the reference's state/model fields as NumPy arrays.

    Mirrors what ``MPMStateStruct.init`` + ``MPMModelStruct.init/init_other_params`` + ``MPMWARP.__init__``
    allocate (mpm_data_structure.py:51-156,647-715; mpm_solver.py:18-51).
    """

    def __init__(self, n_particles, n_elements, n_vertices, n_grid=100, grid_lim=1.0, mesh_vertices=None,
                 mesh_faces=None, num_joint_t=0, num_joint_v=0, num_joint_f=0, omp=False, n_threads=1, fma=False):
        self.lib = _load(omp, fma and omp)   # fma: the FMA-contracted OpenMP build (ensemble member, see oracle/Makefile)
        self.n_particles, self.n_elements, self.n_vertices = n_particles, n_elements, n_vertices
        self.n_nv = n_nv = n_particles - n_vertices
        self.n_grid, self.grid_lim = n_grid, grid_lim
        G3 = n_grid ** 3
        z = lambda *s: np.zeros(s, np.float32)
        self.x, self.v, self.C = z(n_particles, 3), z(n_particles, 3), z(n_particles, 3, 3)
        self.F, self.F_trial, self.stress = z(n_nv, 3, 3), z(n_nv, 3, 3), z(n_nv, 3, 3)
        self.d, self.R_inv, self.faces = z(n_elements, 3, 3), z(n_elements, 3), z(n_elements, 3)
        self.vertex_force = z(n_vertices, 3)
        self.vol, self.mass, self.density = z(n_particles), z(n_particles), z(n_particles)
        self.selection = np.zeros(n_particles, np.int32)
        self.grid_m, self.grid_v_in, self.grid_v_out = z(G3), z(G3, 3), z(G3, 3)
        self.E, self.nu, self.mu, self.lam = z(n_particles), z(n_particles), z(n_particles), z(n_particles)
        self.gamma, self.kappa, self.yield_stress = z(n_particles), z(n_particles), z(n_particles)
        s = self.sim = _Sim()
        s.n_particles, s.n_elements, s.n_vertices, s.n_grid = n_particles, n_elements, n_vertices, n_grid
        s.grid_lim = grid_lim
        s.dx, s.inv_dx = grid_lim / n_grid, float(n_grid / grid_lim)  # mpm_data_structure.py:692-697
        # init_other_params defaults, mpm_data_structure.py:699-715
        s.material = 0
        s.plastic_viscosity, s.softening = 0.0, 0.1
        self.set_friction_angle(0.0)
        s.g = f3(0.0, 0.0, 0.0)
        s.rpic_damping, s.grid_v_damping_scale = 0.0, 1.1
        s.hardening, s.xi = 0.0, 0.0
        s.num_joint_v, s.num_joint_f = num_joint_v, num_joint_f
        s.n_threads = n_threads
        self._keep = []
        if mesh_vertices is not None and mesh_faces is not None:
            self.mesh_points = _f32(mesh_vertices, (-1, 3)).copy()
            self.mesh_velocities = np.zeros_like(self.mesh_points)
            self.mesh_indices = np.ascontiguousarray(np.asarray(mesh_faces, np.int32).reshape(-1))
            s.num_mesh_v, s.num_mesh_f = self.mesh_points.shape[0], self.mesh_indices.size // 3
            s.mesh_points, s.mesh_velocities, s.mesh_indices = _p(self.mesh_points), _p(self.mesh_velocities), _p(self.mesh_indices)
        self.mesh_colliders, self.movers = [], []
        self.rebind()

    def rebind(self):
        """(Re)publish array pointers after an attribute was replaced by a new array."""
        s = self.sim
        for name in ("x", "v", "C", "F", "F_trial", "stress", "d", "R_inv", "faces", "vertex_force", "vol", "mass",
                     "density", "selection", "grid_m", "grid_v_in", "grid_v_out", "E", "nu", "mu", "lam", "gamma",
                     "kappa", "yield_stress"):
            a = getattr(self, name)
            want = np.int32 if name == "selection" else np.float32
            if a.dtype != want or not a.flags["C_CONTIGUOUS"]:
                a = np.ascontiguousarray(a, dtype=want)
                setattr(self, name, a)
            setattr(s, name, _p(a))

    # ------------------------------------------------------------ parameters (mpm_solver.py:57-227)
    def set_friction_angle(self, angle):
        s = self.sim  # mpm_solver.py:90-94 (note 3.14159265, not math.pi)
        sin_phi = math.sin(angle / 180.0 * 3.14159265)
        s.friction_coeff = math.tan(angle / 180.0 * 3.14159265)
        s.alpha = math.sqrt(2.0 / 3.0) * 2.0 * sin_phi / (3.0 - sin_phi)

    def set_parameters_dict(self, kwargs):
        s = self.sim
        if "material" in kwargs:
            if kwargs["material"] not in MATERIALS:
                raise TypeError("Undefined material type")
            s.material = MATERIALS[kwargs["material"]]
        if "yield_stress" in kwargs:
            self.yield_stress[:] = kwargs["yield_stress"]
        for k in ("hardening", "xi", "rpic_damping", "plastic_viscosity", "softening", "grid_v_damping_scale"):
            if k in kwargs:
                setattr(s, k, kwargs[k])
        if "friction_angle" in kwargs:
            self.set_friction_angle(kwargs["friction_angle"])
        if "g" in kwargs:
            s.g = f3(*kwargs["g"])
        if "density" in kwargs:
            self.density[:] = kwargs["density"]
            self.mass[:] = self.density * self.vol

    def prepare_mu_lam(self):  # mpm_utils.py:402-408
        E, nu = self.E, self.nu
        one, two = np.float32(1.0), np.float32(2.0)
        self.mu[:] = E / (two * (one + nu))
        self.lam[:] = E * nu / ((one + nu) * (one - two * nu))

    def reset_state(self):  # mpm_data_structure.py:341-374
        self.C[:] = 0
        self.F[:] = np.eye(3, dtype=np.float32)
        self.F_trial[:] = np.eye(3, dtype=np.float32)
        self.stress[:] = 0
        self.vertex_force[:] = 0

    # ------------------------------------------------------------ colliders / BCs
    def add_mesh_collider(self, friction=0.0):
        G3 = self.n_grid ** 3
        arrs = dict(weight=np.zeros(G3, np.float32), v_in=np.zeros((G3, 3), np.float32),
                    v_out=np.zeros((G3, 3), np.float32), normal=np.zeros((G3, 3), np.float32))
        k = self.sim.n_mesh_colliders
        mc = self.sim.mesh_colliders[k]
        mc.friction = friction
        mc.weight, mc.v_in, mc.v_out, mc.normal = (_p(arrs[n]) for n in ("weight", "v_in", "v_out", "normal"))
        self.mesh_colliders.append(arrs)
        self.sim.n_mesh_colliders = k + 1

    def add_particle_mover(self):
        G3 = self.n_grid ** 3
        arrs = dict(weight=np.zeros(G3, np.float32), velocity=np.zeros((G3, 3), np.float32))
        k = self.sim.n_movers
        self.sim.movers[k].weight, self.sim.movers[k].velocity = _p(arrs["weight"]), _p(arrs["velocity"])
        self.movers.append(arrs)
        self.sim.n_movers = k + 1

    def _new_bc(self, type_, start_time, end_time):
        k = self.sim.n_bc
        bc = self.sim.bc[k]
        bc.type, bc.start_time, bc.end_time = type_, start_time, end_time
        self.sim.n_bc = k + 1
        return bc

    def add_surface_collider(self, point, normal, surface="sticky", friction=0.0, start_time=0.0, end_time=999.0):
        nrm = 1.0 / math.sqrt(float(sum(x ** 2 for x in normal)))  # mpm_solver.py:575-576
        normal = [nrm * x for x in normal]
        if surface == "sticky" and friction != 0:
            raise ValueError("friction must be 0 on sticky surfaces.")
        bc = self._new_bc(BC_SURFACE, start_time, end_time)
        bc.surface_type = {"sticky": 0, "slip": 1, "cut": 11}.get(surface, 2)
        bc.point, bc.normal, bc.friction = f3(*point), f3(*normal), friction

    def set_velocity_on_cuboid(self, point, size, velocity, start_time=0.0, end_time=999.0, reset=0):
        bc = self._new_bc(BC_CUBOID, start_time, end_time)
        bc.point, bc.size, bc.velocity, bc.reset = f3(*point), f3(*size), f3(*velocity), reset

    def add_bounding_box(self, start_time=0.0, end_time=999.0):
        self._new_bc(BC_BBOX, start_time, end_time)

    def enforce_grid_velocity_by_mask(self, mask):
        mask = np.ascontiguousarray(np.asarray(mask, np.int32).reshape(-1))
        self._keep.append(mask)
        bc = self._new_bc(BC_GRIDMASK, 0.0, 0.0)
        bc.mask = _p(mask)

    def _new_pre(self, type_, mask, start_time, end_time):
        mask = np.ascontiguousarray(np.asarray(mask, np.int32).reshape(-1))
        self._keep.append(mask)
        k = self.sim.n_pre
        op = self.sim.pre[k]
        op.type, op.mask, op.start_time, op.end_time = type_, _p(mask), start_time, end_time
        self.sim.n_pre = k + 1
        return op

    def box_mask(self, point, size):  # mpm_utils.py:1198-1227
        off = np.abs(self.x - np.asarray(point, np.float32))
        return np.all(off < np.asarray(size, np.float32), axis=1).astype(np.int32)

    def add_impulse_on_particles(self, force, dt, point=(1, 1, 1), size=(1, 1, 1), num_dt=1, start_time=0.0):
        op = self._new_pre(PRE_IMPULSE, self.box_mask(point, size), start_time, start_time + dt * num_dt)
        op.force = f3(*force)

    def add_impulse_on_particles_with_mask(self, force, particle_mask, end_time=1, start_time=0.0):
        # the reference re-runs the box selection over the caller's mask with the default box
        # point=[1,1,1], size=[1,1,1] (mpm_solver.py:1389-1394); reproduce that overwrite.
        op = self._new_pre(PRE_IMPULSE_MASK, self.box_mask((1, 1, 1), (1, 1, 1)), start_time, end_time)
        op.force = f3(*force)

    def enforce_particle_velocity_translation(self, point, size, velocity, start_time, end_time):
        op = self._new_pre(PRE_VEL_SET, self.box_mask(point, size), start_time, end_time)
        op.velocity = f3(*velocity)

    def enforce_particle_velocity_by_mask(self, mask, velocity, start_time, end_time):
        op = self._new_pre(PRE_VEL_SET, mask, start_time, end_time)
        op.velocity = f3(*velocity)

    def enforce_particle_velocity_rotation(self, point, normal, half_height_and_radius, rotation_scale, translation_scale,
                                           start_time, end_time):
        """mpm_solver.py:1156-1257 (host part :1168-1193: unit normal, two horizontal axes; selection mpm_utils.py:1230-1248)."""
        ns = 1.0 / math.sqrt(float(normal[0] ** 2 + normal[1] ** 2 + normal[2] ** 2))
        nrm = np.array([ns * x for x in normal], np.float32)
        h1 = np.array([1.0, 1.0, 1.0], np.float32)
        if abs(float(nrm @ h1)) < 0.01:
            h1 = np.array([0.72, 0.37, -0.67], np.float32)
        h1 = h1 - np.float32(h1 @ nrm) * nrm
        h1 = h1 * np.float32(1.0 / np.sqrt(np.float32(h1 @ h1)))
        h2 = np.cross(h1, nrm).astype(np.float32)
        off = self.x - np.asarray(point, np.float32)
        dn = off @ nrm
        horiz = np.linalg.norm(off - dn[:, None] * nrm[None], axis=1)
        mask = ((np.abs(dn) < np.float32(half_height_and_radius[0])) & (horiz < np.float32(half_height_and_radius[1]))).astype(np.int32)
        op = self._new_pre(PRE_VEL_ROTATE, mask, start_time, end_time)
        op.point, op.normal, op.axis1, op.axis2 = f3(*point), f3(*nrm), f3(*h1), f3(*h2)
        op.rotation_scale, op.translation_scale = rotation_scale, translation_scale

    # ------------------------------------------------------------ kernels
    def _sp(self):
        return C.byref(self.sim)

    def zero_grid(self): self.lib.orc_zero_grid(self._sp())
    def pre_p2g(self, dt): self.lib.orc_pre_p2g(self._sp(), dt)
    def damping(self, scale): self.lib.orc_add_damping_via_grid(self._sp(), scale)
    def compute_stress(self, dt): self.lib.orc_compute_stress_from_F_trial(self._sp(), dt)
    def p2g(self, dt): self.lib.orc_p2g(self._sp(), dt)
    def grid_update(self, dt): self.lib.orc_grid_normalization_and_gravity(self._sp(), dt)
    def mesh_collide(self, k=0): self.lib.orc_mesh_collide(self._sp(), k)
    def apply_bc(self, k, dt): self.lib.orc_apply_bc(self._sp(), k, dt)
    def g2p_v(self, dt): self.lib.orc_g2p_v(self._sp(), dt)
    def g2p_e(self, dt): self.lib.orc_g2p_e(self._sp(), dt)

    def particle_move(self, k, joint_t_v, joint_v_v, joint_f_v):
        jt = None if joint_t_v is None else _f32(joint_t_v, (-1, 3))
        jv, jf = _f32(joint_v_v, (-1, 3)), _f32(joint_f_v, (-1, 3))
        self.lib.orc_particle_move(self._sp(), k, _p(jt), 0 if jt is None else jt.shape[0], _p(jv), _p(jf))

    def _opt(self, a):
        return None if a is None else _f32(a, (-1, 3))

    def p2g2p(self, dt, mesh_x=None, mesh_v=None, joint_traditional_v=None, joint_verts_v=None, joint_faces_v=None):
        mx, mv, jt, jv, jf = map(self._opt, (mesh_x, mesh_v, joint_traditional_v, joint_verts_v, joint_faces_v))
        self.lib.orc_p2g2p(self._sp(), dt, _p(mx), _p(mv), _p(jt), 0 if jt is None else jt.shape[0], _p(jv), _p(jf))

    def p2g2p_n(self, dt, n, mesh_x=None, mesh_v=None, joint_traditional_v=None, joint_verts_v=None, joint_faces_v=None):
        mx, mv, jt, jv, jf = map(self._opt, (mesh_x, mesh_v, joint_traditional_v, joint_verts_v, joint_faces_v))
        self.lib.orc_p2g2p_n(self._sp(), dt, n, _p(mx), _p(mv), _p(jt), 0 if jt is None else jt.shape[0], _p(jv), _p(jf))

    @property
    def time(self):
        return self.sim.time
