"""ctypes binding of libmpmhip.so (include/mpmhip.h).  No torch types cross this boundary.

The library is built in-tree by ``python -m mpmavatar_amd.build`` (or ``__graft_entry__.build()``).
Loading never builds implicitly and never falls back to another implementation: a missing library
raises ``MPMHipError`` naming the build command.
"""
from __future__ import annotations

import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "lib", "libmpmhip.so")

OK, ERR_INVALID, ERR_NO_DEVICE, ERR_HIP, ERR_STATE, ERR_LIMIT = 0, -1, -2, -3, -4, -5
MODE_FAST, MODE_BASELINE = 0, 1

fp = C.POINTER(C.c_float)
ip = C.POINTER(C.c_int32)
f3 = C.c_float * 3
vp = C.c_void_p


class MPMHipError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"libmpmhip error {code}: {msg}")
        self.code = code


class Config(C.Structure):
    _fields_ = [("n_particles", C.c_int32), ("n_elements", C.c_int32), ("n_vertices", C.c_int32),
                ("n_grid", C.c_int32), ("grid_lim", C.c_float), ("num_joint_t", C.c_int32),
                ("num_joint_v", C.c_int32), ("num_joint_f", C.c_int32), ("device", C.c_int32), ("mode", C.c_int32),
                ("rebin_interval", C.c_int32), ("own_stream", C.c_int32), ("stream", vp), ("p2g_tile", C.c_int32),
                ("reserved_", C.c_int32)]


class StatePtrs(C.Structure):
    _fields_ = [(n, vp) for n in ("particle_x", "particle_v", "particle_C", "particle_F", "particle_F_trial",
                                  "particle_stress", "particle_d", "particle_R_inv", "faces", "vertex_force",
                                  "particle_vol", "particle_mass", "particle_selection")]


class ModelPtrs(C.Structure):
    _fields_ = [(n, vp) for n in ("mu", "lam", "gamma", "kappa", "yield_stress")]


class ModelScalars(C.Structure):
    _fields_ = [("material", C.c_int32), ("friction_coeff", C.c_float), ("alpha", C.c_float), ("g", f3),
                ("hardening", C.c_float), ("xi", C.c_float), ("plastic_viscosity", C.c_float),
                ("softening", C.c_float), ("rpic_damping", C.c_float), ("grid_v_damping_scale", C.c_float)]


class Stats(C.Structure):
    _fields_ = [("substeps", C.c_int64), ("rebins", C.c_int64), ("n_active_blocks", C.c_int32),
                ("n_active_nodes", C.c_int32), ("n_collider_nodes", C.c_int32), ("n_mover_nodes", C.c_int32),
                ("n_fallback_particles", C.c_int32), ("n_dropped", C.c_int32), ("g2p2g_launches", C.c_int64), ("p2g_tile_in_use", C.c_int32), ("kept_collider_substeps", C.c_int32)]


class DistPeer(C.Structure):
    _fields_ = [("n_blocks", C.c_int32), ("blocks", vp), ("halo_send", vp), ("halo_recv", vp),
                ("n_send_p", C.c_int32), ("n_recv_p", C.c_int32), ("n_send_e", C.c_int32), ("n_recv_e", C.c_int32),
                ("send_p", vp), ("recv_p", vp), ("send_e", vp), ("recv_e", vp), ("ghost_send", vp), ("ghost_recv", vp)]


# name -> (restype, argtypes); this table is also what tests/test_abi.py checks against the header
SIGNATURES = {
    "mpmhip_version": (C.c_int, []),
    "mpmhip_device_count": (C.c_int, []),
    "mpmhip_create": (C.c_int, [C.POINTER(Config), C.POINTER(vp)]),
    "mpmhip_destroy": (None, [vp]),
    "mpmhip_last_error": (C.c_char_p, [vp]),
    "mpmhip_bind_state": (C.c_int, [vp, C.POINTER(StatePtrs)]),
    "mpmhip_bind_model": (C.c_int, [vp, C.POINTER(ModelPtrs)]),
    "mpmhip_set_model_scalars": (C.c_int, [vp, C.POINTER(ModelScalars)]),
    "mpmhip_push_state": (C.c_int, [vp]),
    "mpmhip_pull_state": (C.c_int, [vp]),
    "mpmhip_set_body_mesh": (C.c_int, [vp, C.c_int32, C.c_int32, vp, vp]),
    "mpmhip_add_mesh_collider": (C.c_int, [vp, C.c_float]),
    "mpmhip_add_particle_mover": (C.c_int, [vp]),
    "mpmhip_add_surface_collider": (C.c_int, [vp, f3, f3, C.c_int32, C.c_float, C.c_float, C.c_float]),
    "mpmhip_add_velocity_cuboid": (C.c_int, [vp, f3, f3, f3, C.c_float, C.c_float, C.c_int32]),
    "mpmhip_add_bounding_box": (C.c_int, [vp, C.c_float, C.c_float]),
    "mpmhip_add_grid_mask": (C.c_int, [vp, vp]),
    "mpmhip_select_box": (C.c_int, [vp, f3, f3, vp]),
    "mpmhip_select_cylinder": (C.c_int, [vp, f3, f3, C.c_float, C.c_float, vp]),
    "mpmhip_add_impulse": (C.c_int, [vp, f3, vp, C.c_int32, C.c_float, C.c_float]),
    "mpmhip_add_velocity_set": (C.c_int, [vp, f3, vp, C.c_float, C.c_float]),
    "mpmhip_add_velocity_rotation": (C.c_int, [vp, f3, f3, f3, f3, C.c_float, C.c_float, vp, C.c_float, C.c_float]),
    "mpmhip_step": (C.c_int, [vp, C.c_float, vp, vp, vp, C.c_int32, vp, vp]),
    "mpmhip_steps": (C.c_int, [vp, C.c_float, C.c_int32, vp, vp, vp, C.c_int32, vp, vp]),
    "mpmhip_cov_from_F": (C.c_int, [C.c_int32, vp, vp, vp, C.c_int32, vp]),
    "mpmhip_face_frames": (C.c_int, [C.c_int32, vp, vp, vp, C.c_int32, vp, vp, vp, vp]),
    "mpmhip_bind_gaussians": (C.c_int, [C.c_int32, vp, C.c_int32, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp]),
    "mpmhip_render_inputs": (C.c_int, [C.c_int32, vp, C.c_int32, C.c_int32] + [vp] * 18),
    "mpmhip_dist_enable": (C.c_int, [vp]),
    "mpmhip_dist_set_ghost_mode": (C.c_int, [vp, C.c_int32]),
    "mpmhip_dist_set_mass_span": (C.c_int, [vp, C.c_float, C.c_float]),
    "mpmhip_dist_ghost_pack": (C.c_int, [vp]),
    "mpmhip_dist_ghost_unpack": (C.c_int, [vp]),
    "mpmhip_dist_num_blocks": (C.c_int, [vp]),
    "mpmhip_dist_drift_flag": (C.c_int, [vp, C.POINTER(C.c_int32)]),
    "mpmhip_dist_rebin": (C.c_int, [vp, vp]),
    "mpmhip_dist_set_peers": (C.c_int, [vp, C.c_int32, C.POINTER(DistPeer)]),
    "mpmhip_dist_step_begin": (C.c_int, [vp, C.c_float, vp, vp, C.c_float, vp, C.c_int32, vp, vp]),
    "mpmhip_dist_step_mid": (C.c_int, [vp]),
    "mpmhip_dist_step_end": (C.c_int, [vp]),
    "mpmhip_rccl_unique_id": (C.c_int, [C.c_char * 128]),
    "mpmhip_rccl_init": (C.c_int, [vp, C.c_int32, C.c_int32, C.c_char * 128]),
    "mpmhip_rccl_set_ghosts": (C.c_int, [vp, C.c_int32, ip, ip, C.POINTER(ip), ip, C.POINTER(ip), ip, C.POINTER(ip), ip, C.POINTER(ip)]),
    "mpmhip_rccl_steps": (C.c_int, [vp, C.c_float, C.c_int32, C.c_int64, C.c_int32, vp, vp, vp, C.c_int32, vp, vp]),
    "mpmhip_synchronize": (C.c_int, [vp]),
    "mpmhip_get_time": (C.c_double, [vp]),
    "mpmhip_set_time": (C.c_int, [vp, C.c_double]),
    "mpmhip_set_host_dt": (C.c_int, [vp, C.c_double]),
    "mpmhip_set_debug_flags": (C.c_int, [vp, C.c_int32]),
    "mpmhip_debug_counter": (C.c_int, [vp, C.c_int32, C.POINTER(C.c_int64)]),
    "mpmhip_debug_wgtrace": (C.c_int, [vp, C.c_int32, vp, C.c_int32]),
    "mpmhip_debug_sort": (C.c_int, [vp, vp, C.c_int32, C.c_int32, vp, vp]),
    "mpmhip_dist_halo_bytes": (C.c_int, [vp, C.POINTER(C.c_int64)]),
    "mpmhip_dist_halo_transport": (C.c_int, [vp, C.POINTER(C.c_int32)]),
    "mpmhip_dist_fused_halo_steps": (C.c_int, [vp, C.POINTER(C.c_int64)]),
    "mpmhip_export_grid": (C.c_int, [vp, vp, vp, vp]),
    "mpmhip_get_stats": (C.c_int, [vp, C.POINTER(Stats)]),
    "mpmhip_profile_enable": (C.c_int, [vp, C.c_int32]),
    "mpmhip_profile_count": (C.c_int, [vp]),
    "mpmhip_profile_get": (C.c_int, [vp, C.c_int32, C.POINTER(C.c_char_p), C.POINTER(C.c_double), C.POINTER(C.c_int64)]),
    "mpmhip_profile_get_kernel": (C.c_int, [vp, C.c_int32, C.POINTER(C.c_double), C.POINTER(C.c_int64)]),
    "mpmhip_profile_reset": (C.c_int, [vp]),
}

_lib = None


def load():
    """dlopen libmpmhip.so and attach the signatures above."""
    global _lib
    if _lib is not None:
        return _lib
    global LIB_PATH
    LIB_PATH = os.environ.get("MPMHIP_LIB", LIB_PATH)  # kernel A/B experiments: another build of the same sources
    if not os.path.exists(LIB_PATH):
        raise MPMHipError(ERR_INVALID, f"{LIB_PATH} not found: build it with `python -m mpmavatar_amd.build` "
                                       "(hipcc --offload-arch=gfx950); there is no CPU fallback")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the .so does not export a declared symbol
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(lib, ctx, rc):
    if rc != OK:
        msg = lib.mpmhip_last_error(ctx)
        raise MPMHipError(rc, msg.decode() if msg else "?")
