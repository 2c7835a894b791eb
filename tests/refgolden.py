"""Helpers for the fixtures that the REFERENCE'S OWN SOURCE produced (tests/golden/ref_*.npz, generated in the build
container by tests/golden/make_golden_ref.py: /root/reference/warp_mpm/*.py imported unchanged over a NumPy stand-in of
the ``warp`` module).  Only data is read here; nothing of the reference is needed at test time.
"""
from __future__ import annotations

import glob
import json
import os

import numpy as np

from mpmavatar_amd.scenes import Scene

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")

_ARRAYS = ("x", "v", "vol", "faces", "d", "R_inv", "mesh_vertices", "mesh_faces", "mesh_v", "joint_verts_v", "joint_faces_v",
           "selection")


def names(kind):
    return sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(GOLDEN, f"ref_{kind}_*.npz")))


def load(name):
    return np.load(os.path.join(GOLDEN, name + ".npz"))


def scene_from_npz(z) -> Scene:
    meta = json.loads(str(z["scene_meta"]))
    kw = {k: (z["scene_" + k] if "scene_" + k in z.files else None) for k in _ARRAYS}
    bcs = [(kind, d) for kind, d in meta.pop("bcs")]
    return Scene(params=meta.pop("params"), bcs=bcs, **meta, **kw)


def rel(a, b, floor=1e-3):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    if a.size == 0:
        return 0.0
    return float(np.abs(a - b).max() / max(np.abs(b).max(), floor))


def rel_pp(a, b, floor=1e-3):
    """SURVEY 8(d)'s parity metric: max_i |a_i - b_i| / max(|b_i|, floor) with per-PARTICLE norms (rows), so that a slow
    particle next to a fast one is held to its own magnitude (`rel` divides by the global maximum).  The floor is absolute
    (1e-3 in the field's unit), as 8(d) writes it."""
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    if a.size == 0:
        return 0.0
    a, b = a.reshape(len(a), -1), b.reshape(len(b), -1)
    return float((np.linalg.norm(a - b, axis=1) / np.maximum(np.linalg.norm(b, axis=1), floor)).max())


def rel_pp_scaled(a, b, frac=1e-3):
    """The same with the floor relative to the fastest particle: max_i |da_i| / max(|b_i|, frac * max_j |b_j|)."""
    b64 = np.asarray(b, np.float64).reshape(len(b), -1)
    return rel_pp(a, b, floor=max(frac * float(np.linalg.norm(b64, axis=1).max()) if b64.size else 0.0, 1e-30))


ENVELOPE_FACTOR = 1.5


def seq_bound(z, cp, field="particle_v"):
    """1e-4 (north star), or -- where the reference's own trajectory is that sensitive -- 1.5 x the larger of two distances of
    the reference from ITSELF: fp64-accurate vs fp32-accurate svd3 / qr3 (``alt_`` arrays), and the same particles enumerated
    in another order, i.e. another summation order of the atomic_adds (``alt2_`` arrays).  (Rounds 1-2: 3 x; tightened when
    the cloth QR of the HIP path was made the oracle's, tests/test_hip_math_on_host.py.)"""
    envs = [rel(z[f"{tag}_s{cp}_{field}"], z[f"s{cp}_{field}"]) for tag in ("alt", "alt2") if f"{tag}_s{cp}_{field}" in z.files]
    return max([1e-4] + [ENVELOPE_FACTOR * e for e in envs])


PRE_OPS = {"impulse": "add_impulse_on_particles", "vel_translation": "enforce_particle_velocity_translation",
           "vel_rotation": "enforce_particle_velocity_rotation"}


def launches(z):
    return json.loads(str(z["launches"]))


def pre_ops(z):
    return json.loads(str(z["preops_json"]))


def state_after(z, upto):
    """The reference's complete state after launch number `upto` (-1: before the traced substep)."""
    st = {k[4:]: z[k] for k in z.files if k.startswith("pre_")}
    for i in range(upto + 1):
        pref = f"L{i:02d}_"
        for k in z.files:
            if k.startswith(pref):
                st[k[len(pref):]] = z[k]
    return st


def step_inputs(sc, step):
    """p2g2p arguments at substep `step` (NumPy), as oracle.scene_adapter.run_scene builds them."""
    kw = {}
    if sc.mesh_vertices is not None:
        mx, mv = sc.body_at(step)
        kw["mesh_x"], kw["mesh_v"] = np.ascontiguousarray(mx, np.float32), np.ascontiguousarray(mv, np.float32)
    if sc.joint_verts_v is not None:
        jv, jf = sc.joints_at(step)
        kw["joint_verts_v"] = np.ascontiguousarray(jv, np.float32)
        kw["joint_faces_v"] = np.ascontiguousarray(jf, np.float32).reshape(-1, 3)
    if sc.joint_t_hold > 0:
        kw["joint_traditional_v"] = np.zeros((sc.joint_t_count(step), 3), np.float32)
    return kw
