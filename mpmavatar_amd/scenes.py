"""Deterministic synthetic scenes (SURVEY.md 8(d) S1-S4 + reduced test sizes).  NumPy only.

The reference's data (Actor01 meshes, SMPL-X) is not shipped, so every BASELINE.json
config is represented by a seeded synthetic stand-in of the same size and particle-class mix.  A
``Scene`` is a plain description of solver inputs in the reference's conventions (particle order
elements | traditional | vertices; joint entries first) and is consumed by
``mpmavatar_amd.harness.build_solver`` (HIP path) and by ``oracle/scene_adapter.py`` (CPU oracle).
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Optional

import numpy as np

from . import garment


@dataclass
class Scene:
    name: str
    n_grid: int
    grid_lim: float
    n_elements: int
    n_traditional: int
    n_vertices: int
    x: np.ndarray                 # [n_p,3] elements | traditional | vertices
    v: np.ndarray                 # [n_p,3]
    vol: np.ndarray               # [n_p]
    faces: np.ndarray             # [n_e,3] int32, indices local to the vertex block
    d: np.ndarray                 # [n_e,3,3] init_dir
    R_inv: np.ndarray             # [n_e,3]
    density: float = 1.0
    E: float = 100.0
    nu: float = 0.3
    gamma: float = 500.0
    kappa: float = 500.0
    params: dict = field(default_factory=dict)   # kwargs of set_parameters_dict
    mesh_vertices: Optional[np.ndarray] = None   # body / collider mesh
    mesh_faces: Optional[np.ndarray] = None
    mesh_v: Optional[np.ndarray] = None          # per-vertex velocity (constant over a run)
    mesh_friction: float = 0.5
    num_joint_v: int = 0
    num_joint_f: int = 0
    joint_verts_v: Optional[np.ndarray] = None
    joint_faces_v: Optional[np.ndarray] = None
    bcs: list = field(default_factory=list)      # [("bounding_box", {}), ("surface_collider", {...})]
    dt: float = 1e-4
    n_steps: int = 100
    has_mover: Optional[bool] = None             # None: a particle mover is registered iff the scene has joints
    selection: Optional[np.ndarray] = None       # particle_selection (0 simulate, 1 frozen, 2 ghost copy); None = all 0
    # staged release of the trailing traditional particles (run_demo.py:524: the mover holds the sand at zero velocity,
    # then lets go of `joint_t_rate` more particles every `joint_t_every` substeps from substep `joint_t_start` on)
    joint_t_hold: int = 0
    joint_t_start: int = 0
    joint_t_every: int = 1
    joint_t_rate: int = 0
    # swaying body (velocity amplitude * sin(2 pi f t) along x): (amplitude m/s, frequency Hz, substeps per frame).
    # Like the reference drivers (train_material_params.py:616-626) the body is posed once per FRAME and moves with the constant
    # finite-difference velocity to the next pose inside it; joints attached to the body get the same velocity.
    mesh_sway: Optional[tuple] = None

    def joint_t_count(self, step: int) -> int:
        """Length of joint_traditional_v at this substep (max(n_t - max(i - 100, 0) * 1000, 0) per frame in the reference)."""
        if self.joint_t_hold <= 0:
            return 0
        stages = max(step - self.joint_t_start, 0) // max(self.joint_t_every, 1) if step >= self.joint_t_start else 0
        return max(self.joint_t_hold - stages * self.joint_t_rate, 0)

    @property
    def n_particles(self):
        return self.n_elements + self.n_traditional + self.n_vertices

    def frame_of(self, step: int):
        """(first substep of the frame that contains `step`, substeps per frame); (0, None) for a body in uniform motion."""
        if self.mesh_sway is None:
            return 0, None
        spf = int(self.mesh_sway[2])
        return (step // spf) * spf, spf

    def _sway_offset(self, t: float) -> float:
        amp, freq = float(self.mesh_sway[0]), float(self.mesh_sway[1])
        return amp / (2.0 * np.pi * freq) * (1.0 - np.cos(2.0 * np.pi * freq * t))   # integral of amp sin(2 pi f t)

    def body_at(self, step: int):
        """(mesh_x, mesh_v) a caller hands to p2g2p at substep `step`, or (None, None) without a body mesh."""
        if self.mesh_vertices is None:
            return None, None
        if self.mesh_sway is None:
            return (self.mesh_vertices + np.float32(self.dt * step) * self.mesh_v).astype(np.float32), self.mesh_v
        f0, spf = self.frame_of(step)
        x0, x1 = self._sway_offset(f0 * self.dt), self._sway_offset((f0 + spf) * self.dt)
        v = np.zeros_like(self.mesh_vertices)
        v[:, 0] = np.float32((x1 - x0) / (spf * self.dt))
        base = self.mesh_vertices.copy()
        base[:, 0] += np.float32(x0)
        return (base + np.float32(self.dt * (step - f0)) * v).astype(np.float32), v

    def joints_at(self, step: int):
        """(joint_verts_v, joint_faces_v) at substep `step`: the scene's constant arrays, or the swaying body's velocity."""
        if self.joint_verts_v is None or self.mesh_sway is None:
            return self.joint_verts_v, self.joint_faces_v
        _, v = self.body_at(step)
        vx = v[0]
        return (np.tile(vx[None], (self.joint_verts_v.shape[0], 1)).astype(np.float32),
                np.tile(vx[None], (np.asarray(self.joint_faces_v).reshape(-1, 3).shape[0], 1)).astype(np.float32))


def _cloth_scene(name, verts, faces, n_grid, **kw) -> Scene:
    """Assemble a cloth-only scene the way train_material_params.py:375-397 does."""
    init_dir, rest_dir, e_vol, v_vol = garment.compute_dir_vol(verts, faces, thickness=1e-5)
    R_inv = garment.compute_rest_dir_inv(rest_dir)
    elts = verts[faces].mean(1).astype(np.float32)
    x = np.concatenate([elts, verts], 0).astype(np.float32)
    vol = np.concatenate([e_vol, v_vol], 0).astype(np.float32)
    params = {"material": "cloth", "g": [0.0, -9.8, 0.0], "density": 1.0, "grid_v_damping_scale": 1.1,
              "friction_angle": 40.0}
    params.update(kw.pop("params", {}))
    return Scene(name=name, n_grid=n_grid, grid_lim=2.0, n_elements=faces.shape[0], n_traditional=0,
                 n_vertices=verts.shape[0], x=x, v=np.zeros_like(x), vol=vol, faces=faces.astype(np.int32),
                 d=init_dir, R_inv=R_inv, params=params, **kw)


def _trad_scene(name, pts, vol, n_grid, material="jelly", v=None, **kw) -> Scene:
    n = pts.shape[0]
    params = {"material": material, "g": [0.0, -9.8, 0.0], "density": 1.0, "grid_v_damping_scale": 1.1}
    params.update(kw.pop("params", {}))
    return Scene(name=name, n_grid=n_grid, grid_lim=2.0, n_elements=0, n_traditional=n, n_vertices=0,
                 x=pts.astype(np.float32), v=(np.zeros_like(pts) if v is None else v).astype(np.float32),
                 vol=np.full(n, vol, np.float32), faces=np.zeros((0, 3), np.int32), d=np.zeros((0, 3, 3), np.float32),
                 R_inv=np.zeros((0, 3), np.float32), params=params, **kw)


def _lattice(n, spacing, corner, jitter, seed):
    idx = np.stack(np.meshgrid(np.arange(n[0]), np.arange(n[1]), np.arange(n[2]), indexing="ij"), -1).reshape(-1, 3)
    pts = np.asarray(corner, np.float64) + (idx + 0.5) * spacing
    rng = np.random.default_rng(seed)
    pts = pts + rng.uniform(-jitter, jitter, pts.shape) * spacing
    return pts.astype(np.float32)


# ---------------------------------------------------------------------------- BASELINE configs
def cube(n=20, n_grid=64, spacing=0.025, corner=(0.75, 1.0, 0.75), material="jelly", E=100.0, n_steps=100,
         seed=0, name=None, **kw) -> Scene:
    """S1 'cube-8k' (BASELINE config 1): n^3 traditional particles, spinning about z, bounding box."""
    pts = _lattice((n, n, n), spacing, corner, 0.1, seed)
    c = pts.mean(0)
    omega = np.array([0.0, 0.0, 2.0], np.float32)
    v = np.cross(omega, pts - c).astype(np.float32)
    return _trad_scene(name or f"cube-{n**3}", pts, spacing ** 3, n_grid, material=material, v=v, E=E,
                       bcs=[("bounding_box", {})], n_steps=n_steps, **kw)


def block(n=80, n_grid=256, spacing=None, n_steps=200, seed=2, material="jelly") -> Scene:
    """S4t: n^3 traditional jelly particles (pure scatter/gather test), ~8 particles per cell."""
    dx = 2.0 / n_grid
    spacing = spacing or dx / 2.0
    ext = n * spacing
    pts = _lattice((n, n, n), spacing, (1.0 - ext / 2, 1.0 - ext / 2, 1.0 - ext / 2), 0.1, seed)
    return _trad_scene(f"block-{n**3}", pts, spacing ** 3, n_grid, material=material,
                       bcs=[("bounding_box", {})], n_steps=n_steps)


def garment_cylinder(n_theta=200, n_h=200, n_grid=128, aniso=True, collider_subdiv=5, n_steps=1000,
                     joint_rows=2, name=None, sway=None) -> Scene:
    """S2/S3 stand-ins for the Actor01 garment (BASELINE configs 2 and 3).

    aniso=False: every mesh point is a traditional fixed-corotated ('jelly') particle (S2).
    aniso=True : elements + vertices with the anisotropic cloth model, a capsule body collider
                 (friction 0.5) moving sideways, and the top ``joint_rows`` rows attached to the body
                 through the particle mover (S3).  sway = (amplitude, frequency, substeps per frame): the body
                 velocity is amplitude * sin(2 pi f t) along x (per-frame poses); None: uniform 0.3 m/s.
    """
    verts, faces = garment.cylinder(n_theta, n_h, 0.25, 0.8, (1.0, 1.0, 1.0))
    if not aniso:
        init_dir, rest_dir, e_vol, v_vol = garment.compute_dir_vol(verts, faces, thickness=1e-5)
        pts = np.concatenate([verts[faces].mean(1), verts], 0)
        # membrane volumes are ~1e-11; run the isotropic stand-in with a volumetric particle size instead
        spacing = 0.8 / n_h
        sc = _trad_scene(name or f"garment-{pts.shape[0]}-iso", pts, spacing ** 3 / 4, n_grid, material="jelly",
                         bcs=[("bounding_box", {})], n_steps=n_steps)
        return sc
    mv, mf = garment.capsule(collider_subdiv, 0.2, 0.3, (1.0, 1.0, 1.0))
    mesh_v = np.tile(np.array([[0.3, 0.0, 0.0]], np.float32), (mv.shape[0], 1))
    njv = joint_rows * n_theta
    njf = 2 * n_theta * (joint_rows - 1)
    jv = np.tile(np.array([[0.3, 0.0, 0.0]], np.float32), (njv, 1))
    jf = jv[faces[:njf]].mean(1) if njf else np.zeros((0, 3), np.float32)
    return _cloth_scene(name or f"garment-{faces.shape[0] + verts.shape[0]}-aniso", verts, faces, n_grid,
                        mesh_vertices=mv, mesh_faces=mf, mesh_v=mesh_v, mesh_friction=0.5, num_joint_v=njv,
                        num_joint_f=njf, joint_verts_v=jv, joint_faces_v=jf.astype(np.float32),
                        bcs=[("bounding_box", {})], n_steps=n_steps, mesh_sway=sway)


def sheet(n=408, n_grid=256, collider_subdiv=5, n_steps=1000, seed=1, name=None, y=1.2, span=(0.2, 1.8),
          sphere_r=0.3, sphere_c=(1.0, 0.9, 1.0)) -> Scene:
    """S4 'sheet-500k' (BASELINE config 4, the headline metric's workload): n x n vertex sheet in the x-z
    plane (n=408 -> 166,464 vertices + 331,298 elements = 497,762 particles) above a static sphere
    collider (20,480 faces), bounding box, seeded 1e-4 height jitter."""
    verts, faces = garment.grid_sheet(n, n, span[0], span[1], span[0], span[1], y)
    rng = np.random.default_rng(seed)
    verts[:, 1] += rng.uniform(-1e-4, 1e-4, verts.shape[0]).astype(np.float32)
    mv, mf = garment.icosphere(collider_subdiv, sphere_r, sphere_c)
    return _cloth_scene(name or f"sheet-{faces.shape[0] + verts.shape[0]}", verts, faces, n_grid, mesh_vertices=mv,
                        mesh_faces=mf, mesh_v=np.zeros_like(mv), mesh_friction=0.5, bcs=[("bounding_box", {})],
                        n_steps=n_steps)


def sheet_stack(layers, n=408, n_grid=256, dy=0.08, collider_subdiv=5, n_steps=1000, seed=1) -> Scene:
    """Weak-scaling workload for the slab decomposition: `layers` copies of the S4 sheet (497,762 particles each) stacked dy
    apart above the same sphere, in the same 256^3 grid.  x-slabs cut every layer, so N ranks on N layers own one sheet's worth
    of particles each, whatever N (bench.py reports it beside the strong-scaling headline for N > 1)."""
    vs, fs = [], []
    rng = np.random.default_rng(seed)
    for k in range(layers):
        v, f = garment.grid_sheet(n, n, 0.2, 1.8, 0.2, 1.8, 1.2 + dy * k)
        v[:, 1] += rng.uniform(-1e-4, 1e-4, v.shape[0]).astype(np.float32)
        fs.append(f + sum(a.shape[0] for a in vs))
        vs.append(v)
    verts, faces = np.concatenate(vs, 0), np.concatenate(fs, 0)
    mv, mf = garment.icosphere(collider_subdiv, 0.3, (1.0, 0.9, 1.0))
    return _cloth_scene(f"sheet-{faces.shape[0] + verts.shape[0]}-x{layers}", verts, faces, n_grid, mesh_vertices=mv,
                        mesh_faces=mf, mesh_v=np.zeros_like(mv), mesh_friction=0.5, bcs=[("bounding_box", {})],
                        n_steps=n_steps)


def demo_mix(n_grid=64, n_sheet=24, sand=(24, 4, 12), n_steps=200, seed=3, hold=None) -> Scene:
    """Reduced stand-in of the run_demo.py scene: cloth sheet + sand block (material 2)
    + floor plane (sticky surface collider) + body mesh collider + particle mover that pins the first sheet row and
    holds the sand, releasing it in stages (run_demo.py:309-315,377-379,524).  hold = (start, every, rate) in
    substeps / particles; default: everything held for 40 substeps, then 1/8 of the sand released every 10;
    hold=False: the sand is free from the start."""
    verts, faces = garment.grid_sheet(n_sheet, n_sheet, 0.7, 1.3, 0.7, 1.3, 1.25)
    init_dir, rest_dir, e_vol, v_vol = garment.compute_dir_vol(verts, faces, thickness=1e-5)
    R_inv = garment.compute_rest_dir_inv(rest_dir)
    pts, t_vols = garment.get_sand(center=(0.75, 1.45, 0.875), length=(0.5, 0.04, 0.25), res=sand, noise=0.002,
                                   rng=np.random.default_rng(seed))  # utils/demo_utils.py:6-24
    t_vol = float(t_vols[0])
    elts = verts[faces].mean(1)
    x = np.concatenate([elts, pts, verts], 0).astype(np.float32)
    vol = np.concatenate([e_vol, np.full(pts.shape[0], t_vol, np.float32), v_vol], 0).astype(np.float32)
    mv, mf = garment.icosphere(3, 0.25, (1.0, 0.9, 1.0))
    njv = n_sheet  # first lattice row is 'attached'
    jv = np.zeros((njv, 3), np.float32)
    params = {"material": "sand", "g": [0.0, -9.8, 0.0], "density": 1.0, "grid_v_damping_scale": 1.1,
              "friction_angle": 40.0}
    start, every, rate = hold if hold else (40, 10, max(pts.shape[0] // 8, 1))
    return Scene(name="demo-mix", joint_t_hold=0 if hold is False else pts.shape[0], joint_t_start=start, joint_t_every=every, joint_t_rate=rate,
                 n_grid=n_grid, grid_lim=2.0, n_elements=faces.shape[0], n_traditional=pts.shape[0],
                 n_vertices=verts.shape[0], x=x, v=np.zeros_like(x), vol=vol, faces=faces.astype(np.int32), d=init_dir,
                 R_inv=R_inv, params=params, mesh_vertices=mv, mesh_faces=mf, mesh_v=np.zeros_like(mv),
                 mesh_friction=0.5, num_joint_v=njv, num_joint_f=0, joint_verts_v=jv,
                 joint_faces_v=np.zeros((0, 3), np.float32),
                 bcs=[("surface_collider", {"point": [0.0, 0.1, 0.0], "normal": [0.0, 1.0, 0.0]})], n_steps=n_steps)


# ---------------------------------------------------------------------------- reduced sizes for tests
def small_cube(n=8, n_grid=32, material="jelly", **kw):
    return cube(n=n, n_grid=n_grid, spacing=0.03, corner=(0.85, 1.0, 0.85), material=material,
                name=f"cube-{n**3}-{material}", **kw)


def small_sheet(n=24, n_grid=32, n_steps=200):
    """32x32-class sheet over a sphere, sized for the serial oracle."""
    return sheet(n=n, n_grid=n_grid, collider_subdiv=3, n_steps=n_steps, span=(0.6, 1.4), y=1.22,
                 sphere_r=0.2, sphere_c=(1.0, 0.98, 1.0), name=f"sheet-{n}x{n}")


def small_garment(n_theta=32, n_h=24, n_grid=48, n_steps=200):
    return garment_cylinder(n_theta=n_theta, n_h=n_h, n_grid=n_grid, aniso=True, collider_subdiv=2, n_steps=n_steps,
                            name=f"garment-{n_theta}x{n_h}")


REGISTRY = {
    "cube-8k": lambda: cube(),
    "garment-120k-iso": lambda: garment_cylinder(aniso=False),
    # S3: the body sways with 0.5 sin(2 pi t) m/s, posed every 400 substeps like the reference's frames
    "garment-120k-aniso": lambda: garment_cylinder(aniso=True, sway=(0.5, 1.0, 400)),
    "sheet-500k": lambda: sheet(),
    "block-512k": lambda: block(),
    "demo-mix": lambda: demo_mix(),
    # run_demo.py-sized stand-in: 250^3 grid, 100,000 sand particles (utils/demo_utils.py:6), 200x200 garment sheet
    # (sand held for 100 substeps, then 1000 particles released every 4 substeps: the reference's per-frame schedule
    # compressed so that a few hundred timed substeps see held, mixed and free sand)
    "demo-250": lambda: demo_mix(n_grid=250, n_sheet=200, sand=(200, 10, 50), n_steps=400, hold=(100, 4, 1000)),
}
