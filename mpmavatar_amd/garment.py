"""Caller-side particle construction for garment meshes (NumPy).

Restates the maths the reference's physics driver performs before it hands tensors to the solver
(SURVEY.md 8(f) N1): ``compute_dir_vol`` (/root/reference/train_material_params.py:533-553),
``compute_rest_dir_inv`` (:508-515), ``compute_rest_dir_inv_from_vf`` (:517-531) and the world->sim
normalisation (:365-373).  The solver itself never calls this module; the synthetic scenes, the
bench and the tests do.
"""
from __future__ import annotations

import numpy as np


def compute_dir_vol(vertices: np.ndarray, faces: np.ndarray, thickness: float = 1e-5):
    """init_dir [n_e,3,3] (columns d1,d2,d3), rest_dir [n_e,3] (R11,R12,R22), element_vol, vertex_vol.

    train_material_params.py:533-553: d1=v1-v0, d2=v2-v0, d3=unit normal; rest_dir is the upper
    triangle of the QR factor of [d1 d2]; element_vol = 0.25*thickness*area, vertex_vol = sum of the
    volumes of incident elements.
    """
    vertices = np.asarray(vertices, np.float32)
    faces = np.asarray(faces, np.int64)
    d1 = vertices[faces[:, 1]] - vertices[faces[:, 0]]
    d2 = vertices[faces[:, 2]] - vertices[faces[:, 0]]
    cr = np.cross(d1, d2)
    d3 = cr / np.linalg.norm(cr, axis=1, keepdims=True)
    init_dir = np.stack([d1, d2, d3], -1).astype(np.float32)
    R11 = np.linalg.norm(d1, axis=1)
    R12 = (d1 * d2).sum(1) / R11
    R22 = np.linalg.norm(d2 - (R12 / R11)[:, None] * d1, axis=1)
    rest_dir = np.stack([R11, R12, R22], -1).astype(np.float32)
    area = 0.5 * np.linalg.norm(cr, axis=1)
    element_vol = (0.25 * thickness * area).astype(np.float32)
    vertex_vol = np.zeros(vertices.shape[0], np.float32)
    np.add.at(vertex_vol, faces.reshape(-1), np.repeat(element_vol, 3))
    return init_dir, rest_dir, element_vol, vertex_vol


def compute_rest_dir_inv(rest_dir: np.ndarray) -> np.ndarray:
    """(iR11, iR12, iR22) of the inverse upper-triangular 2x2; train_material_params.py:508-515."""
    R11, R12, R22 = rest_dir[:, 0], rest_dir[:, 1], rest_dir[:, 2]
    iR11 = 1.0 / R11
    iR22 = 1.0 / R22
    iR12 = -R12 * iR11 * iR22
    return np.stack([iR11, iR12, iR22], -1).astype(np.float32)


def compute_rest_dir_inv_from_vf(vertices: np.ndarray, faces: np.ndarray) -> np.ndarray:
    """train_material_params.py:517-531 (used with the H-scaled rest pose, :587)."""
    vertices = np.asarray(vertices, np.float32)
    d1 = vertices[faces[:, 1]] - vertices[faces[:, 0]]
    d2 = vertices[faces[:, 2]] - vertices[faces[:, 0]]
    R11 = np.linalg.norm(d1, axis=1)
    R12 = (d1 * d2).sum(1) / R11
    R22 = np.linalg.norm(d2 - (R12 / R11)[:, None] * d1, axis=1)
    return compute_rest_dir_inv(np.stack([R11, R12, R22], -1))


def world_to_sim(verts: np.ndarray):
    """scale/shift so that the garment's bounding box is centred at (1,1,1) with unit max extent.

    train_material_params.py:365-373.  Returns (scale, shift).
    """
    mn, mx = verts.min(0), verts.max(0)
    scale = 1.0 / float((mx - mn).max())
    shift = np.array([1.0, 1.0, 1.0], np.float32) - (mn + mx) / 2.0 * scale
    return scale, shift.astype(np.float32)


def get_sand(center=(-0.4, 1.8, -0.1), length=(0.8, 0.04, 0.2), res=(200, 10, 50), noise=0.01, rng=None):
    """Sand block of run_demo.py (utils/demo_utils.py:6-24): a res[0] x res[1] x res[2] lattice spanning `length` from
    `center` (its min corner), enumerated with res[0] (x) fastest, then res[2] (z), then res[1] (y) -- the order matters,
    run_demo.py:524 releases the sand by slicing this list -- plus Gaussian jitter; volume = box volume / count each.
    Returns (points [n,3] float32, volumes [n] float32)."""
    res = np.asarray(res, np.int64)
    idx = np.stack(np.meshgrid(np.arange(res[1]), np.arange(res[2]), np.arange(res[0]), indexing="ij"), -1)
    pts = idx.reshape(-1, 3).astype(np.float32)[:, [2, 0, 1]]
    pts = pts / np.array([res[0] - 1, res[1] - 1, res[2] - 1], np.float32)
    pts = pts * np.asarray(length, np.float32) + np.asarray(center, np.float32)
    if noise:
        rng = rng or np.random.default_rng(0)
        pts = pts + rng.normal(size=pts.shape).astype(np.float32) * np.float32(noise)
    n = int(res.prod())
    vol = np.full(n, (length[0] * length[1] * length[2]) / n, np.float32)
    return pts.astype(np.float32), vol


# ------------------------------------------------------------------ synthetic surface meshes
def grid_sheet(nx: int, nz: int, x0: float, x1: float, z0: float, z1: float, y: float):
    """nx x nz vertex lattice in the x-z plane, two triangles per quad. Vertex id = ix*nz + iz."""
    xs = np.linspace(x0, x1, nx, dtype=np.float64)
    zs = np.linspace(z0, z1, nz, dtype=np.float64)
    X, Z = np.meshgrid(xs, zs, indexing="ij")
    verts = np.stack([X, np.full_like(X, y), Z], -1).reshape(-1, 3).astype(np.float32)
    ix, iz = np.meshgrid(np.arange(nx - 1), np.arange(nz - 1), indexing="ij")
    a = (ix * nz + iz).reshape(-1)
    b, c, d = a + nz, a + 1, a + nz + 1
    faces = np.concatenate([np.stack([a, c, b], -1), np.stack([c, d, b], -1)], 0).astype(np.int32)
    # interleave the two triangles of each quad so that face order follows the lattice
    faces = faces.reshape(2, -1, 3).transpose(1, 0, 2).reshape(-1, 3)
    return verts, np.ascontiguousarray(faces)


def cylinder(n_theta: int, n_h: int, radius: float, height: float, center):
    """Open cylinder around the y axis, periodic in theta; rows ordered top -> bottom so that the first
    rows (the 'waistband') come first, as the reference's joint-first ordering requires
    (preprocess/split_garments.py:72-92).  Vertex id = row*n_theta + t.  Faces of row band r are
    [2*n_theta*r, 2*n_theta*(r+1))."""
    th = np.arange(n_theta) * (2.0 * np.pi / n_theta)
    ys = center[1] + height / 2.0 - np.arange(n_h) * (height / (n_h - 1))
    T, Y = np.meshgrid(th, ys, indexing="xy")  # [n_h, n_theta]
    verts = np.stack([center[0] + radius * np.cos(T), Y, center[2] + radius * np.sin(T)], -1)
    verts = verts.reshape(-1, 3).astype(np.float32)
    r, t = np.meshgrid(np.arange(n_h - 1), np.arange(n_theta), indexing="ij")
    a = (r * n_theta + t).reshape(-1)
    b = (r * n_theta + (t + 1) % n_theta).reshape(-1)
    c, d = a + n_theta, b + n_theta
    faces = np.stack([np.stack([a, b, c], -1), np.stack([b, d, c], -1)], 1).reshape(-1, 3).astype(np.int32)
    return verts, np.ascontiguousarray(faces)


def icosphere(subdiv: int, radius: float, center):
    """Icosphere with 20*4**subdiv faces (subdiv=5 -> 20,480 faces / 10,242 vertices, the SMPL-X-sized
    collider of SURVEY.md 8(d) S4)."""
    t = (1.0 + 5.0 ** 0.5) / 2.0
    v = [(-1, t, 0), (1, t, 0), (-1, -t, 0), (1, -t, 0), (0, -1, t), (0, 1, t), (0, -1, -t), (0, 1, -t),
         (t, 0, -1), (t, 0, 1), (-t, 0, -1), (-t, 0, 1)]
    f = [(0, 11, 5), (0, 5, 1), (0, 1, 7), (0, 7, 10), (0, 10, 11), (1, 5, 9), (5, 11, 4), (11, 10, 2), (10, 7, 6),
         (7, 1, 8), (3, 9, 4), (3, 4, 2), (3, 2, 6), (3, 6, 8), (3, 8, 9), (4, 9, 5), (2, 4, 11), (6, 2, 10), (8, 6, 7),
         (9, 8, 1)]
    verts = [np.array(p, np.float64) / np.linalg.norm(p) for p in v]
    faces = [tuple(x) for x in f]
    for _ in range(subdiv):
        cache, nf = {}, []

        def mid(a, b):
            key = (a, b) if a < b else (b, a)
            if key not in cache:
                m = verts[a] + verts[b]
                verts.append(m / np.linalg.norm(m))
                cache[key] = len(verts) - 1
            return cache[key]

        for a, b, c in faces:
            ab, bc, ca = mid(a, b), mid(b, c), mid(c, a)
            nf += [(a, ab, ca), (b, bc, ab), (c, ca, bc), (ab, bc, ca)]
        faces = nf
    V = np.asarray(verts) * radius + np.asarray(center, np.float64)
    return V.astype(np.float32), np.asarray(faces, np.int32)


def capsule(subdiv: int, radius: float, half_height: float, center):
    """Icosphere whose upper/lower hemispheres are shifted by +-half_height along y."""
    V, F = icosphere(subdiv, radius, (0.0, 0.0, 0.0))
    V = V.copy()
    V[:, 1] += np.where(V[:, 1] >= 0, half_height, -half_height).astype(np.float32)
    return (V + np.asarray(center, np.float32)).astype(np.float32), F
