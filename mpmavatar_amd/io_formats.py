"""On-disk hand-off formats of the reference drivers, so that a run on this solver produces and
consumes the same files: per-frame ``uvmesh/NNN.obj`` (train_material_params.py:776-822, run_demo.py:500-545),
``read_obj`` (utils/general_utils.py:318-334), ``split_idx.npz`` (preprocess/split_garments.py:84-94) and the
``best_param_*.npz`` / ``last_param_*.npz`` checkpoints (train_material_params.py:145-148,725-728).  Host-side Python,
like the reference.  The Blender AO bake and the diff_gauss rasteriser that consume these files are out of scope."""
from __future__ import annotations

import os
from dataclasses import dataclass

import numpy as np


def read_obj(filename):
    """'v ' lines -> float32 [n,3]; 'f ' lines -> int32 [m,k], the index before the first '/', 0-based."""
    vertices, indices = [], []
    with open(filename, "r") as f:
        for line in f:
            if line.startswith("v "):
                p = line.strip().split()
                vertices.append([float(p[1]), float(p[2]), float(p[3])])
            elif line.startswith("f "):
                p = line.strip().split()
                indices.append([int(q.split("/")[0]) - 1 for q in p[1:]])
    return np.array(vertices, dtype=np.float32), np.array(indices, dtype=np.int32)


def _v_lines(verts):
    # the reference formats numpy float32 scalars with str() (shortest round-trip repr); keep that byte for byte
    verts = np.asarray(verts, np.float32)
    return [f"v {v[0]} {v[1]} {v[2]}\n" for v in verts]


class UVMeshWriter:
    """uvmesh/NNN.obj: the frame's vertices followed by the UV template's 'vt' lines and 'f v/vt v/vt v/vt' lines built
    from the mesh faces (train_material_params.py:776-783)."""

    def __init__(self, uv_path, faces):
        vt_f, ft = [], []
        with open(uv_path, "r") as f:
            for line in f:
                if line[:2] == "vt":
                    vt_f.append(line)
                elif line[:2] == "f ":
                    p = line.strip().split()
                    ft.append([int(p[1].split("/")[1]), int(p[2].split("/")[1]), int(p[3].split("/")[1])])
        faces1 = np.asarray(faces).astype(np.int64) + 1
        if len(ft) != faces1.shape[0]:
            raise ValueError(f"UV template has {len(ft)} faces, the mesh has {faces1.shape[0]}")
        vt_f += [f"f {v[0]}/{vt[0]} {v[1]}/{vt[1]} {v[2]}/{vt[2]}\n" for v, vt in zip(faces1, ft)]
        self.vt_f = vt_f

    def write(self, directory, frame, verts):
        os.makedirs(directory, exist_ok=True)
        path = os.path.join(directory, f"{frame:03d}.obj")
        with open(path, "w") as f:
            f.writelines(_v_lines(verts))
            f.writelines(self.vt_f)
        return path


def write_points_obj(directory, frame, points):
    """sand/NNN.obj of run_demo.py:506-509: vertices only."""
    os.makedirs(directory, exist_ok=True)
    path = os.path.join(directory, f"{frame:03d}.obj")
    with open(path, "w") as f:
        f.writelines(_v_lines(points))
    return path


@dataclass
class SplitIdx:
    """split_idx.npz: joint-first reordering of the cloth / body parts of the tracked mesh (the first num_joint_v cloth
    vertices and num_joint_f cloth faces are attached to the body: the mover's particles)."""
    num_joint_v: int
    num_joint_f: int
    reordered_cloth_v_idx: np.ndarray
    reordered_cloth_f_idx: np.ndarray
    reordered_human_v_idx: np.ndarray
    reordered_human_f_idx: np.ndarray
    new_cloth_faces: np.ndarray
    new_human_faces: np.ndarray

    def save(self, filename):
        np.savez(filename, **self.__dict__)


def load_split_idx(filename) -> SplitIdx:
    z = np.load(filename)
    need = list(SplitIdx.__dataclass_fields__)
    missing = [k for k in need if k not in z]
    if missing:
        raise KeyError(f"{filename}: missing {missing}")
    s = SplitIdx(int(z["num_joint_v"]), int(z["num_joint_f"]), *(np.asarray(z[k]) for k in need[2:]))
    if not (0 <= s.num_joint_v <= s.reordered_cloth_v_idx.shape[0] and 0 <= s.num_joint_f <= s.new_cloth_faces.shape[0]):
        raise ValueError(f"{filename}: joint counts exceed the cloth part")
    if s.new_cloth_faces.size and s.new_cloth_faces.max() >= s.reordered_cloth_v_idx.shape[0]:
        raise ValueError(f"{filename}: new_cloth_faces index past the cloth vertices")
    return s


def save_params(directory, step, best_params, last_params):
    """best_param_XXXXX.npz / last_param_XXXXX.npz (train_material_params.py:725-728)."""
    os.makedirs(directory, exist_ok=True)
    np.savez(os.path.join(directory, f"best_param_{step:05d}.npz"), **best_params)
    np.savez(os.path.join(directory, f"last_param_{step:05d}.npz"), **last_params)


def load_params(filename):
    """--init_params_path (train_material_params.py:145-148): the stored arrays with loss / step reset."""
    p = {k: v for k, v in np.load(filename).items()}
    p["loss"] = 1.0
    p["step"] = -1
    return p
