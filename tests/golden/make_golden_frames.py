"""Golden vectors for the face-frame step FROM THE REFERENCE ITSELF: compute_face_orientation is
plain PyTorch (utils/graphics_utils.py:88-106) and importable in the build container, so its outputs on seeded inputs
pin oracle/face_frames.py.  Run here (needs /root/reference):  python tests/golden/make_golden_frames.py
The fixture holds inputs and expected outputs only."""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, "/root/reference")
from utils.graphics_utils import compute_face_orientation  # noqa: E402  (the reference's function)
from mpmavatar_amd import garment  # noqa: E402


def main():
    rng = np.random.default_rng(17)
    # a curved garment-like sheet with jitter, a sphere, and a handful of degenerate / needle triangles
    v1, f1 = garment.grid_sheet(24, 18, 0.6, 1.4, 0.7, 1.3, 1.2)
    v1 = v1 + np.stack([0 * v1[:, 0], 0.08 * np.sin(7 * v1[:, 0]) * np.cos(5 * v1[:, 2]), 0 * v1[:, 0]], 1)
    v1 = (v1 + rng.normal(0, 2e-3, v1.shape)).astype(np.float32)
    v2, f2 = garment.icosphere(2, 0.3, (1.0, 0.9, 1.0))
    v3 = np.array([[0, 0, 0], [1, 0, 0], [2, 0, 0], [0, 0, 0], [0, 0, 0], [1e-4, 0, 0], [0.5, 1e-7, 0],
                   [3, 4, 5], [3, 4, 5.000001], [3.1, 4, 5]], np.float32)
    f3 = np.array([[0, 1, 2], [3, 4, 5], [0, 5, 6], [7, 8, 9], [0, 0, 0]], np.int32)
    verts = np.concatenate([v1, v2, v3]).astype(np.float32)
    faces = np.concatenate([f1, f2 + v1.shape[0], f3 + v1.shape[0] + v2.shape[0]]).astype(np.int32)
    ori, scale = compute_face_orientation(torch.from_numpy(verts), torch.from_numpy(faces), return_scale=True)
    center = torch.from_numpy(verts)[torch.from_numpy(faces).long()].mean(dim=-2)   # mesh_gaussian_model.py:139-142
    np.savez_compressed(os.path.join(HERE, "frames.npz"), verts=verts, faces=faces, orientation=ori.numpy(),
                        scale=scale.numpy(), center=center.numpy(), n_regular=np.int32(f1.shape[0] + f2.shape[0]))
    print("frames.npz:", faces.shape[0], "faces")


if __name__ == "__main__":
    main()
