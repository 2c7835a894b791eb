#!/usr/bin/env python
"""Time the two post-solver maps on garment-sized inputs: 79,600 cloth faces, 400k bound Gaussians.
HIP events on torch's current stream (the kernels are launched there)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from mpmavatar_amd import scenes
from mpmavatar_amd.mesh_frames import MeshFrames

dev = torch.device("cuda:0")
sc = scenes.garment_cylinder(aniso=True)
verts = torch.as_tensor(sc.x[sc.n_elements + sc.n_traditional:], device=dev).contiguous()
faces = torch.as_tensor(sc.faces, device=dev)
fr = MeshFrames(faces)
n_g = 400_000
g = torch.Generator(device=dev).manual_seed(0)
binding = torch.randint(0, faces.shape[0], (n_g,), device=dev, generator=g).sort().values.to(torch.int32)
xyz, rot, scl = (torch.randn(n_g, k, device=dev, generator=g) for k in (3, 4, 3))
for _ in range(5):
    fr.set_mesh_by_verts(verts); fr.get_all(binding, xyz, rot, scl)
e = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
n = 50
torch.cuda.synchronize()
e[0].record()
for _ in range(n): fr.set_mesh_by_verts(verts)
e[1].record()
for _ in range(n): fr.get_all(binding, xyz, rot, scl)
e[2].record()
torch.cuda.synchronize()
t_f, t_g = e[0].elapsed_time(e[1]) / n * 1e3, e[1].elapsed_time(e[2]) / n * 1e3
b_f = faces.shape[0] * (12 + 36 + 68)
b_g = n_g * (4 + 40 + 32 + 40)
print(f"face_frames: {faces.shape[0]} faces {t_f:.1f} us/call ({b_f / t_f / 1e3:.0f} GB/s incl. allocation of the 4 outputs); "
      f"bind_gaussians: {n_g} gaussians {t_g:.1f} us/call ({b_g / t_g / 1e3:.0f} GB/s)")
