#!/usr/bin/env python
"""The reference's eval / demo loop (train_material_params.py:742-857, run_demo.py:514-545) on synthetic inputs, end to
end on one MI355X, minus the two third-party stages (Blender AO bake, diff_gauss rasteriser):

    per frame:  num_substeps x p2g2p (one fused call, mesh advected on the device)
                -> read back the cloth vertices -> uvmesh/NNN.obj (+ sand/NNN.obj)
                -> per-face frames -> world-space parameters of the Gaussians bound to the faces (on the device)

    python examples/synthetic_demo.py --out /tmp/demo --frames 5 --substeps 400 [--scene demo-mix|garment]
"""
import argparse
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mpmavatar_amd import harness, io_formats, scenes  # noqa: E402
from mpmavatar_amd.mesh_frames import MeshFrames  # noqa: E402


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default="/tmp/mpmhip_demo")
    ap.add_argument("--frames", type=int, default=5)
    ap.add_argument("--substeps", type=int, default=400)
    ap.add_argument("--scene", default="demo-mix", choices=["demo-mix", "garment"])
    ap.add_argument("--gaussians-per-face", type=int, default=4)
    a = ap.parse_args(argv)
    sc = scenes.demo_mix() if a.scene == "demo-mix" else scenes.small_garment()
    sim = harness.build_solver(sc, "cuda:0")
    dev = sim.solver.device
    ne, nt = sc.n_elements, sc.n_traditional
    faces = torch.as_tensor(sc.faces, device=dev)
    # a UV template with one 'vt' per face corner (the real one comes with the dataset)
    os.makedirs(a.out, exist_ok=True)
    uv = os.path.join(a.out, "uv_template.obj")
    with open(uv, "w") as f:
        f.writelines(f"vt {(i % 97) / 97.0} {(i % 89) / 89.0}\n" for i in range(3 * sc.n_elements))
        f.writelines(f"f {v[0] + 1}/{3 * i + 1} {v[1] + 1}/{3 * i + 2} {v[2] + 1}/{3 * i + 3}\n" for i, v in enumerate(sc.faces))
    writer = io_formats.UVMeshWriter(uv, sc.faces)
    frames = MeshFrames(faces)
    n_g = a.gaussians_per_face * sc.n_elements
    g = torch.Generator(device=dev).manual_seed(0)
    binding = torch.arange(sc.n_elements, device=dev, dtype=torch.int32).repeat_interleave(a.gaussians_per_face)
    _xyz = 0.3 * torch.randn(n_g, 3, device=dev, generator=g)
    _rot = torch.randn(n_g, 4, device=dev, generator=g)
    _scl = -3.0 + 0.3 * torch.randn(n_g, 3, device=dev, generator=g)
    t_sim = t_io = 0.0
    for frame in range(a.frames):
        t0 = time.perf_counter()
        harness.run(sim, a.substeps, fused=True)
        pos = sim.state.particle_x.clone()                    # run_demo.py:532 (wp.to_torch(particle_x).clone())
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        verts = pos[ne + nt:].contiguous()
        frames.set_mesh_by_verts(verts)
        xyz, rot, scale = frames.get_all(binding, _xyz, _rot, _scl)
        writer.write(os.path.join(a.out, "uvmesh"), frame + 1, verts.cpu().numpy())
        if nt:
            io_formats.write_points_obj(os.path.join(a.out, "sand"), frame + 1, pos[ne:ne + nt].cpu().numpy())
        t2 = time.perf_counter()
        t_sim += t1 - t0
        t_io += t2 - t1
        assert torch.isfinite(xyz).all() and torch.isfinite(rot).all() and torch.isfinite(scale).all()
    print(f"{a.frames} frames x {a.substeps} substeps of {sc.name}: simulation {1e3 * t_sim / a.frames:.1f} ms/frame "
          f"({a.frames * a.substeps / t_sim:.0f} substeps/s), frames + OBJ output {1e3 * t_io / a.frames:.1f} ms/frame; "
          f"{n_g} Gaussians bound to {sc.n_elements} faces; files under {a.out}")
    return dict(xyz=xyz, rot=rot, scale=scale, verts=verts)


if __name__ == "__main__":
    main()
