"""Golden vectors for the hand-off into the rasteriser FROM THE REFERENCE'S OWN GETTERS.

scene/gaussian_model.py imports three third-party modules that are not in this image (plyfile, simple_knn, roma).  Two are not
touched by the getters; roma supplies three quaternion helpers.  They are replaced by stand-ins in ``sys.modules`` -- plyfile and
simple_knn empty, roma's ``quat_product`` / ``quat_xyzw_to_wxyz`` / ``quat_wxyz_to_xyzw`` restated (Hamilton product in XYZW) -- and
then the reference's ``GaussianModel`` is imported UNCHANGED from /root/reference and its properties ``get_xyz``, ``get_rotation``,
``get_scaling``, ``get_opacity``, ``get_features`` (scene/gaussian_model.py:112-161) are evaluated on seeded inputs.  The lists the
render call concatenates (gaussian_renderer/__init__.py:84-91) are formed from them with the reference's ``get_extra_attr``
(utils/demo_utils.py:59-85; its device="cuda" literals patched to the CPU).  Data only: inputs and expected outputs.
    python tests/golden/make_golden_render.py        (build container: needs /root/reference)"""
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, "/root/reference")


def _stub(name, **attrs):
    m = types.ModuleType(name)
    for k, v in attrs.items():
        setattr(m, k, v)
    sys.modules[name] = m
    return m


def quat_product(p, q):  # roma.quat_product: Hamilton product, XYZW
    px, py, pz, pw = p.unbind(-1)
    qx, qy, qz, qw = q.unbind(-1)
    return torch.stack([pw * qx + px * qw + py * qz - pz * qy, pw * qy - px * qz + py * qw + pz * qx,
                        pw * qz + px * qy - py * qx + pz * qw, pw * qw - px * qx - py * qy - pz * qz], -1)


_stub("plyfile", PlyData=object, PlyElement=object)
_stub("simple_knn")
_stub("simple_knn._C", distCUDA2=None)
_stub("roma", quat_product=quat_product, quat_xyzw_to_wxyz=lambda q: torch.cat([q[..., 3:], q[..., :3]], -1),
      quat_wxyz_to_xyzw=lambda q: torch.cat([q[..., 1:], q[..., :1]], -1))

import importlib.util  # noqa: E402
_spec = importlib.util.spec_from_file_location("ref_gaussian_model", "/root/reference/scene/gaussian_model.py")  # (not the package:
_mod = importlib.util.module_from_spec(_spec)       # scene/__init__.py pulls the dataset readers and their dependencies in)
_spec.loader.exec_module(_mod)
GaussianModel = _mod.GaussianModel  # the reference's class
from utils.graphics_utils import compute_face_orientation  # noqa: E402
from mpmavatar_amd import garment  # noqa: E402
from oracle import face_frames as ff  # noqa: E402  (only for rotmat -> quaternion, pinned against SciPy in tests/test_frames.py)


def main():
    rng = np.random.default_rng(23)
    verts, faces = garment.grid_sheet(12, 10, 0.6, 1.4, 0.7, 1.3, 1.2)
    verts = (verts + np.stack([0 * verts[:, 0], 0.07 * np.sin(6 * verts[:, 0]) * np.cos(4 * verts[:, 2]), 0 * verts[:, 0]], 1)
             + rng.normal(0, 2e-3, verts.shape)).astype(np.float32)
    n_f = faces.shape[0]
    n = 2 * n_f + 17
    binding = rng.integers(0, n_f, n).astype(np.int64)
    sh_degree = 2
    inp = {"_xyz": rng.normal(0, 0.4, (n, 3)), "_rotation": rng.normal(size=(n, 4)), "_scaling": rng.normal(-1.0, 0.6, (n, 3)),
           "_opacity": rng.normal(0, 2, (n, 1)), "_features_dc": rng.normal(size=(n, 1, 3)),
           "_features_rest": rng.normal(size=(n, (sh_degree + 1) ** 2 - 1, 3))}
    inp = {k: v.astype(np.float32) for k, v in inp.items()}
    pc = GaussianModel(sh_degree)
    for k, v in inp.items():
        setattr(pc, k, torch.from_numpy(v))
    pc.binding = torch.from_numpy(binding)
    tv, tf = torch.from_numpy(verts), torch.from_numpy(faces.astype(np.int64))
    pc.face_center = tv[tf].mean(dim=-2)                                          # mesh_gaussian_model.py:137-146
    pc.face_orien_mat, pc.face_scaling = compute_face_orientation(tv, tf, return_scale=True)
    pc.face_orien_quat = torch.from_numpy(ff.xyzw_to_wxyz(ff.rotmat_to_unitquat_xyzw(pc.face_orien_mat.numpy())))
    out = {"means3D": pc.get_xyz, "rotations": pc.get_rotation, "scales": pc.get_scaling, "opacities": pc.get_opacity,
           "shs": pc.get_features}
    # the `extra` primitives of run_demo.py (sand + chair), through the reference's get_extra_attr on the CPU
    du_file = "/root/reference/utils/demo_utils.py"   # (read, not imported: the module pulls the camera / dataset packages in)
    src = open(du_file).read().replace('device="cuda"', 'device="cpu"')
    ns = {}
    exec(compile(src[src.index("def get_extra_attr"):src.index("def prune_faces")], du_file, "exec"), {"torch": torch}, ns)
    n_s, n_c = 40, 25
    sand = torch.from_numpy(rng.uniform(0.5, 1.5, (n_s, 3)).astype(np.float32))
    chair = {"xyz": torch.from_numpy(rng.uniform(0, 2, (n_c, 3)).astype(np.float32)), "opacity": torch.rand(n_c, 1, generator=torch.Generator().manual_seed(1)),
             "rotation": torch.nn.functional.normalize(torch.randn(n_c, 4, generator=torch.Generator().manual_seed(2))),
             "scale": torch.rand(n_c, 3, generator=torch.Generator().manual_seed(3)) * 0.01}
    chair_color = torch.rand(n_c, 3, generator=torch.Generator().manual_seed(4))
    extra_attr, _, sand_color = ns["get_extra_attr"](chair, chair_color, sand)
    colors = torch.rand(n, 3, generator=torch.Generator().manual_seed(5))          # override_color of the render call
    ex = {"x_means3D": torch.cat([out["means3D"], extra_attr[0]]), "x_opacities": torch.cat([out["opacities"], extra_attr[2]]),
          "x_scales": torch.cat([out["scales"], extra_attr[3]]), "x_rotations": torch.cat([out["rotations"], extra_attr[4]]),
          "x_colors_precomp": torch.cat([colors, extra_attr[1]])}                  # gaussian_renderer/__init__.py:84-91
    np.savez_compressed(os.path.join(HERE, "render_inputs.npz"), verts=verts, faces=faces.astype(np.int32), binding=binding.astype(np.int32),
                        override_color=colors.numpy(), **inp, **{k: v.detach().numpy() for k, v in out.items()},
                        **{k: v.detach().numpy() for k, v in ex.items()},
                        **{f"extra_{k}": v.numpy() for k, v in zip(("xyz", "colors", "opacity", "scales", "rotations"), extra_attr)})
    print("render_inputs.npz:", n, "bound Gaussians,", n_s + n_c, "extra")


if __name__ == "__main__":
    main()
