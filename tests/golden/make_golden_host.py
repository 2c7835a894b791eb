"""Golden vectors for the caller-side helpers FROM THE REFERENCE ITSELF.

The helpers are plain PyTorch / Python, but they live in modules whose imports (warp, diff_gauss, plyfile ...) are not
installed here, so each function is cut out of its file by name with `ast` and executed on its own with the real torch
/ numpy (Tensor.cuda is made a no-op: there is no GPU in the build container).  The fixture holds inputs and expected
outputs only.  Run here (needs /root/reference):  python tests/golden/make_golden_host.py
  * Trainer.compute_dir_vol / compute_rest_dir_inv / compute_rest_dir_inv_from_vf  (train_material_params.py:508-553)
  * get_sand (utils/demo_utils.py:6-24), noise = 0 (its jitter comes from torch's global RNG)
  * read_obj (utils/general_utils.py:318-334) on OBJ files written by mpmavatar_amd.io_formats
"""
import ast
import os
import sys
import tempfile

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
sys.path.insert(0, ROOT)
from mpmavatar_amd import garment, io_formats  # noqa: E402


def cut(path, names):
    """{name: function} for the named (possibly nested-in-class) function definitions of a reference file."""
    src = open(os.path.join(REF, path)).read()
    found = {}
    for node in ast.walk(ast.parse(src)):
        if isinstance(node, ast.FunctionDef) and node.name in names and node.name not in found:
            ns = {"torch": torch, "np": np}
            exec(compile(ast.Module(body=[node], type_ignores=[]), path, "exec"), ns)
            found[node.name] = ns[node.name]
    missing = set(names) - set(found)
    assert not missing, missing
    return found


def main():
    torch.Tensor.cuda = lambda self, *a, **k: self
    rng = np.random.default_rng(23)
    out = {}
    # ---- N1: garment particle construction
    f = cut("train_material_params.py", ["compute_dir_vol", "compute_rest_dir_inv", "compute_rest_dir_inv_from_vf"])
    verts, faces = garment.cylinder(14, 9, 0.25, 0.8, (1.0, 1.0, 1.0))
    verts = (verts + rng.normal(0, 4e-3, verts.shape)).astype(np.float32)
    tv, tf = torch.from_numpy(verts), torch.from_numpy(faces.astype(np.int64))
    init_dir, rest_dir, e_vol, v_vol = f["compute_dir_vol"](None, tv.clone(), tf, 1e-5)
    out.update(n1_verts=verts, n1_faces=faces.astype(np.int32), n1_init_dir=init_dir.numpy(), n1_rest_dir=rest_dir.numpy(),
               n1_element_vol=e_vol.numpy(), n1_vertex_vol=v_vol.numpy(),
               n1_rest_dir_inv=f["compute_rest_dir_inv"](None, rest_dir).numpy(),
               n1_rest_dir_inv_vf=f["compute_rest_dir_inv_from_vf"](None, tv * torch.tensor([[1.0, 0.9, 1.0]]), tf).numpy())
    # ---- N2: the sand block
    g = cut("utils/demo_utils.py", ["get_sand"])
    for tag, kw in (("default", dict(center=[-0.4, 1.8, -0.1], length=[0.8, 0.04, 0.2], res=[20, 4, 7])),
                    ("other", dict(center=[0.75, 1.45, 0.875], length=[0.5, 0.04, 0.25], res=[5, 3, 2]))):
        pts, vol = g["get_sand"](noise=0.0, **kw)
        out.update({f"n2_{tag}_center": np.array(kw["center"], np.float32), f"n2_{tag}_length": np.array(kw["length"], np.float32),
                    f"n2_{tag}_res": np.array(kw["res"], np.int32), f"n2_{tag}_points": pts.numpy(), f"n2_{tag}_vol": vol.numpy()})
    # ---- N4: OBJ files written here, parsed by the reference's reader
    r = cut("utils/general_utils.py", ["read_obj"])
    sheet_v, sheet_f = garment.grid_sheet(5, 4, 0.1, 0.9, 0.2, 0.8, 1.1)
    sheet_v = (sheet_v + rng.normal(0, 1e-2, sheet_v.shape)).astype(np.float32)
    with tempfile.TemporaryDirectory() as td:
        uv = os.path.join(td, "uv.obj")
        with open(uv, "w") as fh:   # a uv template like data/.../uvmesh: vt lines and v/vt faces
            for v in sheet_v:
                fh.write("v %f %f %f\n" % tuple(v))
            for v in sheet_v:
                fh.write("vt %f %f\n" % (v[0], v[2]))
            for t in sheet_f + 1:
                fh.write("f %d/%d %d/%d %d/%d\n" % (t[0], t[0], t[1], t[1], t[2], t[2]))
        w = io_formats.UVMeshWriter(uv, sheet_f)
        moved = (sheet_v * 1.5 + 0.25).astype(np.float32)
        path = w.write(td, 7, moved)
        rv, rf = r["read_obj"](path)
        out.update(n4_obj_text=np.frombuffer(open(path, "rb").read(), np.uint8), n4_verts_written=moved, n4_ref_verts=rv, n4_ref_faces=rf)
        ppath = io_formats.write_points_obj(td, 3, moved[:6])
        pv, pf = r["read_obj"](ppath)
        out.update(n4_points_text=np.frombuffer(open(ppath, "rb").read(), np.uint8), n4_ref_points=pv, n4_ref_points_faces_n=np.int32(pf.size))
    np.savez_compressed(os.path.join(HERE, "host.npz"), **out)
    print("host.npz:", {k: v.shape for k, v in out.items()})


if __name__ == "__main__":
    main()
