"""On-disk hand-off formats: OBJ frames, split_idx.npz, parameter checkpoints."""
import numpy as np
import pytest

from mpmavatar_amd import garment, io_formats as io


def test_uvmesh_obj_round_trip_and_text(tmp_path):
    verts, faces = garment.grid_sheet(5, 4, 0.0, 1.0, 0.0, 1.0, 0.5)
    verts = (verts + np.random.default_rng(0).normal(0, 0.1, verts.shape)).astype(np.float32)
    uv = tmp_path / "uv.obj"
    with open(uv, "w") as f:
        f.write("# template\n")
        f.writelines(f"vt {0.1 * i} {0.2 * i}\n" for i in range(faces.shape[0] + 2))
        f.writelines(f"f {a + 1}/{i + 1} {b + 1}/{i + 2} {c + 1}/{i + 3}\n" for i, (a, b, c) in enumerate(faces))
    w = io.UVMeshWriter(str(uv), faces)
    path = w.write(str(tmp_path / "uvmesh"), 7, verts)
    assert path.endswith("007.obj")
    text = open(path).read().splitlines()
    v0 = verts[0]
    assert text[0] == f"v {v0[0]} {v0[1]} {v0[2]}"                      # the reference's f-string on float32 scalars
    a, b, c = faces[0] + 1
    assert f"f {a}/1 {b}/2 {c}/3" in text
    rv, rf = io.read_obj(path)
    assert rv.dtype == np.float32 and np.array_equal(rv, verts)         # shortest repr round-trips float32 exactly
    assert rf.dtype == np.int32 and np.array_equal(rf, faces)
    with pytest.raises(ValueError):
        io.UVMeshWriter(str(uv), faces[:-1])
    p = io.write_points_obj(str(tmp_path / "sand"), 0, verts[:5])
    assert np.array_equal(io.read_obj(p)[0], verts[:5]) and io.read_obj(p)[1].size == 0


def test_split_idx_and_params(tmp_path):
    s = io.SplitIdx(3, 2, np.arange(10), np.arange(12), np.arange(10, 30), np.arange(12, 40),
                    np.array([[0, 1, 2], [2, 3, 9]]), np.array([[0, 1, 2]]))
    fn = str(tmp_path / "split_idx.npz")
    s.save(fn)
    t = io.load_split_idx(fn)
    assert t.num_joint_v == 3 and t.num_joint_f == 2 and np.array_equal(t.new_cloth_faces, s.new_cloth_faces)
    np.savez(str(tmp_path / "bad.npz"), num_joint_v=1)
    with pytest.raises(KeyError):
        io.load_split_idx(str(tmp_path / "bad.npz"))
    s.num_joint_v = 11
    s.save(fn)
    with pytest.raises(ValueError):
        io.load_split_idx(fn)
    best = dict(D=np.float32(1.2), E=np.float32(150.0), H=np.float32(0.9), loss=0.03, step=17)
    io.save_params(str(tmp_path), 17, best, dict(best, loss=0.05))
    p = io.load_params(str(tmp_path / "best_param_00017.npz"))
    assert p["loss"] == 1.0 and p["step"] == -1 and float(p["E"]) == 150.0
    assert float(np.load(str(tmp_path / "last_param_00017.npz"))["loss"]) == 0.05


@pytest.mark.gpu
def test_synthetic_demo_end_to_end(tmp_path):
    """Solver -> read-back -> uvmesh OBJ -> face frames -> bound Gaussians (examples/synthetic_demo.py), small."""
    import importlib.util
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("synthetic_demo", os.path.join(root, "examples", "synthetic_demo.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    out = mod.main(["--out", str(tmp_path), "--frames", "2", "--substeps", "30", "--scene", "garment"])
    v, f = io.read_obj(str(tmp_path / "uvmesh" / "002.obj"))
    assert np.array_equal(v, out["verts"].cpu().numpy()) and f.shape[1] == 3
