"""Caller-side helpers against golden vectors produced by the reference's own functions
(tests/golden/make_golden_host.py cuts them out of train_material_params.py, utils/demo_utils.py and
utils/general_utils.py and runs them with the real torch)."""
import os

import numpy as np

from mpmavatar_amd import garment, io_formats

G = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "host.npz"))


def test_compute_dir_vol_matches_reference():
    init_dir, rest_dir, e_vol, v_vol = garment.compute_dir_vol(G["n1_verts"], G["n1_faces"], thickness=1e-5)
    np.testing.assert_allclose(init_dir, G["n1_init_dir"], rtol=2e-6, atol=1e-7)
    np.testing.assert_allclose(rest_dir, G["n1_rest_dir"], rtol=2e-6, atol=1e-8)
    np.testing.assert_allclose(e_vol, G["n1_element_vol"], rtol=2e-6)
    np.testing.assert_allclose(v_vol, G["n1_vertex_vol"], rtol=5e-6)   # index_add_ order vs np.add.at


def test_rest_dir_inverses_match_reference():
    np.testing.assert_allclose(garment.compute_rest_dir_inv(G["n1_rest_dir"]), G["n1_rest_dir_inv"], rtol=2e-6)
    scaled = (G["n1_verts"] * np.array([[1.0, 0.9, 1.0]], np.float32)).astype(np.float32)   # the H-scaled rest pose, :587
    np.testing.assert_allclose(garment.compute_rest_dir_inv_from_vf(scaled, G["n1_faces"]), G["n1_rest_dir_inv_vf"], rtol=1e-5)


def test_get_sand_matches_reference_order_and_volume():
    for tag in ("default", "other"):
        pts, vol = garment.get_sand(center=G[f"n2_{tag}_center"], length=G[f"n2_{tag}_length"], res=G[f"n2_{tag}_res"], noise=0.0)
        assert pts.shape == G[f"n2_{tag}_points"].shape
        np.testing.assert_allclose(pts, G[f"n2_{tag}_points"], rtol=0, atol=2e-7)   # same enumeration order, not just the same set
        np.testing.assert_allclose(vol, G[f"n2_{tag}_vol"], rtol=1e-6)


def test_obj_files_round_trip_through_the_reference_reader(tmp_path):
    # what this package writes is what the reference's read_obj parsed when the fixture was made ...
    np.testing.assert_array_equal(G["n4_ref_verts"], G["n4_verts_written"])
    # ... this package's reader parses the same bytes to the same arrays ...
    p = tmp_path / "007.obj"
    p.write_bytes(G["n4_obj_text"].tobytes())
    v, f = io_formats.read_obj(str(p))
    np.testing.assert_array_equal(v, G["n4_ref_verts"])
    np.testing.assert_array_equal(f, G["n4_ref_faces"])
    assert v.dtype == G["n4_ref_verts"].dtype and f.dtype == G["n4_ref_faces"].dtype
    # ... and the writer still produces those bytes
    q = tmp_path / "pts"
    path = io_formats.write_points_obj(str(q), 3, G["n4_ref_points"])
    assert open(path, "rb").read() == G["n4_points_text"].tobytes()
    pv, pf = io_formats.read_obj(path)
    np.testing.assert_array_equal(pv, G["n4_ref_points"])
    assert pf.size == int(G["n4_ref_points_faces_n"])
