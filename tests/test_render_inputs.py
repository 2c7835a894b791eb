"""the tensors handed to the rasteriser.  Pinned by tests/golden/render_inputs.npz, which
tests/golden/make_golden_render.py produced with the REFERENCE'S OWN GaussianModel getters (scene/gaussian_model.py:112-161, imported
unchanged; plyfile / simple_knn / roma stubbed) and its get_extra_attr (utils/demo_utils.py:59-85).  CPU: the oracle
(oracle/face_frames.py) against the fixture.  GPU: mpmhip_render_inputs through mpmavatar_amd.render_inputs against the fixture, and
the property that makes the per-frame OBJ round trip of train_material_params.py:819-845 unnecessary for the geometry."""
import os

import numpy as np
import pytest

from oracle import face_frames as ff

GOLD = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "render_inputs.npz"))
KEYS = ("means3D", "rotations", "scales", "opacities")
TOL = {"means3D": 2e-6, "rotations": 5e-6, "scales": 2e-6, "opacities": 2e-7}


def _extra():
    return tuple(GOLD["extra_" + k] for k in ("xyz", "colors", "opacity", "scales", "rotations"))


def _close(a, b, tol):
    return bool((np.abs(a - b) <= tol * np.maximum(np.abs(b), 1.0)).all())


def test_oracle_reproduces_the_reference_getters():
    o = ff.MeshFramesOracle(GOLD["faces"])
    o.set_mesh_by_verts(GOLD["verts"])
    got = o.render_inputs(GOLD["binding"], GOLD["_xyz"], GOLD["_rotation"], GOLD["_scaling"], GOLD["_opacity"])
    for k in KEYS:
        assert got[k].shape == GOLD[k].shape and _close(got[k], GOLD[k], TOL[k]), k
    gx = o.render_inputs(GOLD["binding"], GOLD["_xyz"], GOLD["_rotation"], GOLD["_scaling"], GOLD["_opacity"], extra=_extra())
    for k in KEYS:
        assert gx[k].shape == GOLD["x_" + k].shape and _close(gx[k], GOLD["x_" + k], TOL[k]), k
    assert (gx["means2D"] == 0).all() and gx["means2D"].shape == gx["means3D"].shape


@pytest.mark.gpu
def test_hip_render_inputs_match_the_reference_getters():
    import torch
    from mpmavatar_amd.mesh_frames import MeshFrames
    from mpmavatar_amd.render_inputs import BoundGaussians
    dev = torch.device("cuda:0")
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    fr = MeshFrames(t(GOLD["faces"]))
    fr.set_mesh_by_verts(t(GOLD["verts"]))
    g = BoundGaussians(t(GOLD["_xyz"]), t(GOLD["_rotation"]), t(GOLD["_scaling"]), t(GOLD["_opacity"]), t(GOLD["_features_dc"]),
                       t(GOLD["_features_rest"]), t(GOLD["binding"]))
    a = g.render_inputs(fr)
    for k in KEYS:
        assert _close(a[k].cpu().numpy(), GOLD[k], TOL[k]), k
    assert np.array_equal(a["shs"].cpu().numpy(), GOLD["shs"]) and a["colors_precomp"] is None and a["cov3Ds_precomp"] is None
    assert (a["means2D"] == 0).all()
    b = g.render_inputs(fr, override_color=t(GOLD["override_color"]), extra=tuple(t(x) for x in _extra()))
    for k in KEYS:
        assert _close(b[k].cpu().numpy(), GOLD["x_" + k], TOL[k]), k
    assert np.array_equal(b["colors_precomp"].cpu().numpy(), GOLD["x_colors_precomp"]) and b["shs"] is None
    with pytest.raises(RuntimeError):
        g.render_inputs(fr, extra=tuple(t(x) for x in _extra()))        # extra without override_color: like the reference, no colours to cat


@pytest.mark.gpu
def test_the_obj_round_trip_is_not_needed_for_the_geometry(tmp_path):
    """train_material_params.py:819-845 writes each simulated frame as `v x y z` lines, and the render pass reads the files back
    before it poses the Gaussians.  Frames and rasteriser inputs taken from the solver's particle_x on the device equal the ones
    taken from the re-read file to the precision Python's float repr round-trips (exactly)."""
    import torch
    from mpmavatar_amd import harness, io_formats, scenes
    from mpmavatar_amd.mesh_frames import MeshFrames
    from mpmavatar_amd.render_inputs import BoundGaussians
    sc = scenes.small_garment()
    sim = harness.build_solver(sc, "cuda:0", mode="fast")
    harness.run(sim, 30, fused=True)
    verts = sim.state.particle_x[sc.n_elements + sc.n_traditional:].contiguous()      # stays in HBM
    faces = torch.as_tensor(sc.faces, device=verts.device)
    rng = np.random.default_rng(3)
    n = 2 * sc.n_elements
    t = lambda a: torch.from_numpy(a.astype(np.float32)).to(verts.device)
    g = BoundGaussians(t(rng.normal(0, 0.3, (n, 3))), t(rng.normal(size=(n, 4))), t(rng.normal(-1, 0.5, (n, 3))), t(rng.normal(size=(n, 1))),
                       t(rng.normal(size=(n, 1, 3))), t(rng.normal(size=(n, 3, 3))), torch.from_numpy(rng.integers(0, sc.n_elements, n)).to(verts.device))
    fr = MeshFrames(faces)
    fr.set_mesh_by_verts(verts)
    direct = g.render_inputs(fr)
    path = tmp_path / "001.obj"                     # the reference's detour
    with open(path, "w") as f:
        f.writelines([f"v {v[0]} {v[1]} {v[2]}\n" for v in verts.detach().cpu().numpy()])
    back, _ = io_formats.read_obj(str(path))
    fr2 = MeshFrames(faces)
    fr2.set_mesh_by_verts(torch.from_numpy(np.asarray(back, np.float32)).to(verts.device))
    via_file = g.render_inputs(fr2)
    for k in KEYS:
        assert torch.equal(direct[k], via_file[k]), k
