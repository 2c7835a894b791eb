"""Face frames and bound Gaussians.

CPU: oracle/face_frames.py against golden vectors produced by the REFERENCE's own compute_face_orientation
(tests/golden/frames.npz, tests/golden/make_golden_frames.py) -- this row of the oracle is pinned -- and the restated
roma quaternion helpers against SciPy.  GPU: the HIP kernels through the C ABI against the oracle and the fixture."""
import numpy as np
import pytest
from scipy.spatial.transform import Rotation as Rot

from oracle import face_frames as ff

GOLD = np.load(__import__("os").path.join(__import__("os").path.dirname(__file__), "golden", "frames.npz"))


def _inputs(seed=0, n_g=4000):
    rng = np.random.default_rng(seed)
    n_f = GOLD["faces"].shape[0]
    return dict(binding=rng.integers(0, int(GOLD["n_regular"]), n_g).astype(np.int32),
                xyz=rng.normal(0, 0.3, (n_g, 3)).astype(np.float32), rot=rng.normal(0, 1, (n_g, 4)).astype(np.float32),
                scl=rng.normal(-3, 0.5, (n_g, 3)).astype(np.float32)), n_f


def test_oracle_matches_the_reference_function():
    o, s = ff.compute_face_orientation(GOLD["verts"], GOLD["faces"])
    n = int(GOLD["n_regular"])
    assert np.abs(o[:n] - GOLD["orientation"][:n]).max() < 5e-7
    assert np.abs(s[:n] - GOLD["scale"][:n]).max() < 1e-7
    # degenerate / needle triangles: eps-clamped normalisation, same numbers as the reference (incl. zeros)
    assert np.allclose(o[n:], GOLD["orientation"][n:], atol=1e-6) and np.allclose(s[n:], GOLD["scale"][n:], atol=1e-7)
    m = ff.MeshFramesOracle(GOLD["faces"])
    m.set_mesh_by_verts(GOLD["verts"])
    assert np.abs(m.face_center - GOLD["center"]).max() < 2e-7


def test_quaternion_helpers_against_scipy():
    n = int(GOLD["n_regular"])
    R = GOLD["orientation"][:n].astype(np.float64)
    assert np.abs(np.einsum("nij,nkj->nik", R, R) - np.eye(3)).max() < 1e-5 and np.abs(np.linalg.det(R) - 1).max() < 1e-5
    q = ff.rotmat_to_unitquat_xyzw(R)
    qs = Rot.from_matrix(R).as_quat()
    assert np.abs(q - qs).max() < 1e-6               # same branch selection and sign as SciPy
    assert np.abs(Rot.from_quat(q).as_matrix() - R).max() < 1e-5
    rng = np.random.default_rng(1)
    a, b = rng.normal(size=(500, 4)), rng.normal(size=(500, 4))
    a, b = a / np.linalg.norm(a, axis=1, keepdims=True), b / np.linalg.norm(b, axis=1, keepdims=True)
    prod = ff.quat_product_xyzw(a.astype(np.float32), b.astype(np.float32))
    ref = (Rot.from_quat(a) * Rot.from_quat(b)).as_quat()
    assert np.minimum(np.abs(prod - ref).max(1), np.abs(prod + ref).max(1)).max() < 1e-5


@pytest.mark.gpu
def test_hip_frames_and_binding():
    import torch
    from mpmavatar_amd.mesh_frames import MeshFrames
    dev = torch.device("cuda:0")
    verts, faces = torch.from_numpy(GOLD["verts"]).to(dev), torch.from_numpy(GOLD["faces"]).to(dev)
    fr = MeshFrames(faces)
    fr.set_mesh_by_verts(verts)
    n = int(GOLD["n_regular"])
    mat, scl, ctr = fr.face_orien_mat.cpu().numpy(), fr.face_scaling.cpu().numpy(), fr.face_center.cpu().numpy()
    assert np.abs(mat[:n] - GOLD["orientation"][:n]).max() < 2e-6      # vs the reference's outputs
    assert np.abs(scl[:n] - GOLD["scale"][:n]).max() < 1e-6 and np.abs(ctr - GOLD["center"]).max() < 5e-7
    assert np.isfinite(mat).all() and np.allclose(mat[n:], GOLD["orientation"][n:], atol=1e-5)
    o = ff.MeshFramesOracle(GOLD["faces"])
    o.set_mesh_by_verts(GOLD["verts"])
    q = fr.face_orien_quat.cpu().numpy()
    assert np.abs(q[:n] - o.face_orien_quat[:n]).max() < 5e-6           # WXYZ, same sign convention
    assert np.abs(Rot.from_quat(q[:n][:, [1, 2, 3, 0]]).as_matrix() - GOLD["orientation"][:n]).max() < 1e-5
    inp, _ = _inputs()
    t = {k: torch.from_numpy(v).to(dev) for k, v in inp.items()}
    xyz, rot, s3 = fr.get_all(t["binding"], t["xyz"], t["rot"], t["scl"])
    assert np.abs(xyz.cpu().numpy() - o.get_xyz(inp["binding"], inp["xyz"])).max() < 2e-6
    assert np.abs(rot.cpu().numpy() - o.get_rotation(inp["binding"], inp["rot"])).max() < 5e-6
    assert rel_ok(s3.cpu().numpy(), o.get_scaling(inp["binding"], inp["scl"]))
    assert np.abs(fr.get_xyz(t["binding"], t["xyz"]).cpu().numpy() - xyz.cpu().numpy()).max() == 0.0


def rel_ok(a, b, tol=2e-6):
    return bool((np.abs(a - b) <= tol * np.abs(b) + 1e-12).all())


@pytest.mark.gpu
def test_hip_frames_from_simulated_vertices():
    """The intended use: frames of the cloth faces straight from the solver's particle_x (no host copy), sized like the
    reference's garments (~80k faces); size-independent properties: orthonormal right-handed frames, centres inside
    the triangles' bounding boxes, quaternion <-> matrix round trip."""
    import torch
    from mpmavatar_amd import harness, scenes
    from mpmavatar_amd.mesh_frames import MeshFrames
    sc = scenes.garment_cylinder(aniso=True)
    sim = harness.build_solver(sc, "cuda:0", mode="fast")
    harness.run(sim, 20, fused=True)
    verts = sim.state.particle_x[sc.n_elements + sc.n_traditional:].contiguous()
    fr = MeshFrames(torch.as_tensor(sc.faces, device=verts.device))
    fr.set_mesh_by_verts(verts)
    R = fr.face_orien_mat.double()
    eye = torch.eye(3, dtype=torch.float64, device=R.device)
    assert (R.transpose(1, 2) @ R - eye).abs().max() < 1e-5 and (torch.linalg.det(R) - 1).abs().max() < 1e-5
    tri = verts[torch.as_tensor(sc.faces, device=verts.device).long()]
    assert (fr.face_center <= tri.max(1).values + 1e-6).all() and (fr.face_center >= tri.min(1).values - 1e-6).all()
    q = fr.face_orien_quat.cpu().numpy()[:, [1, 2, 3, 0]]
    assert np.abs(Rot.from_quat(q).as_matrix() - R.cpu().numpy()).max() < 1e-5
    assert fr.face_scaling.min() > 0
