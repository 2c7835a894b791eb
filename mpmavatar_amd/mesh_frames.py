"""Per-face frames and bound Gaussians on the GPU: the step right after the solver in the reference's render loop
(SURVEY.md 8(f) N3).  Mirrors the relevant surface of the reference's ``MeshGaussianModel`` / ``GaussianModel``
(/root/reference/scene/mesh_gaussian_model.py:137-146, scene/gaussian_model.py:112-151) over two HIP kernels
(``mpmhip_face_frames``, ``mpmhip_bind_gaussians``): simulated vertices stay on the device,

    frames = MeshFrames(faces)                       # faces: [n_f, 3] int tensor (cloth + body mesh)
    frames.set_mesh_by_verts(sim2wld(state.particle_x[n_e + n_t:]))
    xyz, rot, scale = frames.get_xyz(binding, _xyz), frames.get_rotation(binding, _rotation), frames.get_scaling(binding, _scaling)

and feed the rasteriser.  No CPU fallback: tensors must live on an MI355X.
"""
from __future__ import annotations

import torch

from . import _lib as L


def _chk(t, dtype, name, shape_last=None):
    if not (isinstance(t, torch.Tensor) and t.is_cuda and t.dtype == dtype and t.is_contiguous()):
        raise RuntimeError(f"{name}: expected a contiguous {dtype} tensor on the GPU")
    if shape_last is not None and (t.dim() < 1 or t.shape[-1] != shape_last):
        raise RuntimeError(f"{name}: last dimension must be {shape_last}")
    return t


class MeshFrames:
    def __init__(self, faces: torch.Tensor):
        self.faces = _chk(faces.to(torch.int32).contiguous(), torch.int32, "faces", 3)
        self._lib = L.load()
        self.verts = None
        self.face_center = self.face_orien_mat = self.face_orien_quat = self.face_scaling = None

    def _call(self, name, *args):
        rc = getattr(self._lib, name)(*args)
        if rc != L.OK:
            raise L.MPMHipError(rc, f"{name} failed")

    # mesh_gaussian_model.py:137-146
    def set_mesh_by_verts(self, verts: torch.Tensor):
        v = _chk(verts, torch.float32, "verts", 3)
        if v.device != self.faces.device:
            raise RuntimeError("verts and faces must be on the same device")
        n_f = self.faces.shape[0]
        dev = v.device
        new = lambda *s: torch.empty(*s, dtype=torch.float32, device=dev)
        self.verts = v
        self.face_center, self.face_orien_mat = new(n_f, 3), new(n_f, 3, 3)
        self.face_orien_quat, self.face_scaling = new(n_f, 4), new(n_f, 1)
        self._call("mpmhip_face_frames", dev.index or 0, torch.cuda.current_stream(dev).cuda_stream, v.data_ptr(),
                   self.faces.data_ptr(), n_f, self.face_center.data_ptr(), self.face_orien_mat.data_ptr(),
                   self.face_orien_quat.data_ptr(), self.face_scaling.data_ptr())

    def _bind(self, binding, xyz_local=None, rotation=None, scaling=None):
        if self.face_center is None:
            raise RuntimeError("call set_mesh_by_verts first")
        b = _chk(binding.to(torch.int32).contiguous(), torch.int32, "binding")
        n = b.numel()
        dev = b.device
        out = [None, None, None]
        ptr = lambda t: None if t is None else t.data_ptr()
        if xyz_local is not None:
            _chk(xyz_local, torch.float32, "_xyz", 3)
            out[0] = torch.empty(n, 3, dtype=torch.float32, device=dev)
        if rotation is not None:
            _chk(rotation, torch.float32, "_rotation", 4)
            out[1] = torch.empty(n, 4, dtype=torch.float32, device=dev)
        if scaling is not None:
            _chk(scaling, torch.float32, "_scaling", 3)
            out[2] = torch.empty(n, 3, dtype=torch.float32, device=dev)
        self._call("mpmhip_bind_gaussians", dev.index or 0, torch.cuda.current_stream(dev).cuda_stream, n, b.data_ptr(),
                   ptr(xyz_local), ptr(rotation), ptr(scaling), self.face_center.data_ptr(), self.face_orien_mat.data_ptr(),
                   self.face_orien_quat.data_ptr(), self.face_scaling.data_ptr(), ptr(out[0]), ptr(out[1]), ptr(out[2]))
        return out

    # gaussian_model.py:141-151 / :124-138 / :112-122 (binding is not None branch)
    def get_xyz(self, binding, xyz_local):
        return self._bind(binding, xyz_local=xyz_local)[0]

    def get_rotation(self, binding, rotation):
        return self._bind(binding, rotation=rotation)[1]

    def get_scaling(self, binding, scaling):
        return self._bind(binding, scaling=scaling)[2]

    def get_all(self, binding, xyz_local, rotation, scaling):
        """All three in one launch."""
        return tuple(self._bind(binding, xyz_local, rotation, scaling))
