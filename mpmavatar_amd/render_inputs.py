"""What the rasteriser is handed: from device-resident simulated vertices to the argument lists of the reference's
render call (/root/reference/gaussian_renderer/__init__.py:52-103) without leaving the GPU.

The reference writes every simulated frame to an OBJ file, has Blender bake ambient occlusion, reads the meshes back and only then
poses the Gaussians and renders (train_material_params.py:819-845, run_demo.py:540-604).  The geometry half of that detour is not
needed: ``MeshFrames.set_mesh_by_verts`` takes the solver's ``particle_x`` as it lies in HBM, and ``BoundGaussians.render_inputs``
returns exactly ``means3D, means2D, opacities, scales, rotations`` (+ the caller's ``shs`` / ``colors_precomp`` and the ``extra``
primitives concatenated behind, :84-91) in one launch (``mpmhip_render_inputs``).  The rasteriser (diff_gauss) and the AO bake stay
out of scope; the OBJ files are still what Blender needs (``io_formats.write_uv_obj``), but nothing has to be read back from them.

    frames = MeshFrames(faces);  frames.set_mesh_by_verts(sim2wld(state.particle_x[n_e + n_t:]))
    args = gaussians.render_inputs(frames, override_color=colors, extra=(xyz, colors, opacity, scales, rotations))
    rasterizer(**args)          # keys: means3D means2D shs colors_precomp opacities scales rotations cov3Ds_precomp
"""
from __future__ import annotations

import torch

from . import _lib as L
from .mesh_frames import MeshFrames, _chk


class BoundGaussians:
    """The parameters of a mesh-bound GaussianModel that the render call reads (scene/gaussian_model.py:45-73): raw ``_xyz`` (face-local),
    ``_rotation``, ``_scaling``, ``_opacity``, ``_features_dc`` / ``_features_rest`` and ``binding`` (Gaussian -> face)."""

    def __init__(self, xyz, rotation, scaling, opacity, features_dc, features_rest, binding):
        self._xyz, self._rotation, self._scaling = _chk(xyz, torch.float32, "_xyz", 3), _chk(rotation, torch.float32, "_rotation", 4), \
            _chk(scaling, torch.float32, "_scaling", 3)
        self._opacity = _chk(opacity, torch.float32, "_opacity", 1)
        self._features_dc, self._features_rest = features_dc, features_rest
        self.binding = _chk(binding.to(torch.int32).contiguous(), torch.int32, "binding")
        n = self.binding.numel()
        if not (self._xyz.shape[0] == self._rotation.shape[0] == self._scaling.shape[0] == self._opacity.shape[0] == n):
            raise RuntimeError("BoundGaussians: parameter tensors must have one row per binding entry")
        self._lib = L.load()

    @property
    def get_features(self):  # gaussian_model.py:153-157
        return torch.cat((self._features_dc, self._features_rest), dim=1)

    def render_inputs(self, frames: MeshFrames, override_color=None, extra=None):
        """-> dict of the rasteriser's keyword arguments (gaussian_renderer/__init__.py:94-102).  extra = (xyz [m,3], colors [m,3],
        opacity [m,1], scales [m,3], rotations [m,4]) as run_demo.py builds it (needs override_color, like the reference)."""
        if frames.face_center is None:
            raise RuntimeError("call frames.set_mesh_by_verts first")
        dev = self.binding.device
        n, m = self.binding.numel(), 0
        ex = [None] * 5
        if extra is not None:
            if override_color is None:
                raise RuntimeError("extra primitives carry precomputed colours: pass override_color for the bound Gaussians too")
            ex = [_chk(t.contiguous(), torch.float32, f"extra[{i}]", w) for i, (t, w) in enumerate(zip(extra, (3, 3, 1, 3, 4)))]
            m = ex[0].shape[0]
        new = lambda w: torch.empty(n + m, w, dtype=torch.float32, device=dev)
        means3D, means2D, opac, scales, rots = new(3), new(3), new(1), new(3), new(4)
        p = lambda t: None if t is None else t.data_ptr()
        rc = self._lib.mpmhip_render_inputs(dev.index or 0, torch.cuda.current_stream(dev).cuda_stream, n, m, self.binding.data_ptr(),
                                            self._xyz.data_ptr(), self._rotation.data_ptr(), self._scaling.data_ptr(),
                                            self._opacity.data_ptr(), frames.face_center.data_ptr(), frames.face_orien_mat.data_ptr(),
                                            frames.face_orien_quat.data_ptr(), frames.face_scaling.data_ptr(), p(ex[0]), p(ex[2]),
                                            p(ex[3]), p(ex[4]), means3D.data_ptr(), means2D.data_ptr(), opac.data_ptr(),
                                            scales.data_ptr(), rots.data_ptr())
        if rc != L.OK:
            raise L.MPMHipError(rc, "mpmhip_render_inputs failed")
        shs = colors = None
        if override_color is None:
            shs = self.get_features                      # SH -> RGB in the rasteriser (:81)
        else:
            colors = override_color if extra is None else torch.cat([override_color, ex[1]])   # :91
        return {"means3D": means3D, "means2D": means2D, "shs": shs, "colors_precomp": colors, "opacities": opac, "scales": scales,
                "rotations": rots, "cov3Ds_precomp": None}
