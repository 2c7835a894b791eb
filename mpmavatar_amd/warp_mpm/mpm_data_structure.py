"""``MPMStateStruct`` / ``MPMModelStruct`` with the reference's names and signatures
(/root/reference/warp_mpm/mpm_data_structure.py:13-530, 610-733), holding plain torch tensors.

Ownership follows the reference: ``particle_x, v, d, R_inv, C`` alias (clones of) caller
tensors and are *rebound* by ``reset_state`` / ``continue_from_torch``; ``vol, D_inv, faces, cov, flags`` are
copied in; grids are solver-owned and not exposed.  ``requires_grad`` is accepted and ignored everywhere (the
reference solver is forward-only).

The fast HIP back end keeps a cell-sorted SoA copy of the particle state on the device.  Reading a mutable
field (``particle_x`` ...) through this class first asks the attached solver to write its results back into
the tensor -- the equivalent of the reference's zero-copy ``wp.to_torch(mpm_state.particle_x)`` -- and marks
the tensors as possibly caller-modified so the next substep re-imports them.
"""
from __future__ import annotations

import weakref
from typing import Optional, Sequence, Union

import numpy as np
import torch
from torch import Tensor

from .warp_utils import from_torch_safe

_SOLVER_WRITTEN = ("particle_x", "particle_v", "particle_C", "particle_F", "particle_F_trial", "particle_stress",
                   "particle_d", "vertex_force")
_STATE_OTHER = ("particle_cov", "particle_vol", "particle_mass", "particle_density", "particle_R_inv",
                "particle_D_inv", "particle_traditional", "particle_vertices", "particle_elements",
                "particle_selection", "faces")


def _dev(device):
    return torch.device("cuda:0" if device is None else device)


class _Tracked:
    """Attribute container that counts rebinding of its tensor fields (``_version``) and lets an attached
    solver synchronise before solver-written fields are read."""

    _fields: tuple = ()
    _synced: tuple = ()

    def __init__(self):
        object.__setattr__(self, "_t", {})
        object.__setattr__(self, "_version", 0)
        object.__setattr__(self, "_solver", None)

    def __getattr__(self, name):  # only called when normal lookup fails
        t = object.__getattribute__(self, "_t")
        if name in t:
            if name in type(self)._synced:
                s = object.__getattribute__(self, "_solver")
                s = s() if s is not None else None
                if s is not None:
                    s._before_caller_read(self)
            return t[name]
        raise AttributeError(name)

    def __setattr__(self, name, value):
        if name in type(self)._fields:
            self._t[name] = value
            object.__setattr__(self, "_version", self._version + 1)
        else:
            object.__setattr__(self, name, value)

    def _raw(self, name):
        """Field access without the solver synchronisation hook (for the shim's own plumbing)."""
        return self._t[name]

    def _touch(self):
        """An in-place modification of a bound tensor happened."""
        object.__setattr__(self, "_version", self._version + 1)

    def _attach(self, solver):
        object.__setattr__(self, "_solver", weakref.ref(solver))


class MPMStateStruct(_Tracked):
    _fields = _SOLVER_WRITTEN + _STATE_OTHER
    _synced = _SOLVER_WRITTEN

    # mpm_data_structure.py:51-134
    def init(self, n_particles: int, n_elements: int, n_vertices: int, device=None, requires_grad=False) -> None:
        dev = _dev(device)
        n_nv = n_particles - n_vertices
        z = lambda *s, dt=torch.float32: torch.zeros(*s, dtype=dt, device=dev)
        self.n_particles, self.n_elements, self.n_vertices = n_particles, n_elements, n_vertices
        self.device = dev
        self.particle_x = z(n_particles, 3)
        self.particle_v = z(n_particles, 3)
        self.particle_F = z(n_nv, 3, 3)
        self.particle_d = z(n_elements, 3, 3)
        self.particle_cov = z(n_nv * 6)
        self.particle_F_trial = z(n_nv, 3, 3)
        self.particle_stress = z(n_nv, 3, 3)
        self.particle_C = z(n_particles, 3, 3)
        self.particle_vol = z(n_particles)
        self.particle_mass = z(n_particles)
        self.particle_density = z(n_particles)
        self.particle_R_inv = z(n_elements, 3)
        self.particle_D_inv = z(n_elements, 3, 3)
        self.faces = z(n_elements, 3)
        self.vertex_force = z(n_vertices, 3)
        self.particle_traditional = z(n_particles, dt=torch.int32)
        self.particle_vertices = z(n_particles, dt=torch.int32)
        self.particle_elements = z(n_particles, dt=torch.int32)
        self.particle_selection = z(n_particles, dt=torch.int32)
        self.grid_res = 10

    # mpm_data_structure.py:136-156 -- grids are owned by the solver context; only the resolution is recorded
    def init_grid(self, grid_res: int, device=None, requires_grad=False):
        self.grid_res = int(grid_res)

    # mpm_data_structure.py:158-260
    def from_torch(self, tensor_x: Tensor, tensor_volume: Tensor, tensor_D_inv: Tensor, tensor_R_inv: Tensor,
                   tensor_faces: Tensor, particle_traditional, particle_vertices, particle_elements,
                   tensor_cov: Optional[Tensor] = None, tensor_velocity: Optional[Tensor] = None, n_grid: int = 100,
                   grid_lim=1.0, device="cuda:0", requires_grad=True):
        dev = _dev(device)
        assert tensor_x.shape[0] == tensor_volume.shape[0]
        self.init_grid(grid_res=n_grid, device=device, requires_grad=requires_grad)
        f32 = lambda t: t.detach().to(device=dev, dtype=torch.float32).contiguous().clone()
        if tensor_x is not None:
            self.particle_x = from_torch_safe(f32(tensor_x), dtype="vec3")
        if tensor_volume is not None:
            self.particle_vol = f32(tensor_volume)
        if tensor_D_inv is not None:
            self.particle_D_inv = f32(tensor_D_inv)
        if tensor_R_inv is not None:
            self.particle_R_inv = from_torch_safe(f32(tensor_R_inv), dtype="vec3") if tensor_R_inv.numel() else f32(tensor_R_inv).reshape(0, 3)
        if tensor_faces is not None:
            # float-encoded vertex indices, like wp.from_numpy(faces, dtype=wp.vec3) (:211-215, quirk Q8)
            self.faces = f32(tensor_faces).reshape(-1, 3)
        if tensor_cov is not None:
            self.particle_cov = f32(tensor_cov).reshape(-1)
        if tensor_velocity is not None:
            self.particle_v = from_torch_safe(f32(tensor_velocity), dtype="vec3")
        as_i32 = lambda a: torch.as_tensor(np.asarray(a), dtype=torch.int32, device=dev).contiguous()
        self.particle_traditional = as_i32(particle_traditional)
        self.particle_vertices = as_i32(particle_vertices)
        self.particle_elements = as_i32(particle_elements)
        self._check_classes()
        print("Particles initialized from torch data.")
        print("Total particles: ", tensor_x.shape[0])

    def set_particles(self, *args, **kwargs):
        """BASELINE.json's prose name for ``from_torch`` (the reference itself has no such method)."""
        return self.from_torch(*args, **kwargs)

    def _check_classes(self):
        """The kernels' launch ranges assume elements | traditional | vertices index blocks
        (mpm_solver.py:327-332,518-534); the flag arrays must agree with that layout."""
        n_e, n_p, n_v = self.n_elements, self.n_particles, self.n_vertices
        el = self._raw("particle_elements").cpu().numpy()
        tr = self._raw("particle_traditional").cpu().numpy()
        ve = self._raw("particle_vertices").cpu().numpy()
        ok = (el[:n_e] == 1).all() and (el[n_e:] == 0).all() and (ve[n_p - n_v:] == 1).all() and \
             (ve[:n_p - n_v] == 0).all() and (tr[n_e:n_p - n_v] == 1).all() and (tr[:n_e] == 0).all() and \
             (tr[n_p - n_v:] == 0).all()
        if not ok:
            raise RuntimeError("particle_elements/traditional/vertices flags must mark the contiguous index blocks "
                               "[0,n_e) | [n_e,n_p-n_v) | [n_p-n_v,n_p) (the reference kernels' launch ranges assume it)")

    # mpm_data_structure.py:262-374
    def reset_state(self, n_vertices, tensor_x: Tensor, tensor_d: Tensor, tensor_cov: Optional[Tensor] = None,
                    tensor_velocity: Optional[Tensor] = None, tensor_density: Optional[Tensor] = None,
                    selection_mask: Optional[Tensor] = None, tensor_R_inv: Optional[Tensor] = None, device="cuda:0",
                    requires_grad=True):
        dev = _dev(device)
        s = self._solver() if self._solver is not None else None
        if s is not None:
            s._before_caller_write(self)
        f32 = lambda t: t.detach().to(device=dev, dtype=torch.float32).contiguous()
        if tensor_x is not None:
            self.particle_x = from_torch_safe(f32(tensor_x), dtype="vec3")  # aliases the caller's tensor (:283)
        if tensor_d is not None:
            self.particle_d = f32(tensor_d).clone()
        if tensor_R_inv is not None:
            self.particle_R_inv = f32(tensor_R_inv).clone()
        if tensor_cov is not None:
            self.particle_cov = f32(tensor_cov).reshape(-1).clone()
        if tensor_velocity is not None:
            self.particle_v = from_torch_safe(f32(tensor_velocity).clone(), dtype="vec3")
        if tensor_density is not None and selection_mask is not None:
            sel = selection_mask.to(dev).to(torch.int32) == 1  # set_float_vec_to_vec_wmask (:904-911)
            dens = self._raw("particle_density")
            dens[sel] = f32(tensor_density)[sel]
        self._raw("particle_C").zero_()
        eye = torch.eye(3, dtype=torch.float32, device=dev)
        self._raw("particle_F_trial").copy_(eye.expand_as(self._raw("particle_F_trial")))
        self._raw("particle_F").copy_(eye.expand_as(self._raw("particle_F")))
        self._raw("particle_stress").zero_()
        self._raw("vertex_force").zero_()
        self._touch()

    # mpm_data_structure.py:376-419
    def continue_from_torch(self, tensor_x: Tensor, tensor_velocity: Optional[Tensor] = None,
                            tensor_d: Optional[Tensor] = None, tensor_C: Optional[Tensor] = None,
                            tensor_R_inv: Optional[Tensor] = None, device="cuda:0", requires_grad=True):
        dev = _dev(device)
        s = self._solver() if self._solver is not None else None
        if s is not None:
            s._before_caller_write(self)
        f32 = lambda t: t.detach().to(device=dev, dtype=torch.float32).contiguous()
        if tensor_x is not None:
            self.particle_x = from_torch_safe(f32(tensor_x), dtype="vec3")
        if tensor_velocity is not None:
            self.particle_v = from_torch_safe(f32(tensor_velocity).clone(), dtype="vec3")
        if tensor_d is not None:
            self.particle_d = from_torch_safe(f32(tensor_d).clone(), dtype="mat33")
        if tensor_C is not None:
            self.particle_C = from_torch_safe(f32(tensor_C).clone(), dtype="mat33")
        if tensor_R_inv is not None:
            self.particle_R_inv = f32(tensor_R_inv).clone()
        self._touch()

    def set_require_grad(self, requires_grad=True):  # :421-432, no-op: forward-only solver
        return None

    # mpm_data_structure.py:434-467
    def reset_density(self, tensor_density: Tensor, selection_mask: Optional[Tensor] = None, device="cuda:0",
                      requires_grad=True, update_mass=False):
        dev = _dev(device)
        s = self._solver() if self._solver is not None else None
        if s is not None:
            s._before_caller_write(self)
        n = tensor_density.shape[0]
        self._raw("particle_density")[:n].copy_(tensor_density.detach().to(device=dev, dtype=torch.float32))
        if update_mass:
            torch.mul(self._raw("particle_density"), self._raw("particle_vol"), out=self._raw("particle_mass"))
        self._touch()

    # mpm_data_structure.py:469-486
    def reset_rest_dir(self, tensor_R_inv: Tensor, device="cuda:0"):
        dev = _dev(device)
        s = self._solver() if self._solver is not None else None
        if s is not None:
            s._before_caller_write(self)
        n = tensor_R_inv.shape[0]
        self._raw("particle_R_inv")[:n].copy_(tensor_R_inv.detach().to(device=dev, dtype=torch.float32))
        self._touch()

    # mpm_data_structure.py:488-521
    def partial_clone(self, device="cuda:0", requires_grad=True):
        new_state = MPMStateStruct()
        new_state.init(self.n_particles, self.n_elements, self.n_vertices, device=device, requires_grad=requires_grad)
        for name in ("particle_vol", "particle_density", "particle_mass", "particle_selection"):
            new_state._raw(name).copy_(self._raw(name))
        new_state.init_grid(grid_res=self.grid_res, device=device, requires_grad=requires_grad)
        return new_state

    # mpm_data_structure.py:523-530
    def to_small_state(self, device="cuda:0", requires_grad=True):
        small_state = MPMSmallStateStruct()
        small_state.init(self.n_particles, self.n_elements, self.n_vertices, device=device, requires_grad=requires_grad)
        small_state.set_require_grad(requires_grad=requires_grad)
        return small_state


class MPMSmallStateStruct:
    """mpm_data_structure.py:533-607: the reduced state (x, v, C, d) the reference's differentiable kernels write into.
    Nothing on the substep path reads it (those kernels are dead in both drivers); kept so that
    ``MPMStateStruct.to_small_state`` / ``to_large_state`` callers keep working."""

    def init(self, n_particles: int, n_elements: int, n_vertices: int, device=None, requires_grad=False) -> None:
        dev = _dev(device)
        z = lambda *shape: torch.zeros(*shape, dtype=torch.float32, device=dev)
        self.particle_x, self.particle_v = z(n_particles, 3), z(n_particles, 3)
        self.particle_C, self.particle_d = z(n_particles, 3, 3), z(n_elements, 3, 3)

    def set_require_grad(self, requires_grad=True):  # :567-572, forward-only solver
        return None

    def to_large_state(self, mpm_state: "MPMStateStruct", device="cuda:0", requires_grad=True):
        """:574-607 (the reference builds the state and drops it: no return statement; returned here).  copy_state
        (:982-993): x, v, C for every particle, d and R_inv for the elements."""
        new_state = mpm_state.partial_clone(device=device, requires_grad=requires_grad)
        for name in ("particle_D_inv", "faces"):
            new_state._raw(name).copy_(mpm_state._raw(name))
        n_e = self.particle_d.shape[0]
        new_state._raw("particle_x").copy_(self.particle_x)
        new_state._raw("particle_v").copy_(self.particle_v)
        new_state._raw("particle_C").copy_(self.particle_C.reshape(new_state._raw("particle_C").shape))
        if n_e:
            new_state._raw("particle_d").copy_(self.particle_d.reshape(new_state._raw("particle_d").shape))
            new_state._raw("particle_R_inv").copy_(mpm_state._raw("particle_R_inv"))
        new_state._raw("vertex_force").zero_()
        return new_state


class MPMModelStruct(_Tracked):
    _fields = ("E", "nu", "mu", "lam", "gamma", "kappa", "yield_stress")
    _synced = ("mu", "lam", "yield_stress")  # plastic materials 1/5 update these in the stress kernel
    _scalars = ("material", "friction_angle", "friction_coeff", "alpha", "gravitational_accelaration", "hardening",
                "xi", "plastic_viscosity", "softening", "rpic_damping", "grid_v_damping_scale", "update_cov_with_F")

    def __init__(self):
        super().__init__()
        object.__setattr__(self, "_scalar_version", 0)
        # fields the reference leaves uninitialised until set_parameters_dict (zero-initialised wp.struct)
        self.hardening = 0.0
        self.xi = 0.0

    def __setattr__(self, name, value):
        if name in MPMModelStruct._scalars:
            object.__setattr__(self, "_scalar_version", getattr(self, "_scalar_version", 0) + 1)
        super().__setattr__(name, value)

    # mpm_data_structure.py:647-676
    def init(self, shape: Union[Sequence[int], int], device=None, requires_grad=False) -> None:
        dev = _dev(device)
        for name in ("E", "nu", "mu", "lam", "gamma", "kappa", "yield_stress"):
            setattr(self, name, torch.zeros(shape, dtype=torch.float32, device=dev))
        self.device = dev

    # mpm_data_structure.py:678-684, 870-879
    def finalize_mu_lam(self, n_particles, device="cuda:0"):
        s = self._solver() if self._solver is not None else None
        if s is not None:
            s._before_caller_write(self)  # plastic materials update mu / lam / yield_stress: flush those first
        E, nu = self._raw("E"), self._raw("nu")
        self._raw("mu").copy_(E / (2.0 * (1.0 + nu)))
        self._raw("lam").copy_(E * nu / ((1.0 + nu) * (1.0 - 2.0 * nu)))
        self._touch()

    # mpm_data_structure.py:686-715
    def init_other_params(self, n_grid=100, grid_lim=1.0, device="cuda:0"):
        import math
        self.grid_lim = grid_lim
        self.n_grid = n_grid
        self.grid_dim_x = self.grid_dim_y = self.grid_dim_z = n_grid
        self.dx, self.inv_dx = self.grid_lim / self.n_grid, float(n_grid / grid_lim)
        self.update_cov_with_F = False
        self.material = 0
        self.plastic_viscosity = 0.0
        self.softening = 0.1
        self.friction_angle = 0.0
        sin_phi = math.sin(self.friction_angle / 180.0 * 3.14159265)
        self.friction_coeff = math.tan(self.friction_angle / 180.0 * 3.14159265)
        self.alpha = math.sqrt(2.0 / 3.0) * 2.0 * sin_phi / (3.0 - sin_phi)
        self.gravitational_accelaration = (0.0, 0.0, 0.0)
        self.rpic_damping = 0.0
        self.grid_v_damping_scale = 1.1

    # mpm_data_structure.py:717-725
    def from_torch(self, tensor_E: Tensor, tensor_nu: Tensor, tensor_gamma: Tensor, tensor_kappa: Tensor,
                   device="cuda:0", requires_grad=False):
        self.E = from_torch_safe(tensor_E.contiguous())
        self.nu = from_torch_safe(tensor_nu.contiguous())
        self.gamma = from_torch_safe(tensor_gamma.contiguous())
        self.kappa = from_torch_safe(tensor_kappa.contiguous())
        self.finalize_mu_lam(n_particles=tensor_E.shape[0], device=device)

    def set_require_grad(self, requires_grad=True):  # :727-733, no-op
        return None
