"""Drop-in replacement of the reference's ``warp_mpm`` package over libmpmhip.so (MI355X / gfx950).

Same module, class and method names as /root/reference/warp_mpm/{mpm_solver,mpm_data_structure,warp_utils}.py;
the top-level ``warp_mpm`` package of this repository re-exports it under the reference's import paths.
"""
from .mpm_data_structure import MPMModelStruct, MPMSmallStateStruct, MPMStateStruct  # noqa: F401
from .mpm_solver import MPMWARP  # noqa: F401
from .warp_utils import from_torch_safe, to_torch  # noqa: F401
from . import wp_facade  # noqa: F401

MPMSolver = MPMWARP  # BASELINE.json's prose name for the same class
