"""Top-level ``warp_mpm`` package: the reference's import lines resolve to the MI355X solver unchanged.

The reference drivers do (run_demo.py:27-32, train_material_params.py:28-33)

    import warp as wp
    from warp_mpm.mpm_data_structure import (
        MPMStateStruct,
        MPMModelStruct,
    )
    from warp_mpm.mpm_solver import MPMWARP

With this repository on ``sys.path`` the two ``warp_mpm`` imports load ``mpmavatar_amd.warp_mpm`` (ctypes over
``include/mpmhip.h`` / ``libmpmhip.so``).  ``wp`` here is the four-call facade for the callers' own ``wp.*`` lines
(``wp.init()``, ``wp.to_torch(...)``, ``wp.config``): see INTEGRATION.md section 2.
"""
from mpmavatar_amd.warp_mpm import (MPMModelStruct, MPMSmallStateStruct, MPMSolver, MPMStateStruct, MPMWARP,  # noqa: F401
                                    from_torch_safe, to_torch, wp_facade as wp)
