"""``from warp_mpm.mpm_solver import MPMWARP`` (run_demo.py:32, train_material_params.py:33)."""
from mpmavatar_amd.warp_mpm.mpm_solver import *  # noqa: F401,F403
from mpmavatar_amd.warp_mpm.mpm_solver import MPMWARP  # noqa: F401
from mpmavatar_amd.warp_mpm.mpm_data_structure import MPMModelStruct, MPMSmallStateStruct, MPMStateStruct  # noqa: F401
