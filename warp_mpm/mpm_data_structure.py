"""``from warp_mpm.mpm_data_structure import MPMStateStruct, MPMModelStruct`` (run_demo.py:28-31)."""
from mpmavatar_amd.warp_mpm.mpm_data_structure import MPMModelStruct, MPMSmallStateStruct, MPMStateStruct  # noqa: F401
