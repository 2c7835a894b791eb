"""warp.types: the names warp_mpm/{warp_utils,mpm_data_structure}.py reach for (TEST INFRASTRUCTURE, see warp/__init__.py)."""
from . import array, float32, int32, uint64, vec2, vec3, vec4, quat, mat22, mat33, mat44  # noqa: F401
