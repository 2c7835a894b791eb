"""``warp_mpm.warp_utils.from_torch_safe`` (warp_utils.py:12)."""
from mpmavatar_amd.warp_mpm.warp_utils import from_torch_safe, to_torch  # noqa: F401
