"""Optional: ``import warp as wp`` for an UNEDITED reference driver on a machine without NVIDIA Warp.

Put ``mpmavatar_amd/compat`` on PYTHONPATH.  This is the four-name facade of ``mpmavatar_amd/warp_mpm/wp_facade.py`` (the
``wp.*`` calls the drivers themselves make), nothing more: no kernels, no arrays, no emulation of Warp.
"""
from mpmavatar_amd.warp_mpm.wp_facade import config, init, synchronize, to_torch  # noqa: F401
