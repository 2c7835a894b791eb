"""warp.context: only the ``Devicelike`` annotation is used by the reference (mpm_data_structure.py:56)."""
from typing import Any

Devicelike = Any
