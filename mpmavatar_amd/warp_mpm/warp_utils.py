"""torch -> solver array hand-off checks (shim of /root/reference/warp_mpm/warp_utils.py).

The reference's ``from_torch_safe`` wraps a torch tensor as a ``wp.array`` without copying and raises
``RuntimeError`` on dtype / inner-shape / stride mismatch (warp_utils.py:23-49).  Here the "array" is the
tensor itself -- the C ABI takes ``tensor.data_ptr()`` -- so this function only performs the same checks.
"""
from __future__ import annotations

import torch

_INNER = {"float32": (), "int32": (), "vec3": (3,), "mat33": (3, 3)}


def from_torch_safe(t: torch.Tensor, dtype: str = "float32", requires_grad=None, grad=None) -> torch.Tensor:
    if dtype not in _INNER:
        raise RuntimeError(f"Unsupported target dtype {dtype}")
    want = torch.int32 if dtype == "int32" else torch.float32
    if t.dtype != want:
        raise RuntimeError(f"Incompatible data types: {t.dtype} and {dtype}")
    inner = _INNER[dtype]
    shape = tuple(t.shape)
    if inner:
        if len(inner) > len(shape) or shape[-len(inner):] != inner:
            raise RuntimeError(
                f"Could not convert Torch tensor with shape {shape} to array with dtype={dtype}, ensure that source "
                f"inner shape is {inner}")
    if not t.is_contiguous():
        raise RuntimeError(
            f"Could not convert Torch tensor with shape {shape} to array with dtype={dtype}, because the source "
            "strides are not contiguous")
    return t


def to_torch(a: torch.Tensor) -> torch.Tensor:
    """Stand-in for ``wp.to_torch`` at the reference's call sites (run_demo.py:532): state fields already are
    torch tensors."""
    return a
