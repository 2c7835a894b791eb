"""warp.torch: zero-copy torch <-> array hand-off on the CPU (TEST INFRASTRUCTURE, see warp/__init__.py).

``from_torch`` aliases the tensor's memory (``tensor.numpy()`` shares storage), which is what the reference relies on
when it reads results back through ``wp.to_torch(state.particle_x)`` (run_demo.py:532).
"""
import numpy as np
import torch as _torch

import warp as _wp


def device_from_torch(device):
    return str(device)


def device_to_torch(device):
    return "cpu"


def dtype_from_torch(dt):
    if dt == _torch.float32:
        return _wp.float32
    if dt in (_torch.int32, _torch.int64):
        return _wp.int32
    raise TypeError(dt)


def dtype_is_compatible(torch_dtype, warp_dtype):
    np_t = getattr(warp_dtype, "_np_", None)
    if np_t is np.float32:
        return torch_dtype == _torch.float32
    if np_t is np.int32:
        return torch_dtype == _torch.int32
    return False


def from_torch(t, dtype=None, requires_grad=None, grad=None):
    if dtype is None:
        dtype = dtype_from_torch(t.dtype)
    a = _wp.array(dtype=dtype)
    data = t.detach().numpy()
    if t.dtype == _torch.int64:
        data = data.astype(np.int32)
    a._data = data
    a.ndim = data.ndim - len(getattr(a.dtype, "_shape_", ()))
    a.requires_grad = bool(requires_grad) if requires_grad is not None else False
    a._tensor = t
    return a


def to_torch(a, requires_grad=None):
    return _torch.from_numpy(a._data)
