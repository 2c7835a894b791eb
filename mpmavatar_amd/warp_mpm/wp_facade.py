"""The four things the reference DRIVERS themselves call on ``wp`` (not a Warp emulation; nothing here computes):

    wp.init()                                   run_demo.py:264, train_material_params.py:399
    wp.to_torch(mpm_state.particle_x).clone()   run_demo.py:532, train_material_params.py:628,811
    wp.config.mode / wp.config.verify_cuda      run_demo.py:265-266 (commented out in the reference)
    wp.synchronize()                            (not called by the drivers; harmless to offer)

State fields of the shim already are torch tensors, so ``to_torch`` is the identity: reading the field is what syncs it.
A driver keeps its ``wp.*`` lines by importing this module as ``wp`` (``from warp_mpm import wp``), or unedited with
``mpmavatar_amd/compat`` on PYTHONPATH, whose ``warp/__init__.py`` re-exports exactly these names.
"""
import types

import torch

config = types.SimpleNamespace(mode="release", verify_cuda=False)


def init():
    """The library is loaded on first use; fail here, loudly, if it cannot be."""
    from .. import _lib
    _lib.load()


def to_torch(a, requires_grad=None):
    return a


def synchronize():
    if torch.cuda.is_available():
        torch.cuda.synchronize()
