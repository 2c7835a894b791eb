"""C-ABI checks that need no GPU: libmpmhip.so loads, exports every symbol include/mpmhip.h declares, the ctypes
mirror matches the header, and creation fails loudly (no CPU fallback) when no HIP device is visible."""
import ctypes as C
import os
import re

import pytest

from mpmavatar_amd import _lib as L
from mpmavatar_amd import build as hipbuild

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = open(os.path.join(ROOT, "include", "mpmhip.h")).read()


@pytest.fixture(scope="module")
def lib():
    hipbuild.build()
    return L.load()


def declared_functions():
    # every "<type> mpmhip_xxx(" at the start of a declaration
    return sorted(set(re.findall(r"^\s*(?:const\s+)?(?:int|void|double|char)\s*\*?\s*(mpmhip_\w+)\s*\(", HEADER, re.M)))


def test_every_declared_symbol_is_exported_and_bound(lib):
    names = declared_functions()
    assert len(names) >= 35
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/mpmhip.h but not exported by libmpmhip.so"
        assert n in L.SIGNATURES, f"{n} has no ctypes signature in mpmavatar_amd/_lib.py"
    assert sorted(L.SIGNATURES) == names


def test_header_cites_the_reference_interfaces():
    for cite in ("mpm_solver.py:229-536", "mpm_solver.py:14-51", "mpm_data_structure.py:158-419", "mpm_solver.py:805-919",
                 "mpm_solver.py:661-802", "mpm_solver.py:564-658", "mpm_solver.py:986-1053"):
        assert cite in HEADER


def test_struct_layouts_match_header():
    def fields(name):
        end = HEADER.index("} " + name + ";")
        body = HEADER[HEADER.rindex("typedef struct {", 0, end) + len("typedef struct {"):end]
        body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
        out = []
        for decl in body.split(";"):
            decl = decl.strip()
            if not decl:
                continue
            for part in decl.split(","):
                out.append(re.sub(r"\[.*\]", "", part.replace("*", " ").split()[-1]))
        return out
    assert fields("mpmhip_config") == [f for f, _ in L.Config._fields_]
    assert fields("mpmhip_state_ptrs") == [f for f, _ in L.StatePtrs._fields_]
    assert fields("mpmhip_model_ptrs") == [f for f, _ in L.ModelPtrs._fields_]
    assert fields("mpmhip_model_scalars") == [f for f, _ in L.ModelScalars._fields_]
    assert fields("mpmhip_stats") == [f for f, _ in L.Stats._fields_]
    assert fields("mpmhip_dist_peer") == [f for f, _ in L.DistPeer._fields_]


def test_version_and_device_count(lib):
    assert lib.mpmhip_version() == 100
    assert lib.mpmhip_device_count() >= 0


def test_create_validates_and_never_falls_back_to_cpu(lib):
    ctx = L.vp()
    bad = L.Config(8, 9, 0, 16, 2.0, 0, 0, 0, 0, 0, 0, 0, None)      # more elements than particles
    assert lib.mpmhip_create(C.byref(bad), C.byref(ctx)) == L.ERR_INVALID
    assert b"inconsistent" in lib.mpmhip_last_error(None)
    if lib.mpmhip_device_count() == 0:
        ok = L.Config(8, 0, 0, 16, 2.0, 0, 0, 0, 0, 0, 0, 0, None)
        assert lib.mpmhip_create(C.byref(ok), C.byref(ctx)) == L.ERR_NO_DEVICE
        assert b"no CPU fallback" in lib.mpmhip_last_error(None)
        assert not ctx.value


def test_shim_refuses_cpu_device(lib):
    from mpmavatar_amd.warp_mpm import MPMWARP
    with pytest.raises(L.MPMHipError):
        MPMWARP(8, 0, 0, n_grid=16, grid_lim=2.0, device="cpu")


def test_product_never_imports_the_oracle():
    """The oracle is test infrastructure: nothing under mpmavatar_amd/ may import, load or link it."""
    for dirpath, _, files in os.walk(os.path.join(ROOT, "mpmavatar_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".h")):
                txt = open(os.path.join(dirpath, f)).read()
                assert "liboracle" not in txt and "mpm_oracle" not in txt, f
                if f.endswith(".py"):
                    assert not re.search(r"^\s*(from|import)\s+oracle\b", txt, re.M), f


def test_rccl_stand_in_exports_what_the_library_binds():
    """tests/mock_rccl (test infrastructure for the multi-rank GPU tests) must offer every RCCL entry point that
    csrc/dist.hip resolves with dlsym -- a new binding without a stand-in would make those tests fall back silently."""
    import re
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    src = "".join(open(os.path.join(root, "mpmavatar_amd", "csrc", f)).read() for f in ("dist.hip", "fast_state.hpp", "fast.hip"))
    bound = set(re.findall(r'sym\("(nccl\w+)"\)', src))
    assert len(bound) == 10, bound
    sys.path.insert(0, os.path.join(root, "tests", "mock_rccl"))
    from build import build as build_mock
    out = subprocess.run(["nm", "-D", "--defined-only", build_mock()], capture_output=True, text=True, check=True).stdout
    exported = set(re.findall(r" T (nccl\w+)", out))
    assert bound <= exported, bound - exported
