#!/usr/bin/env python
"""Kernel A/B experiments: build libmpmhip.so variants that differ in -D switches of csrc/fast.hip.

    python tools/build_variants.py name1:-DFOO=1,-DBAR=2 name2:-DFOO=0 ...

Each variant lands in mpmavatar_amd/lib/variants/libmpmhip_<name>.so (git-ignored, travels with gpurun) and is selected
at run time with MPMHIP_LIB=<path>.  The default library is untouched.
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from mpmavatar_amd import build as B  # noqa: E402


def one(spec):
    name, _, defs = spec.partition(":")
    defs = [d for d in defs.split(",") if d]
    vdir = os.path.join(B.LIBDIR, "variants")
    os.makedirs(vdir, exist_ok=True)
    obj = os.path.join(vdir, f"fast_{name}.o")
    cmd = [B.HIPCC] + B.FLAGS + defs + ["-c", os.path.join(B.CSRC, "fast.hip"), "-o", obj]
    subprocess.check_call(cmd, stderr=subprocess.DEVNULL)
    objs = [obj if s == "fast.hip" else os.path.join(B.OBJDIR, s.replace(".hip", ".o")) for s in B.SOURCES]
    lib = os.path.join(vdir, f"libmpmhip_{name}.so")
    subprocess.check_call([B.HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib] + objs)
    return lib


if __name__ == "__main__":
    B.build()
    with ThreadPoolExecutor(max_workers=4) as ex:
        for lib in ex.map(one, sys.argv[1:]):
            print(lib)
