#!/usr/bin/env python
"""Kernel A/B experiments: build libmpmhip.so variants that differ in -D switches of the fast back end (csrc/fast.hip, resort.hip, p2g.hip, g2p.hip, dist.hip and their headers).

    python tools/build_variants.py name1:-DFOO=1,-DBAR=2 name2:-DFOO=0 ...

`name:ALL,-flag` compiles every source of the library with the flags (e.g. nofma:ALL,-ffp-contract=off).
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
    every = "ALL" in defs  # name:ALL,-flag,...  compiles every source with the flags (later flags override build.py's)
    only = [d[5:] for d in defs if d.startswith("ONLY=")] or None   # name:ONLY=p2g.hip,-flag  the flags go to that source alone
    return B.build_variant(name, [d for d in defs if d != "ALL" and not d.startswith("ONLY=")], every, only)


if __name__ == "__main__":
    B.build()
    with ThreadPoolExecutor(max_workers=4) as ex:
        for lib in ex.map(one, sys.argv[1:]):
            print(lib)
