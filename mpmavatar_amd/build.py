"""Build libmpmhip.so in-tree with hipcc for gfx950 (no CMake, no JIT cache).

    python -m mpmavatar_amd.build [--force]

The .so lands in mpmavatar_amd/lib/ (git-ignored, travels to the GPU box with the snapshot).
"""
from __future__ import annotations

import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
OBJDIR = os.path.join(LIBDIR, "obj")
LIB = os.path.join(LIBDIR, "libmpmhip.so")
FAST_SOURCES = ["fast.hip", "resort.hip", "p2g.hip", "g2p.hip", "dist.hip"]   # the fast back end (one translation unit until round 4)
SOURCES = ["api.hip", "common.hip", "baseline.hip"] + FAST_SOURCES + ["frames.hip"]
HEADERS = ["ctx.hpp", "mpm_math.hpp", "bc.hpp", "fast_device.hpp", "p2g_device.hpp", "g2p_device.hpp", "fast_state.hpp", os.path.join("..", "..", "include", "mpmhip.h")]
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-munsafe-fp-atomics", "-ffp-contract=fast-honor-pragmas",
         "-Wall", "-Wno-unused-function"]


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force: bool = False, verbose: bool = False) -> str:
    os.makedirs(OBJDIR, exist_ok=True)
    hdrs = [os.path.join(CSRC, h) for h in HEADERS]
    jobs = []
    for s in SOURCES:
        src = os.path.join(CSRC, s)
        obj = os.path.join(OBJDIR, s.replace(".hip", ".o"))
        if force or _stale(obj, [src] + hdrs):
            jobs.append([HIPCC] + FLAGS + ["-c", src, "-o", obj])

    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed:\n" + " ".join(cmd) + "\n" + r.stdout + r.stderr)
        return r.stderr

    with ThreadPoolExecutor(max_workers=4) as ex:
        for warn in ex.map(run, jobs):
            if verbose and warn:
                print(warn)
    objs = [os.path.join(OBJDIR, s.replace(".hip", ".o")) for s in SOURCES]
    if force or jobs or _stale(LIB, objs):
        run([HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs)
    return LIB


def build_variant(name: str, flags, every: bool = False, only=None) -> str:
    """Another build of the same sources with extra compiler flags, in lib/variants/libmpmhip_<name>.so (selected at run time with
    MPMHIP_LIB=<path>; the default library is untouched).  every = False compiles only the fast back end (FAST_SOURCES) with the flags.  Used for the
    contraction-free witness build (tests/test_gpu_ref_golden.py) and for kernel A/B experiments (tools/build_variants.py)."""
    build()
    vdir = os.path.join(LIBDIR, "variants")
    os.makedirs(vdir, exist_ok=True)
    hdrs = [os.path.join(CSRC, h) for h in HEADERS]
    objs = []
    for src in SOURCES:
        if (src not in FAST_SOURCES and not every) or (only is not None and src not in only):   # only: the flags go to these sources alone
            objs.append(os.path.join(OBJDIR, src.replace(".hip", ".o")))
            continue
        obj = os.path.join(vdir, f"{src[:-4]}_{name}.o")
        if _stale(obj, [os.path.join(CSRC, src)] + hdrs):
            r = subprocess.run([HIPCC] + FLAGS + list(flags) + ["-c", os.path.join(CSRC, src), "-o", obj], capture_output=True, text=True)
            if r.returncode != 0:
                raise RuntimeError("hipcc failed:\n" + r.stdout + r.stderr)
        objs.append(obj)
    lib = os.path.join(vdir, f"libmpmhip_{name}.so")
    if _stale(lib, objs):
        subprocess.check_call([HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib] + objs)
    return lib


NOFMA = ("nofma", ["-ffp-contract=off"], True)  # every source without FMA contraction: the strict-rounding witness build


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
