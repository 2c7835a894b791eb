"""The C ABI without Python: examples/c_abi_demo.cpp includes include/mpmhip.h, links libmpmhip.so and drives a falling
cube through it with device memory from the HIP runtime.  CPU: it compiles and links against the built library (every
entry point it uses is declared and exported).  GPU: it runs and reproduces free fall of the centre of mass."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")


def _build(out):
    from mpmavatar_amd import build
    lib = os.path.dirname(build.build())
    cmd = [HIPCC, "-O2", "-I" + os.path.join(ROOT, "include"), os.path.join(ROOT, "examples", "c_abi_demo.cpp"), "-o", out,
           "-L" + lib, "-lmpmhip", "-Wl,-rpath," + lib]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    return out


def test_c_program_compiles_and_links_against_the_library(tmp_path):
    exe = _build(str(tmp_path / "c_abi_demo"))
    assert os.path.getsize(exe) > 0
    r = subprocess.run(["ldd", exe], capture_output=True, text=True)
    assert "libmpmhip.so" in r.stdout and "not found" not in r.stdout.split("libmpmhip.so")[1].split("\n")[0]


@pytest.mark.gpu
def test_c_program_runs_the_solver_without_python(tmp_path):
    exe = _build(str(tmp_path / "c_abi_demo"))
    r = subprocess.run([exe, "300"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "C ABI demo: OK" in r.stdout
