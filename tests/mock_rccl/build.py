"""Build tests/mock_rccl/librccl_mock.so (TEST INFRASTRUCTURE: the shared-memory stand-in for RCCL that lets
`mpmhip_rccl_steps` run with several ranks on a one-GPU box; see mock_rccl.cpp)."""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "mock_rccl.cpp")
LIB = os.path.join(HERE, "librccl_mock.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")


def build(force=False):
    if force or not os.path.exists(LIB) or os.path.getmtime(SRC) > os.path.getmtime(LIB):
        r = subprocess.run([HIPCC, "-O2", "-std=c++17", "-fPIC", "-shared", "-x", "hip", "--offload-arch=gfx950", SRC, "-o", LIB, "-lrt"],
                           capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed:\n" + r.stdout + r.stderr)
    return LIB


if __name__ == "__main__":
    print(build(force=True))
