"""A short run of the randomised back-end cross-check (tools/gpu/fuzz.py: random blobs / sheets / garments / demo mixes,
grid sizes, materials, speeds, time steps, re-sort policies; fast kernels vs reference-structured baseline kernels)."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
@pytest.mark.parametrize("seed", [11, 12])
def test_fuzz_fast_vs_baseline(seed):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "gpu", "fuzz.py"), "25", str(seed)], cwd=ROOT,
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert r.stdout.count("\nok ") + r.stdout.startswith("ok ") >= 15
