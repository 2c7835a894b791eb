#!/usr/bin/env python3
"""kernel_regs.py [-Dflags ...]: VGPRs / SGPR+VGPR spills / LDS bytes of every kernel of p2g.hip and g2p.hip (hipcc -S into /tmp/isa)."""
import os, re, subprocess, sys
os.makedirs("/tmp/isa", exist_ok=True)
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
F = "--offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics -ffp-contract=fast-honor-pragmas -Wno-unused-function".split()
units = [u for u in ("p2g", "g2p") ]
procs = [subprocess.Popen(["/opt/rocm/bin/hipcc"] + F + sys.argv[1:] + ["-S", "--cuda-device-only", "-o", f"/tmp/isa/{u}.s", f"{ROOT}/mpmavatar_amd/csrc/{u}.hip"],
                          stderr=subprocess.DEVNULL) for u in units]
assert all(p.wait() == 0 for p in procs)
for u in units:
    txt = open(f"/tmp/isa/{u}.s").read()
    for blk in txt.split("  - .agpr_count:")[1:]:
        g = lambda k: re.search(rf"\.{k}:\s+(\S+)", blk).group(1)
        name = subprocess.run(["c++filt", g("name")], capture_output=True, text=True).stdout.strip()
        name = re.sub(r"\(mpm::.*", "", name).replace("void mpm::(anonymous namespace)::", "").replace("mpm::(anonymous namespace)::", "")
        print(f"{name:40s} vgpr {g('vgpr_count'):>4s}  vspill {g('vgpr_spill_count'):>3s}  sspill {g('sgpr_spill_count'):>3s}  lds {g('group_segment_fixed_size'):>6s}")
