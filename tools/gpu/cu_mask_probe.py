#!/usr/bin/env python
"""How does hipExtStreamCreateWithCUMask enumerate the CUs of an MI355X (8 XCDs x 32 CUs)?  Runs the garment scene on streams with different
masks and prints, from the debug build's per-workgroup HW_ID / XCC_ID stamps, which XCDs and how many distinct CUs the p2g chunk workgroups ran
on, and the substep rate.    MPMHIP_LIB=.../libmpmhip_dbg.so python tools/gpu/cu_mask_probe.py"""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from mpmavatar_amd import harness, scenes

hip = C.CDLL("libamdhip64.so")
torch.cuda.init()

def masked_stream(words):
    s = C.c_void_p()
    arr = (C.c_uint32 * len(words))(*words)
    rc = hip.hipExtStreamCreateWithCUMask(C.byref(s), len(words), arr)
    assert rc == 0, rc
    return torch.cuda.ExternalStream(s.value)

def bits(pred, n=256):
    w = [0] * (n // 32)
    for i in range(n):
        if pred(i): w[i // 32] |= 1 << (i % 32)
    return w

MASKS = {"all 256": bits(lambda i: True), "bits 0..63": bits(lambda i: i < 64), "bits 0..127": bits(lambda i: i < 128),
         "bits i%8==0": bits(lambda i: i % 8 == 0), "bits i%8<2": bits(lambda i: i % 8 < 2), "bits 64..127": bits(lambda i: 64 <= i < 128)}
MAXWG = 16384
for name, words in MASKS.items():
    st = masked_stream(words)
    with torch.cuda.stream(st):
        sim = harness.build_solver(scenes.REGISTRY["garment-120k-aniso"](), "cuda:0", mode="fast")
        sv = sim.solver
        harness.run(sim, 100, fused=True)
        st.synchronize()
        t0 = time.perf_counter(); harness.run(sim, 400, fused=True); st.synchronize(); rate = 400 / (time.perf_counter() - t0)
        msg = f"{name:14s}: {rate:8.0f} substeps/s"
        if "dbg" in os.environ.get("MPMHIP_LIB", ""):
            sv._call("mpmhip_debug_wgtrace", 0, None, 0)
            harness.run(sim, 1, fused=True)
            buf = np.zeros((MAXWG, 8), np.uint64)
            sv._call("mpmhip_debug_wgtrace", 0, buf.ctypes.data_as(C.c_void_p), MAXWG)
            ran = buf[:, 0] > 0
            hw = buf[ran, 7]
            xcc = ((hw >> np.uint64(32)) & np.uint64(0xf)).astype(np.int64)
            cu = ((hw >> np.uint64(8)) & np.uint64(0xf)).astype(np.int64); se = ((hw >> np.uint64(13)) & np.uint64(0x7)).astype(np.int64); sh = ((hw >> np.uint64(12)) & np.uint64(1)).astype(np.int64)
            cuid = xcc * 1000 + se * 100 + sh * 50 + cu
            msg += f"; XCC histogram {np.bincount(xcc, minlength=8).tolist()}, {len(np.unique(cuid))} distinct CUs"
        print(msg, flush=True)
        del sim
