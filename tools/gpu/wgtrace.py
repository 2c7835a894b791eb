#!/usr/bin/env python
"""Per-workgroup timeline of the p2g and g2p launches (the instruction-level profilers are not usable on this pool:
`rocprofv3 --att` needs librocprof-trace-decoder, absent from the image; `rocprofv3-avail list --pc-sampling` lists no agent).

    MPMHIP_LIB=mpmavatar_amd/lib/variants/libmpmhip_dbg.so python tools/gpu/wgtrace.py <scene> [pre_advance] [tag]

The -DMPMHIP_DEBUG=1 build stamps the 100 MHz constant clock at fixed points of every workgroup (WGT() in csrc/fast_device.hpp; a stamp
waits for all outstanding memory operations of wavefront 0 first, so it perturbs the kernel a little: compare the span with the
rocprofv3 average).  Prints, per kernel: span, workgroups by kind, phase durations (median / p90), how many workgroups run
concurrently over time, and the start-time structure ("rounds").  Saves the raw stamps to gpurun_out/wgtrace_<tag>_<scene>.npz."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from mpmavatar_amd import harness, scenes

scene = sys.argv[1] if len(sys.argv) > 1 else "sheet-500k"
pre = int(sys.argv[2]) if len(sys.argv) > 2 else 100
tag = sys.argv[3] if len(sys.argv) > 3 else "r03"
MAXWG = 16384
sim = harness.build_solver(scenes.REGISTRY[scene](), "cuda:0", mode="fast")
sv = sim.solver
harness.run(sim, pre, fused=True)
torch.cuda.synchronize()
out = {}
P2G = ["start", "record", "loads+clear", "forces+barrier", "scatter(w0)", "barrier", "flush"]
G2P = ["start", "rec+x+flags", "staged(w0)", "barrier", "sweep1+store(w0)", "-", "end"]
for rep in range(3):  # three single-substep samples
    sv._call("mpmhip_debug_wgtrace", 0, None, 0)
    harness.run(sim, 1, fused=True)
    for k, name in ((0, "p2g"), (1, "g2p")):
        buf = np.zeros((MAXWG, 8), np.uint64)
        sv._call("mpmhip_debug_wgtrace", k, buf.ctypes.data_as(C.c_void_p), MAXWG)
        out[f"{name}_{rep}"] = buf
st = sv.stats()
os.makedirs("gpurun_out", exist_ok=True)
np.savez_compressed(f"gpurun_out/wgtrace_{tag}_{scene}.npz", **out)
print(f"# workgroup timeline, {scene} after {pre} substeps ({st['n_active_blocks']} active blocks, rebins {st['rebins']})\n")


def analyse(name, buf, labels):
    t = buf[:, :7].astype(np.int64)
    ran = t[:, 0] > 0
    done = ran & (t[:, 6] > 0)
    t0 = t[ran, 0].min()
    us = lambda a: (a - t0) / 100.0
    end_all = t[done, 6].max()
    print(f"## {name}: {ran.sum()} workgroups started, {done.sum()} with work, span {us(end_all):.1f} us")
    full = done & (t[:, 1] > 0) & (t[:, 5] > 0 if name == "p2g" else t[:, 3] > 0)   # chunk workgroups (all stamps)
    other = done & ~full
    if other.any():
        d = (t[other, 6] - t[other, 0]) / 100.0
        print(f"   {other.sum()} extra workgroups (splat / clearing): duration median {np.median(d):.2f} us, p90 {np.quantile(d, .9):.2f}, max {d.max():.2f}; "
              f"start median {np.median(us(t[other, 0])):.1f} us, last end {us(t[other, 6].max()):.1f} us")
    sp = other & (t[:, 2] > 0) & (t[:, 3] > 0) & (t[:, 4] > 0)   # one-pass small-bin splat workgroups (stamps 2, 3, 4)
    if name == "p2g" and sp.any():
        ts = t[sp]
        seg = lambda a, b: (ts[:, b] - ts[:, a]) / 100.0
        print(f"   {sp.sum()} one-pass splat workgroups: life median {np.median(seg(0, 6)):.2f} us (p90 {np.quantile(seg(0, 6), .9):.2f}); start -> record + tile cleared "
              f"{np.median(seg(0, 2)):.2f}, -> faces loaded + LDS atomics {np.median(seg(2, 3)):.2f}, -> barrier {np.median(seg(3, 4)):.2f}, -> flushed {np.median(seg(4, 6)):.2f}; "
              f"start median {np.median(us(ts[:, 0])):.1f} us")
    idx = np.where(full)[0]
    tt = t[idx]
    dur = (tt[:, 6] - tt[:, 0]) / 100.0
    print(f"   {len(idx)} chunk workgroups: duration median {np.median(dur):.2f} us, p10 {np.quantile(dur, .1):.2f}, p90 {np.quantile(dur, .9):.2f}, max {dur.max():.2f}")
    prev = 0
    for s in range(1, 7):
        if (tt[:, s] == 0).all():
            continue
        d = (tt[:, s] - tt[:, prev]) / 100.0
        print(f"     {labels[prev]:>18} -> {labels[s]:<18}: median {np.median(d):6.2f} us   p90 {np.quantile(d, .9):6.2f}   mean {d.mean():6.2f}")
        prev = s
    # rounds: start times of the chunk workgroups
    s0 = np.sort(us(tt[:, 0]))
    hist, edges = np.histogram(s0, bins=np.arange(0, max(s0.max(), 1) + 1.0, 1.0))
    print("   chunk workgroup starts per us: " + " ".join(f"{int(h)}" for h in hist))
    e0 = us(tt[:, 6])
    hist, _ = np.histogram(e0, bins=np.arange(0, max(e0.max(), 1) + 1.0, 1.0))
    print("   chunk workgroup ends per us:   " + " ".join(f"{int(h)}" for h in hist))
    # concurrency over time (all workgroups with work)
    ev = np.concatenate([np.stack([us(t[done, 0]), np.ones(done.sum())], 1), np.stack([us(t[done, 6]), -np.ones(done.sum())], 1)])
    ev = ev[np.argsort(ev[:, 0])]
    conc = np.cumsum(ev[:, 1])
    grid = np.arange(0, us(end_all), 1.0)
    at = [conc[np.searchsorted(ev[:, 0], g, side="right") - 1] if g >= ev[0, 0] else 0 for g in grid]
    print("   workgroups in flight at each us: " + " ".join(f"{int(a)}" for a in at))
    hw = buf[idx, 7]
    xcc = (hw >> np.uint64(32)) & np.uint64(0xf)
    cu = (hw >> np.uint64(8)) & np.uint64(0xf)
    se = (hw >> np.uint64(13)) & np.uint64(0x7)
    sh = (hw >> np.uint64(12)) & np.uint64(0x1)
    cuid = xcc.astype(np.int64) * 1000 + se.astype(np.int64) * 100 + sh.astype(np.int64) * 50 + cu.astype(np.int64)
    n_cu = len(np.unique(cuid))
    per = np.bincount(np.unique(cuid, return_inverse=True)[1])
    print(f"   chunk workgroups ran on {n_cu} distinct CUs; per CU min {per.min()} / median {int(np.median(per))} / max {per.max()}; XCC histogram {np.bincount(xcc.astype(np.int64), minlength=8).tolist()}")
    print()


for rep in range(3):
    print(f"### sample {rep}")
    analyse("p2g", out[f"p2g_{rep}"], P2G)
    analyse("g2p", out[f"g2p_{rep}"], G2P)
