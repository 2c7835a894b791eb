#!/usr/bin/env python
"""Headline benchmark: MPM substeps/s on the synthetic 'sheet-500k' scene (497,762 particles, 256^3 grid;
BASELINE.json metric) on N MI355X GPUs of one node.

    python bench.py --gpus 1 --steps 200 --warmup 20
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

A "step" is one substep (MPMWARP.p2g2p: stress -> p2g -> grid -> g2p) over the whole scene, inputs resident in
HBM before the timed region.  N > 1 shards particles by spatial slab (strong scaling: total work fixed) with an
RCCL halo exchange of the shared grid blocks after p2g (mpmavatar_amd/dist.py).

With --gpus N > 1 and no launcher in the environment (WORLD_SIZE unset) the script starts the N ranks itself
(torch.distributed.run, 127.0.0.1); on a box with fewer GPUs it stops with a message instead.

Rank 0 prints ONE JSON line.  Besides the contract keys it carries
  "roofline":     dominant kernel OF THE TIMED LOOP (HIP events around the fused launches, same kernels as the timed
                  region), algorithmic bytes per launch / launch time vs 8 TB/s
  "kernels":      every launch of the timed loop the same way ("kernels_mode": "fused-loop"), and
  "phases":       the reference's phases, each as its own un-fused launch (time_profile keys of the reference)
  "value_draped": the same rate after --advance more substeps (the sheet has draped over the sphere, moves fast,
                  re-sorts often), with the re-sorts that fell into that window
  "cpu_baseline": the CPU oracle (restatement of the reference algorithm; the reference's Warp path cannot run
                  here) timed on this box's host cores on a bounded number of substeps of the same scene.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
COPY_CEILING_GBS = 6300.0  # the guide's copy ceiling, used only to FLAG a launch whose algorithmic-bytes rate exceeds what a copy can move
                           # (fusion removed traffic, not time); the box's own copy bandwidth is measured in every run
                           # (measure_copy_bandwidth) and reported as the second roofline denominator


def phase_bytes(sc, n_active, n_coll, n_mov):
    """Algorithmic bytes per substep of each phase (SURVEY.md 8(d) per-unit figures)."""
    ne, nv, nt = sc.n_elements, sc.n_vertices, sc.n_traditional
    return {
        "compute_stress_from_F_trial": 188 * ne + 116 * nt + 12 * nv,
        "p2g": 100 * ne + 76 * nv + 104 * nt + 16 * n_active,
        "grid_update": 28 * n_active + 68 * n_coll + 32 * n_mov,
        "g2p_v": (132 + 36) * ne + 72 * nv + 144 * nt + 12 * n_active,  # gather + C/d3 write of elements
        "g2p_e": (228 - 132 - 36) * ne,                                  # x,v,d1,d2 from the three vertices
    }


# kernels of the fused substep loop per reference phase (rocprofv3 names without template arguments, except where the
# argument selects the fused form; k_g2p<true> also does the grid stage)
FUSED_BYTES_NOTE = ("fused loop: k_stress_elem<true> also finalizes the previous substep's elements (g2p_e's x/v/d1/d2 part), "
                    "k_g2p<true> also runs the grid stage; bytes are attributed to the launch that moves them")
PHASE_KERNELS = {"p2g": ["k_p2g"], "g2p_v": ["k_g2p"], "grid_update": ["k_grid<true>"],
                 "compute_stress_from_F_trial": ["k_stress_elem<true>", "k_stress_trad"], "g2p_e": ["k_elem_finalize"]}


def pmc_traffic(phase, workload, only=None):
    """HBM bytes per launch of a phase's kernels from the newest committed PMC summary (profiles/*_pmc.json, written by
    tools/summarize_prof.py from separate `rocprofv3 --pmc FETCH_SIZE` / `--pmc WRITE_SIZE` passes of this command; reads
    carry the gfx950 x2 correction).  PMC counters cannot be collected from inside the process, so this is the profile of
    the same workload, not of this very run; returns (None, None) when no profile of the workload is committed."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "*_pmc.json")))
    if not files or workload != "sheet-500k":
        return None, None
    prof = json.load(open(files[-1]))
    tot = 0.0
    for want in ([only] if only else PHASE_KERNELS.get(phase, [])):
        for k, e in prof["kernels"].items():
            if k == want or ("<" not in want and k.split("<")[0] == want):
                tot += e["hbm_read_bytes"] + e["hbm_write_bytes"]
    return (tot or None), os.path.relpath(files[-1], ROOT)


def measure_copy_bandwidth(dev):
    """The box's own stream-copy bandwidth, measured here (the second roofline denominator beside the 8 TB/s spec
    peak): a device-to-device copy of float4-aligned buffers, bytes read + bytes written over the copy's HIP-event time, best of 5
    runs of 10 copies.  Two footprints: 2 x 1 GiB (streams through HBM) and 2 x 96 MiB (what a substep's ~0.2 GB working set does: it
    sits in the 256 MiB Infinity Cache)."""
    import torch
    res = {}
    for label, n_bytes in (("hbm_2x1GiB", 1 << 30), ("cache_resident_2x96MiB", 96 << 20)):
        a = torch.empty(n_bytes // 4, dtype=torch.float32, device=dev).normal_()
        b = torch.empty_like(a)
        for _ in range(3):
            b.copy_(a)
        best = 0.0
        for _ in range(5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                b.copy_(a)
            e1.record()
            e1.synchronize()
            best = max(best, 10 * 2 * n_bytes / (e0.elapsed_time(e1) * 1e-3) / 1e9)
        res[label + "_GBps"] = best
        del a, b
    torch.cuda.empty_cache()
    res["note"] = "torch device-to-device copy (read + write bytes / HIP-event time), measured in this run on this box"
    return res


def cpu_baseline(sc, budget_s=15.0, serial_budget_s=8.0):
    """Time the CPU oracle (OpenMP build, all host cores) on a bounded number of substeps of the same scene."""
    from oracle.scene_adapter import omp_threads, oracle_from_scene, run_scene
    # The thread count is probed, not taken from os.cpu_count(): the oracle's parallel regions are short and atomics-heavy, and on
    # the MI355X box 256 threads run 20x slower than 16 (profiles/r03_oracle_threads.txt).  Three substeps each, the fastest is timed.
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    best = None
    for th in sorted({t for t in (8, omp_threads(), 32) if t <= avail} or {1}):
        oc = oracle_from_scene(sc, omp=True, n_threads=th)
        run_scene(oc, sc, 1)  # warm caches / page in the dense grids
        t0 = time.perf_counter()
        run_scene(oc, sc, 3, k0=1)
        el = time.perf_counter() - t0
        if best is None or el < best[0]:
            best = (el, th, oc)
    cores, o = best[1], best[2]
    del best, oc
    n, t0 = 0, time.perf_counter()
    while True:
        run_scene(o, sc, 1, k0=n + 4)
        n += 1
        el = time.perf_counter() - t0
        if el > budget_s or n >= 400:
            break
    res = {"value": n / el, "unit": "substeps/s", "cores": cores, "kind": "port", "cpus_visible": os.cpu_count(),
           "sample": f"{n} substeps of {sc.name} (dense-grid OpenMP CPU restatement of the reference algorithm, {cores} threads: "
                     "the fastest of the thread counts probed)"}
    # ... and on ONE thread: the model of Warp's CPU device, which runs a kernel as a serial loop over its threads
    del o
    o1 = oracle_from_scene(sc, omp=False)
    run_scene(o1, sc, 1)
    n1, t1 = 0, time.perf_counter()
    while True:
        run_scene(o1, sc, 1, k0=n1 + 1)
        n1 += 1
        el1 = time.perf_counter() - t1
        if el1 > serial_budget_s or n1 >= 10:
            break
    res["serial"] = {"value": n1 / el1, "unit": "substeps/s", "cores": 1, "kind": "port",
                     "sample": f"{n1} substeps of {sc.name}, serial build of the same restatement"}
    return res


def main():
    # stdout carries exactly one line (the JSON result); everything else this process prints -- the shim repeats the
    # reference's "Particles initialized from torch data." messages -- goes to stderr
    # ... and so does what native libraries write to file descriptor 1 behind Python's back (RCCL prints a version banner
    # through C stdio when the first communicator is created; it would come out after the JSON line at exit)
    sys.stdout.flush()
    real_stdout = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)
    sys.stdout = sys.stderr
    _main(real_stdout)
    real_stdout.flush()


def _main(out_stream):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--scene", default="sheet-500k")
    ap.add_argument("--mode", default="fast", choices=["fast", "baseline"])
    ap.add_argument("--rebin-interval", type=int, default=0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernels", action="store_true", help="skip the HIP-event passes")
    ap.add_argument("--windows", type=int, default=0, help="timed windows of --steps substeps each (median reported); "
                    "0 = 7 when --steps < 100, else 1")
    ap.add_argument("--advance", type=int, default=2000, help="untimed substeps before the second (draped-state) measurement; "
                    "0 = skip it")
    ap.add_argument("--pre-advance", type=int, default=0, help="diagnostics: untimed substeps BEFORE the warm-up (profile the "
                    "draped state with --advance 0)")
    ap.add_argument("--weak", action="store_true", help="N > 1: time ONLY the weak-scaling workload (N stacked copies of the headline "
                    "sheet, one sheet's worth of particles per rank) instead of the strong-scaling headline")
    ap.add_argument("--no-weak", action="store_true", help="N > 1: skip the additional weak-scaling measurement")
    ap.add_argument("--no-shard-floor", action="store_true", help="N > 1: skip the per-rank compute-floor measurement")
    ap.add_argument("--weak-n", type=int, default=0, help="diagnostics / tests: vertices per side of the stacked sheets of the "
                    "weak-scaling workload (default: the headline sheet's 408) -- also runs it beside scenes other than sheet-500k")
    ap.add_argument("--weak-grid", type=int, default=256)
    ap.add_argument("--force-dist", action="store_true", help="diagnostics: run the sharded driver even with one rank "
                    "(launch under torch.distributed.run --nproc-per-node 1)")
    ap.add_argument("--extras-budget", type=int, default=420, help="N > 1: seconds the measurements BEHIND the headline value (kernel events, "
                    "shard floor, weak scaling, draped state) may take together; after that rank 0 prints the line as far as it got "
                    "(\"extras\": \"timed out ...\") and every rank exits -- a stuck extra must not cost the run its headline number; 0 = off")
    ap.add_argument("--watchdog", type=int, default=1500, help="N > 1: seconds after which a rank that is still running dumps its "
                    "Python stacks to stderr and exits (a rank that died leaves the others in a collective for RCCL's 10 minutes per "
                    "call otherwise); 0 = off")
    args = ap.parse_args()

    t_start = time.perf_counter()
    mark = lambda what: print(f"[bench +{time.perf_counter() - t_start:.1f}s] {what}", file=sys.stderr, flush=True)   # (stderr: where the time of a run goes)
    import torch
    from mpmavatar_amd import harness, scenes
    mark("imports")

    if not torch.cuda.is_available():
        sys.exit("bench.py needs MI355X GPUs (the solver has no CPU path)")
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # `python bench.py --gpus N` without a launcher: start the N ranks here (one process per GPU, RCCL)
        shared = os.environ.get("MPMHIP_DIST_BACKEND") == "gloo"  # test mode: ranks may share a GPU
        if torch.cuda.device_count() < args.gpus and not shared:
            sys.exit(f"bench.py --gpus {args.gpus}: this box has {torch.cuda.device_count()} GPU(s); "
                     "the sharded run needs one GPU per rank")
        import socket
        import subprocess
        for attempt in range(3):   # (a free port found by bind(0) can be taken again before the launcher's store binds it: try another)
            with socket.socket() as sk:
                sk.bind(("127.0.0.1", 0))
                port = sk.getsockname()[1]
            cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
                   "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
            r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
            sys.stderr.write(r.stderr)
            lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
            lost_port = any(k in r.stderr.lower() for k in ("address already in use", "eaddrinuse", "failed to listen"))
            if (r.returncode == 0 and lines) or not lost_port:
                break
        if r.returncode != 0 or not lines:
            sys.exit(f"bench.py: the {args.gpus}-rank run failed (exit code {r.returncode})")
        print(lines[-1], file=out_stream, flush=True)
        return
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        sys.exit(f"bench.py: --gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks")
    if world > 1 and args.watchdog > 0:
        import faulthandler
        faulthandler.dump_traceback_later(args.watchdog, exit=True)   # (stderr; the JSON line is the only thing on stdout)
    if os.environ.get("MPMHIP_DIST_BACKEND") == "gloo":
        local_rank = local_rank % torch.cuda.device_count()  # ranks may share a GPU in the gloo test mode
    torch.cuda.set_device(local_rank)
    dev = f"cuda:{local_rank}"

    weak_only = args.weak and world > 1 and args.scene == "sheet-500k"
    weak_scene = lambda: scenes.sheet_stack(world, n=args.weak_n or 408, n_grid=args.weak_grid)
    sc = weak_scene() if weak_only else scenes.REGISTRY[args.scene]()
    sharded = world > 1 or args.force_dist
    if sharded:
        import torch.distributed as dist
        from mpmavatar_amd import dist as mdist
        backend = os.environ.get("MPMHIP_DIST_BACKEND", "nccl")  # "gloo": host-staged exchange (2 ranks on 1 GPU tests)
        if backend == "nccl":
            import datetime
            dist.init_process_group("nccl", device_id=torch.device(dev),
                                    timeout=datetime.timedelta(seconds=min(args.watchdog, 600) if args.watchdog > 0 else 600))
        else:
            dist.init_process_group(backend)
        sim = mdist.build_sharded(sc, dev, rank, world, rebin_interval=args.rebin_interval)
        transport = sim.transport
        box = {"ss": sim}

        def run(n):  # a re-partition (particle migration) replaces the sharded simulation object
            box["ss"] = mdist.run(box["ss"], n)
        barrier = lambda: dist.barrier()
    else:
        sim = harness.build_solver(sc, dev, mode=args.mode, rebin_interval=args.rebin_interval)
        transport = None
        run = lambda n: harness.run(sim, n, fused=True)
        barrier = lambda: None

    def timed(n):
        """Wall time of n substeps, barrier + synchronize on both sides, MAX over ranks."""
        barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        run(n)
        torch.cuda.synchronize()
        barrier()
        el = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=dev)
        if sharded:
            import torch.distributed as dist
            dist.all_reduce(el, op=dist.ReduceOp.MAX)
        return float(el.item())

    mark("scene + solver built")
    copy_bw = measure_copy_bandwidth(dev) if rank == 0 else None
    mark("copy bandwidth measured")
    if args.pre_advance > 0:
        run(args.pre_advance)
    run(args.warmup)
    mark("warm-up done")
    # A window of K substeps is K x ~65 us: with the driver's K = 20 one window is 1.3 ms, shorter than the clock ramp of an idle GPU
    # and than a single re-sort, and one such sample was 14 % below the 400-substep figure.  Windows of EXACTLY K
    # substeps, every one bracketed by barrier + synchronize, are therefore repeated back to back until they cover at least
    # RESORT_SPAN substeps -- more than two re-sort intervals of the scene at t = 0 (one every ~250 substeps; every 65-100 once draped) --
    # and `value` is the MEAN rate over all of them: K x windows / total bracketed time, with the re-sorts that fell into the windows
    # paid for in it (round 5 reported the median window, which by construction never contained one).  The median
    # window, the fastest and the slowest stay beside it.
    RESORT_SPAN = 600
    n_win = args.windows if args.windows > 0 else max(7 if args.steps < 100 else 1, -(-RESORT_SPAN // args.steps))
    rebins0 = sim.solver.stats()["rebins"] if not sharded else None
    win_raw = [timed(args.steps) for _ in range(n_win)]
    rebins_timed = (sim.solver.stats()["rebins"] - rebins0) if not sharded else None
    win = sorted(win_raw)
    elapsed = sum(win_raw) / n_win
    ms_per_step = 1e3 * elapsed / args.steps

    out = {
        "metric": "MPM substeps/sec (500k particles, 256^3 grid)", "value": args.steps / elapsed, "unit": "substeps/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_per_step,
        "windows": n_win, "ms_per_step_min": 1e3 * win[0] / args.steps, "ms_per_step_median": 1e3 * win[len(win) // 2] / args.steps,
        "ms_per_step_max": 1e3 * win[-1] / args.steps, "value_median_window": args.steps / win[len(win) // 2],
        "timed_substeps": n_win * args.steps, "rebins_in_timed_windows": rebins_timed,
        "value_is": "mean rate over all timed windows (K substeps each, barrier + synchronize around every one), re-sorts inside them included",
        "higher_is_better": True, "scaling": "weak" if weak_only else "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": args.scene + (f" x{world} stacked (weak scaling)" if weak_only else ""), "scene_name": sc.name, "n_particles": sc.n_particles, "n_elements": sc.n_elements,
                   "n_vertices": sc.n_vertices, "n_traditional": sc.n_traditional, "n_grid": sc.n_grid,
                   "dt": sc.dt, "mode": args.mode, "parallelism": f"slab{world}" if world > 1 else "single",
                   "exchange": transport},
    }

    if copy_bw:
        out["copy_bandwidth"] = copy_bw
    mark("headline windows timed")
    # N > 1: everything below this point is extra information around a headline value that is already measured.  It involves more
    # collectives, a second sharded scene and thousands of further substeps on hardware this code has never run on: if it is not through
    # within --extras-budget seconds, rank 0 prints the line as far as it got and all ranks leave (same timer on every rank).
    extras_timer = None
    if world > 1 and args.extras_budget > 0:
        import threading

        def _bail():
            print(f"[bench] rank {rank}: extras budget used up, leaving", file=sys.stderr, flush=True)
            if rank == 0:
                snap = dict(out)
                snap["extras"] = f"timed out after {args.extras_budget} s: the line holds what was finished by then"
                print(json.dumps(snap, default=str), file=out_stream, flush=True)
            os._exit(0)
        extras_timer = threading.Timer(args.extras_budget, _bail)
        extras_timer.daemon = True
        extras_timer.start()
        print(f"[bench] rank {rank}: extras budget {args.extras_budget} s", file=sys.stderr, flush=True)

    if not sharded:
        sv = sim.solver
        st = sv.stats()
        n_act, n_col, n_mov = st["n_active_nodes"], st["n_collider_nodes"], st["n_mover_nodes"]
        b_alg = harness.algorithmic_bytes(sc, n_act, n_col, n_mov)
        out["config"].update({"n_active_nodes": n_act, "n_collider_nodes": n_col, "n_mover_nodes": n_mov,
                              "n_active_blocks": st["n_active_blocks"], "rebins": st["rebins"],
                              "fallback_particles": st["n_fallback_particles"], "alg_bytes_per_substep": b_alg["substep"]})
        out["substep_GBps"] = b_alg["substep"] / (ms_per_step * 1e-3) / 1e9
        out["substep_frac_of_hbm_peak"] = out["substep_GBps"] / HBM_PEAK_GBS
        if not args.no_kernels and args.mode == "fast":
            # (1) the launches of the timed loop itself: HIP events on the solver's stream around each fused launch
            pb = phase_bytes(sc, n_act, n_col, n_mov)
            cloth = sc.n_elements > 0
            fused_bytes = {
                "compute_stress_from_F_trial": (188 + 60) * sc.n_elements + 12 * sc.n_vertices,       # + element finalize
                "p2g": pb["p2g"] + 116 * sc.n_traditional + 28 * n_col + 16 * n_mov,                   # + trad. stress, splats
                "g2p_v": pb["g2p_v"] + 28 * n_act + 40 * n_col + 16 * n_mov,                            # + grid stage
            }
            fused_bytes["g2p2g"] = fused_bytes["p2g"] + fused_bytes["g2p_v"]   # (traditional-only scenes: one launch does both)
            fused_kernel = {"compute_stress_from_F_trial": "k_stress_elem<true>", "p2g": "k_p2g", "g2p_v": "k_g2p",
                            "rebin": "re-sort", "g2p2g": "k_g2p2g"}
            med = lambda v: sorted(v)[len(v) // 2] if v else 0.0

            def stamped_pass(n_prof):
                """The launches of n_prof substeps of the fused loop with their own start / stop stamps -> list of kernel dicts.
                `ms` of a hot launch is its OWN start -> stop time (hipExtLaunchKernelGGL stamps: the duration rocprofv3 --kernel-trace
                reports); `ms_events` keeps the hipEvent bracket around it, which adds ~3 us of packet processing.  (The loop also
                brackets one and two null kernels per substep, B1 and B2: 2 B1 - B2 is reported as the bracket's cost.)"""
                sv.enable_profiling(True, fused=True)
                sv.time_profile.clear()
                sv.kernel_profile.clear()
                harness.run(sim, n_prof, fused=True)
                sv.enable_profiling(False)
                ks_out = []
                b1, b2 = med(sv.time_profile.pop("event_null1", None)), med(sv.time_profile.pop("event_null2", None))
                out["event_null_brackets_ms"] = [b1, b2]
                kprof = dict(sv.kernel_profile)
                sv.kernel_profile.clear()
                for name, samples in sv.time_profile.items():
                    raw = sum(samples) / max(len(samples), 1)
                    ks = kprof.get(name)
                    ms = (sum(ks) / len(ks)) if ks else raw
                    k = {"name": fused_kernel.get(name, name), "phase": name, "ms": ms, "ms_events": raw, "launches": len(samples),
                         "timed_by": "kernel start/stop stamps" if ks else "event bracket"}
                    if name == "rebin":
                        k["ms_per_substep"] = sum(samples) / n_prof
                    if name in fused_bytes and ms > 0 and (cloth or name != "compute_stress_from_F_trial"):
                        k["alg_bytes"] = fused_bytes[name]
                        k["GBps"] = fused_bytes[name] / (ms * 1e-3) / 1e9
                        k["frac"] = k["GBps"] / HBM_PEAK_GBS
                        k["frac_of_measured_copy"] = k["GBps"] / copy_bw["hbm_2x1GiB_GBps"]
                        if k["GBps"] > COPY_CEILING_GBS:
                            k["exceeds_copy_ceiling"] = True   # read `traffic_frac`, not `frac`, for this launch
                        tr, src = pmc_traffic(name, args.scene, only=fused_kernel.get(name) if name == "g2p_v" else None)
                        if tr:
                            k["traffic"], k["traffic_GBps"], k["traffic_source"] = tr, tr / (ms * 1e-3) / 1e9, src
                            k["traffic_frac"] = k["traffic_GBps"] / HBM_PEAK_GBS
                    ks_out.append(k)
                sv.time_profile.clear()
                return ks_out
            n_prof = max(args.steps, 100)  # (untimed pass: enough samples per launch even with the driver's short windows)
            kernels = stamped_pass(n_prof)
            out["kernels"], out["kernels_mode"], out["kernels_note"] = kernels, "fused-loop", FUSED_BYTES_NOTE
            # the whole substep against the roofline: algorithmic bytes of all phases / wall time of the timed loop, and the same
            # with the PMC traffic of the three launches (the ~0.2 GB working set sits in the 256 MiB Infinity Cache: "fraction of
            # HBM peak" is a figure of merit against the reference algorithm's mandatory bytes, not a DRAM utilisation)
            tr_all = [k.get("traffic") for k in kernels if "alg_bytes" in k]
            out["substep_roofline"] = {"alg_bytes": b_alg["substep"], "frac": out["substep_frac_of_hbm_peak"],
                                       "frac_of_measured_copy": out["substep_GBps"] / copy_bw["hbm_2x1GiB_GBps"],
                                       "traffic": sum(tr_all) if all(tr_all) and tr_all else None,
                                       "traffic_frac": (sum(tr_all) / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBS) if all(tr_all) and tr_all else None}
            dom = max((k for k in kernels if "alg_bytes" in k), key=lambda k: k["ms"])
            out["roofline"] = {"bound": "hbm", "kernel": dom["name"], "achieved": dom["GBps"], "peak": HBM_PEAK_GBS,
                               "unit": "GB/s", "frac": dom["frac"], "traffic": dom.get("traffic"), "traffic_frac": dom.get("traffic_frac"),
                               "traffic_source": dom.get("traffic_source"), "alg_bytes_per_launch": dom["alg_bytes"],
                               "ms_per_launch": dom["ms"], "ms_events": dom["ms_events"],
                               "peak_copy_measured": copy_bw["hbm_2x1GiB_GBps"], "frac_of_measured_copy": dom["GBps"] / copy_bw["hbm_2x1GiB_GBps"],
                               "measured": "launch of the fused loop timed by the launch's own start/stop timestamps (hipExtLaunchKernelGGL); ms_events = the hipEvent bracket around it"}
            # (2) the reference's phases, each as its own launch (what MPMWARP.time_profile reports)
            sv.enable_profiling(True)
            harness.run(sim, min(args.steps, 50), fused=False)
            sv.enable_profiling(False)
            phases = []
            for name, samples in sv.time_profile.items():
                ms = sum(samples) / max(len(samples), 1)
                k = {"name": name, "ms": ms}
                if name in pb and ms > 0:
                    k["alg_bytes"], k["GBps"] = pb[name], pb[name] / (ms * 1e-3) / 1e9
                    k["frac"] = k["GBps"] / HBM_PEAK_GBS
                phases.append(k)
            out["phases"], out["phases_mode"] = phases, "per-phase (un-fused launches, one sync each; not the timed loop)"
    if sharded and not args.no_kernels:
        # per-rank view of rank 0: its launches of the sharded loop (HIP events, same kernels as the timed region), the
        # dominant one against the roofline with this rank's algorithmic bytes, and what the halo exchange moves
        ss = box["ss"]
        sv = ss.sim.solver
        lsc = ss.shard.scene
        owned = lsc.selection == 0 if lsc.selection is not None else None
        st = sv.stats()
        n_act = st["n_active_nodes"]
        ne_o = int(owned[:lsc.n_elements].sum()) if owned is not None else lsc.n_elements
        nv_o = int(owned[lsc.n_elements + lsc.n_traditional:].sum()) if owned is not None else lsc.n_vertices
        nt_o = int(ss.shard.own_t.size)          # (the local traditional class also holds the free rows of the migration slack)
        rank_bytes = {"compute_stress_from_F_trial": (188 + 60) * lsc.n_elements + 12 * lsc.n_vertices,   # ghosts run the stress update too
                      "p2g": 100 * ne_o + 76 * nv_o + 220 * nt_o + 16 * n_act,
                      "g2p_v": 168 * lsc.n_elements + 72 * lsc.n_vertices + 144 * nt_o + 40 * n_act}       # ghosts gather for themselves
        sv.enable_profiling(True, fused=True)
        sv.time_profile.clear()
        run(min(args.steps, 100))
        sv = box["ss"].sim.solver
        sv._collect_profile()
        sv.enable_profiling(False)
        names = {"compute_stress_from_F_trial": "k_stress_elem<true>", "p2g": "k_p2g", "g2p_v": "k_g2p", "rebin": "re-sort",
                 "halo_exchange": "ncclSend/ncclRecv group"}
        kernels = []
        for name, samples in sv.time_profile.items():
            ms = sum(samples) / max(len(samples), 1)
            k = {"name": names.get(name, name), "phase": name, "ms": ms, "launches": len(samples)}
            if name in rank_bytes and ms > 0 and rank_bytes[name] > 0:
                k["alg_bytes"], k["GBps"] = rank_bytes[name], rank_bytes[name] / (ms * 1e-3) / 1e9
                k["frac"] = k["GBps"] / HBM_PEAK_GBS
            kernels.append(k)
        out["kernels"], out["kernels_mode"] = kernels, "fused-loop, rank 0 of %d" % world
        with_bytes = [k for k in kernels if "alg_bytes" in k]
        if with_bytes:
            dom = max(with_bytes, key=lambda k: k["ms"])
            out["roofline"] = {"bound": "hbm", "kernel": dom["name"], "achieved": dom["GBps"], "peak": HBM_PEAK_GBS, "unit": "GB/s",
                               "frac": dom["frac"], "traffic": None, "alg_bytes_per_launch": dom["alg_bytes"],
                               "ms_per_launch": dom["ms"], "measured": "HIP events around the launch in the sharded loop, rank 0"}
        ch = 8 if sv.particle_movers else 4
        import ctypes
        hb = ctypes.c_int64(0)
        sv._call("mpmhip_dist_halo_bytes", ctypes.byref(hb))
        halo_bytes = int(hb.value)
        halo = None
        if box["ss"].transport == "rccl":
            ht = ctypes.c_int32(0)
            sv._call("mpmhip_dist_halo_transport", ctypes.byref(ht))
            halo = "peer-mapped buffers (HIP IPC, flags in the receiver's memory)" if ht.value == 1 else "ncclSend/ncclRecv"
        out["exchange"] = {"transport": box["ss"].transport, "halo": halo, "halo_bytes_per_substep_sent_by_rank0": halo_bytes,
                           "channels_per_node": ch, "peers_of_rank0": len(box["ss"].static) - 1,
                           "halo_exchange_us": next((k["ms"] * 1e3 for k in kernels if k["phase"] == "halo_exchange"), None),
                           "re_partitions": box["ss"].migrations, "on_device_migrations": box["ss"].trad_migrations,
                           "particles_migrated_on_device": box["ss"].trad_migrated}
    mark("kernel events")
    if sharded and world > 1 and not args.no_shard_floor:
        # What this N can reach at best: rank 0 runs ITS shard (owned particles + ghost copies) as an ordinary single-GPU scene, no
        # exchange at all -- the per-rank compute floor of the slab decomposition (three latency-floored launches per substep).
        # The first run on a real node is then self-explaining: measured value vs this bound = what the halo exchange costs.
        try:
            if rank == 0:
                fsim = harness.build_solver(box["ss"].shard.scene, dev, mode="fast")
                n_fl = max(args.steps, 50)
                harness.run(fsim, max(args.warmup, 20) + n_fl, fused=True)   # (warm-up incl. one window's worth: first re-sort, code objects)
                torch.cuda.synchronize()
                wins = []
                for _ in range(5):                                           # windows like the headline: the MEDIAN is reported
                    tf = time.perf_counter()
                    harness.run(fsim, n_fl, fused=True)
                    torch.cuda.synchronize()
                    wins.append(1e6 * (time.perf_counter() - tf) / n_fl)
                us = sorted(wins)[len(wins) // 2]
                out["shard_floor"] = {"us_per_substep_rank0_alone": us, "substeps_per_s_upper_bound": 1e6 / us,
                                      "us_per_substep_windows": wins, "rebins": int(fsim.solver.stats()["rebins"]),
                                      "local_particles": int(box["ss"].shard.scene.n_particles), "measured_fraction_of_bound": out["value"] * us / 1e6,
                                      "note": "rank 0's shard as a single-GPU scene without any exchange (median of 5 windows); the sharded run cannot beat it"}
                del fsim
        except Exception as e:  # noqa: BLE001 - the headline line must come out whatever happens here
            out["shard_floor"] = {"error": f"{type(e).__name__}: {e}"}
        finally:
            barrier()               # (every rank, whatever happened on rank 0: the others wait here)
    mark("shard floor")
    if sharded and world > 1 and not weak_only and not args.no_weak and (args.scene == "sheet-500k" or args.weak_n):
        # the regime the slab decomposition is made for: the same per-rank work at every N (one sheet's worth of particles per
        # rank: N stacked copies of the headline sheet in the same grid).  Reported beside the strong-scaling headline value.
        try:
            wsc = weak_scene()
            wbox = {"ss": mdist.build_sharded(wsc, dev, rank, world, rebin_interval=args.rebin_interval)}
            wbox["ss"] = mdist.run(wbox["ss"], args.warmup)
            barrier(); torch.cuda.synchronize()
            tw = time.perf_counter()
            wbox["ss"] = mdist.run(wbox["ss"], args.steps)
            torch.cuda.synchronize(); barrier()
            elw = torch.tensor([time.perf_counter() - tw], dtype=torch.float64, device=dev)
            dist.all_reduce(elw, op=dist.ReduceOp.MAX)
            elw = float(elw.item())
            out["weak_scaling"] = {"workload": f"{world} stacked copies of sheet-500k, x-slabs", "n_particles": int(wsc.n_particles),
                                   "particles_per_rank": int(wsc.n_particles // world), "value": args.steps / elw,
                                   "unit": "substeps/s", "ms_per_step": 1e3 * elw / args.steps,
                                   "particle_substeps_per_s": wsc.n_particles * args.steps / elw,
                                   "note": "same metric on N x the particles: compare with the N = 1 headline value"}
            del wbox
        except Exception as e:  # noqa: BLE001 - the headline line must come out whatever happens here
            out["weak_scaling"] = {"error": f"{type(e).__name__}: {e}"}
    mark("weak scaling")
    if args.advance > 0:
        # the steady state: after `advance` more substeps the sheet lies draped over the sphere, moves at metres per second and
        # the particle order is rebuilt every few dozen substeps; re-sorts inside the window are part of the number
        run(args.advance)
        n_d = max(args.steps, 200)
        r0 = sim.solver.stats()["rebins"] if not sharded else None
        el = timed(n_d)
        out["value_draped"] = n_d / el
        out["ms_per_step_draped"] = 1e3 * el / n_d
        out["draped"] = {"advance": args.advance, "steps": n_d,
                         "rebins_in_window": (sim.solver.stats()["rebins"] - r0) if not sharded else None}
        if not sharded:
            st = sim.solver.stats()
            b_d = harness.algorithmic_bytes(sc, st["n_active_nodes"], st["n_collider_nodes"], st["n_mover_nodes"])
            out["draped"].update({"n_active_nodes": st["n_active_nodes"], "n_collider_nodes": st["n_collider_nodes"],
                                  "fallback_particles": st["n_fallback_particles"],
                                  "substep_frac_of_hbm_peak": b_d["substep"] / (out["ms_per_step_draped"] * 1e-3) / 1e9 / HBM_PEAK_GBS})
    mark("draped state")
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(sc)
        mark("cpu baseline")
    if extras_timer is not None:
        extras_timer.cancel()
    if rank == 0:
        print(json.dumps(out), file=out_stream, flush=True)
    if sharded:
        import torch.distributed as dist
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
