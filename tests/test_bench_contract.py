"""bench.py's output contract: exactly one JSON line on stdout with the driver's keys, the `roofline` and `cpu_baseline`
objects at N = 1.  CPU: the committed line of the last profile.  GPU: a short live run on the small cube."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KEYS = {"metric": str, "value": float, "unit": str, "n_gpus": int, "steps": int, "warmup": int, "ms_per_step": float,
        "higher_is_better": bool, "scaling": str, "dtype": str, "data": str, "config": dict}


def _check(o, n_gpus=1, with_cpu=True):
    for k, t in KEYS.items():
        assert k in o and isinstance(o[k], t), (k, o.get(k))
    assert "vs_baseline" in o and o["vs_baseline"] is None          # BASELINE.md publishes no number for this metric
    assert o["unit"] == "substeps/s" and o["higher_is_better"] is True and o["scaling"] == "strong" and o["data"] == "synthetic"
    assert o["n_gpus"] == n_gpus and "workload" in o["config"] and "model" not in o["config"]
    assert abs(o["value"] - 1e3 / o["ms_per_step"]) < 1e-6 * o["value"]
    if n_gpus == 1:
        r = o["roofline"]
        assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0
        assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9 and 0.0 < r["frac"] < 1.0
        assert r["traffic"] is None or r["traffic"] > 0
        if with_cpu:
            c = o["cpu_baseline"]
            assert c["kind"] in ("port", "reference") and c["cores"] >= 1 and c["value"] > 0 and c["unit"] == "substeps/s" and c["sample"]


def test_committed_bench_line_follows_the_contract():
    o = json.load(open(os.path.join(ROOT, "profiles", "r06_sheet-500k_bench.json")))
    _check(o)
    # round 3: traffic-based fractions beside the algorithmic ones, the CPU baseline at a probed thread count and on one thread
    assert 0 < o["roofline"]["traffic_frac"] < 1 and all("traffic_frac" in k for k in o["kernels"] if "alg_bytes" in k)
    assert 0 < o["substep_roofline"]["traffic_frac"] < 1
    assert o["cpu_baseline"]["cores"] <= 64 and o["cpu_baseline"]["serial"]["cores"] == 1 and o["cpu_baseline"]["serial"]["value"] > 0
    assert o["config"]["workload"] == "sheet-500k" and o["config"]["n_particles"] == 497762 and o["config"]["n_grid"] == 256
    # round 2: the roofline's kernel is one of the launches of the timed (fused) loop, and the steady state is reported
    assert o["kernels_mode"] == "fused-loop" and o["roofline"]["kernel"] in {k["name"] for k in o["kernels"]}
    assert o["phases_mode"].startswith("per-phase") and {p["name"] for p in o["phases"]} >= {"p2g", "g2p_v", "g2p_e", "grid_update"}
    assert 0 < o["value_draped"] < o["value"] and o["draped"]["advance"] == 2000 and o["draped"]["rebins_in_window"] >= 1
    assert o["roofline"]["traffic"] > 0 and o["roofline"]["traffic_source"].startswith("profiles/")


@pytest.mark.gpu
def test_live_bench_prints_exactly_one_json_line():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--scene", "cube-8k", "--steps", "40", "--warmup", "10",
                        "--no-cpu-baseline"], cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    print("\n".join(l for l in r.stderr.splitlines() if l.startswith("[bench +")))
    out = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(out) == 1, r.stdout[:500]
    _check(json.loads(out[0]), with_cpu=False)


@pytest.mark.gpu
def test_bench_gpus_n_without_a_launcher():
    """`python bench.py --gpus 2` as the driver calls it (no WORLD_SIZE in the environment): on a box with >= 2 GPUs the
    script starts the ranks itself and prints one line with n_gpus = 2; on a one-GPU box it stops with a message (no
    assert, no traceback); with the gloo test transport two ranks share the GPU and the sharded path runs either way."""
    import torch
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    base = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--scene", "cube-8k", "--steps", "20", "--warmup", "5",
            "--advance", "0", "--no-cpu-baseline", "--weak-n", "48", "--weak-grid", "64"]
    r = subprocess.run(base, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    if torch.cuda.device_count() >= 2:
        assert r.returncode == 0, r.stderr[-2000:]
        o = json.loads([l for l in r.stdout.splitlines() if l.strip()][-1])
        _check(o, n_gpus=2, with_cpu=False)
        assert o["config"]["exchange"] == "rccl"
    else:
        assert r.returncode != 0 and r.stdout.strip() == ""
        assert "needs one GPU per rank" in r.stderr and "Traceback" not in r.stderr
    r = subprocess.run(base, cwd=ROOT, env=dict(env, MPMHIP_DIST_BACKEND="gloo", OMP_NUM_THREADS="1"), capture_output=True,
                       text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    out = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(out) == 1, r.stdout[:500]
    o = json.loads(out[0])
    _check(o, n_gpus=2, with_cpu=False)
    # N > 1 lines carry the per-rank roofline and what the exchange moves
    assert o["roofline"]["bound"] == "hbm" and 0 < o["roofline"]["frac"] < 1 and o["kernels_mode"].startswith("fused-loop, rank 0")
    assert o["exchange"]["transport"] in ("torch", "rccl") and o["exchange"]["halo_bytes_per_substep_sent_by_rank0"] > 0
    # ... and the weak-scaling workload measured in the same run (two stacked sheets cut into two slabs; here a small one)
    w = o["weak_scaling"]
    assert "error" not in w, w
    assert w["value"] > 0 and w["n_particles"] == 2 * w["particles_per_rank"] and w["unit"] == "substeps/s"
    assert o["config"]["exchange"] == o["exchange"]["transport"]
    # ... and the per-rank compute floor beside the measured value (round 4): rank 0's shard alone, no exchange
    sf = o["shard_floor"]
    assert "error" not in sf, sf
    # (recorded, not judged: a ratio of two timings on a shared box is not a parity statement and may not stop the suite)
    assert sf["substeps_per_s_upper_bound"] > 0 and sf["local_particles"] > 0 and sf["measured_fraction_of_bound"] > 0
    print("shard_floor:", json.dumps(sf))
    # the in-library loop (what a multi-GPU node runs), its RCCL entry points bound to the shared-memory stand-in
    sys.path.insert(0, os.path.join(ROOT, "tests", "mock_rccl"))
    from build import build as build_mock
    r = subprocess.run(base, cwd=ROOT, capture_output=True, text=True, timeout=900,
                       env=dict(env, MPMHIP_DIST_BACKEND="gloo", OMP_NUM_THREADS="1", MPMHIP_DIST_TRANSPORT="rccl", MPMHIP_RCCL_LIB=build_mock()))
    assert r.returncode == 0, r.stderr[-3000:]
    o = json.loads([l for l in r.stdout.splitlines() if l.strip()][-1])
    _check(o, n_gpus=2, with_cpu=False)
    assert o["exchange"]["transport"] == "rccl" and o["exchange"]["halo"].startswith("peer-mapped")
    assert 0 < o["roofline"]["frac"] < 1
    # the weak-scaling measurement builds a second sharded simulation (second communicator, second set of peer links) in the
    # same processes while the first is alive
    assert "error" not in o["weak_scaling"] and o["weak_scaling"]["value"] > 0, o["weak_scaling"]


@pytest.mark.gpu
def test_bench_gpus_n_prints_its_headline_when_the_extras_run_out_of_time():
    """--extras-budget: the measurements behind the headline value (kernel events, shard floor, weak scaling, draped state) of an N > 1 run
    may not cost the run its number.  Two gloo ranks sharing the GPU, a draped phase far longer than the budget: still exactly one JSON
    line, exit code 0, the value in it and a note that the extras were cut."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--scene", "cube-8k", "--steps", "20", "--warmup", "5",
           "--advance", "400000", "--no-cpu-baseline", "--no-weak", "--no-shard-floor", "--extras-budget", "4"]
    r = subprocess.run(cmd, cwd=ROOT, env=dict(env, MPMHIP_DIST_BACKEND="gloo", OMP_NUM_THREADS="1"), capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    out = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(out) == 1, r.stdout[:500]
    o = json.loads(out[0])
    assert o["n_gpus"] == 2 and o["value"] > 0 and "timed out" in o["extras"] and "value_draped" not in o
