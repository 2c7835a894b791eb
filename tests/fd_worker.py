"""Worker for tests/test_gpu_fd.py::test_fd_variants_sharded_over_ranks (torch.distributed.run, gloo; the ranks may share
one GPU): every rank simulates its slice of the four finite-difference runs, the losses are all-gathered, every rank
applies the same update; rank 0 compares with a single process that runs all four."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from mpmavatar_amd import fd, scenes  # noqa: E402


def problem(variants):
    sc = scenes.garment_cylinder(n_theta=24, n_h=12, n_grid=48, aniso=True, collider_subdiv=2)
    frames = fd.synthetic_problem(sc, n_frames=3, frame_dt=30e-4)
    dev = f"cuda:{dist.get_rank() % torch.cuda.device_count()}"
    return fd.MaterialFD(sc, frames, init=(1.0, 1.0, 1.04), lrs=(0.05, 0.05, 0.005), iterations=20, frame_dt=30e-4, substeps=30,
                         scale=0.8, shift=(0.2, 0.1, 0.3), concurrent=True, variants=variants, device=dev)


def main():
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    m = problem(list(fd.variant_slice(rank, world)))
    # every rank needs the same captured targets: rank-local capture is deterministic (same scene, same kernels)
    fd.capture(m, 1.0, 1.0, 1.0)
    outs = [m.train_one_step_sharded() for _ in range(3)]
    mine = torch.tensor([m.torch_param[k].item() for k in "DEH"], dtype=torch.float64)
    allp = [torch.zeros_like(mine) for _ in range(world)]
    dist.all_gather(allp, mine)
    ok = all(torch.equal(allp[0], q) for q in allp)            # identical parameters on every rank, no broadcast needed
    ok &= m.substeps_done == 3 * len(m.variants) * 3 * 30
    if rank == 0:
        ref = problem(None)
        fd.capture(ref, 1.0, 1.0, 1.0)
        ref_outs = [ref.train_one_step() for _ in range(3)]
        for a, b in zip(outs, ref_outs):
            ok &= bool(np.isclose(a["loss"], b["loss"], rtol=2e-3))
            ok &= all(np.isclose(a[k], b[k], rtol=1e-3, atol=1e-4) for k in "DEH")
        print(f"fd sharded over {world} ranks: loss {outs[-1]['loss']:.3e} (single process {ref_outs[-1]['loss']:.3e}), "
              f"H {outs[-1]['H']:.5f} vs {ref_outs[-1]['H']:.5f}", flush=True)
        ref.close()
    m.close()
    flag = torch.tensor([1 if ok else 0])
    dist.all_reduce(flag, op=dist.ReduceOp.MIN)
    dist.destroy_process_group()
    sys.exit(0 if flag.item() == 1 else 1)


if __name__ == "__main__":
    main()
