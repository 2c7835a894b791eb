"""CPU worker (gloo): MaterialFD.train_one_step_sharded with the simulations replaced by a closed-form loss, so that the
gather / update bookkeeping is tested without a GPU (tests/test_fd_host.py)."""
import os
import sys

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from mpmavatar_amd import fd  # noqa: E402


class Analytic(fd.MaterialFD):
    def losses(self, D, E, H):
        f = lambda d, e, h: (d - 2.0) ** 2 + 0.5 * (e - 0.5) ** 2 + 10.0 * (h - 1.1) ** 2
        return [f(D + fd.DELTAS[i][0], E + fd.DELTAS[i][1], H + fd.DELTAS[i][2]) for i in self.variants]


def main():
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    m = Analytic(None, [], build=False, variants=list(fd.variant_slice(rank, world)), iterations=50)
    ref = Analytic(None, [], build=False, iterations=50)
    ok = True
    for _ in range(10):
        a, b = m.train_one_step_sharded(), ref.train_one_step()
        ok &= a == b                                    # bit-identical to the single process, on every rank
    ok &= m.best_params == ref.best_params and m.step == 10
    try:
        Analytic(None, [], build=False, variants=[0], iterations=5).train_one_step_sharded()
        ok &= world == 4 and rank == 0                  # [0] is rank 0's slice only when four ranks share the runs
    except RuntimeError:
        pass
    flag = torch.tensor([1 if ok else 0])
    dist.all_reduce(flag, op=dist.ReduceOp.MIN)
    dist.destroy_process_group()
    sys.exit(0 if flag.item() == 1 else 1)


if __name__ == "__main__":
    main()
