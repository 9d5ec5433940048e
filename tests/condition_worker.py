"""Worker of test_bench_clock_conditioning_leaves_every_rank_after_the_same_number_of_steps (torch.distributed.run, gloo, CPU): bench.py's clock-conditioning loop with a step
that is a collective, ranks of very different speed (rank 1 sleeps 3 ms per step, rank 0 does not): every rank must run the same number of steps and return."""
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "carla-ppo_amd"), ROOT):
    if p not in sys.path:
        sys.path.insert(0, p)

import bench  # noqa: E402
from mi355 import dist as midist  # noqa: E402


def main(out):
    world, rank, _ = midist.init_from_env("gloo")
    buf = torch.zeros(1024)
    seen = []

    def step(i):
        seen.append(i)
        if rank == 1:
            time.sleep(0.003)                           # the slow rank: on its own clock it would stop twenty or forty steps earlier than rank 0
        buf.fill_(1.0)
        torch.distributed.all_reduce(buf)               # every step is a collective, as the data-parallel SGD step is
        assert float(buf[0]) == world

    if rank == 1:
        time.sleep(0.08)                                # ... and it ENTERS the loop 80 ms late (its clock starts later: on their own clocks the ranks would leave several chunks apart)
    n = bench.condition_clocks(step, 120.0, world, torch.device("cpu"), chunk=5, sync=lambda: None)
    assert seen == list(range(n))
    json.dump({"rank": rank, "steps": n}, open(os.path.join(out, "cond_rank%d.json" % rank), "w"))
    torch.distributed.barrier()


if __name__ == "__main__":
    main(sys.argv[1])
