"""Child of tests/test_launch.py: what a rank of `bench.py --gpus N` does minus the GPU -- join the process group the
launcher's environment describes (gloo here), exchange per-chromosome rows, rank 0 prints ONE JSON line."""
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

from pymasc_amd import sharding  # noqa: E402


def main():
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    mode = sys.argv[1] if len(sys.argv) > 1 else "ok"
    if mode == "fail" and rank == world - 1:
        sys.exit(3)                                     # a rank dies before the rendezvous completes its first collective
    dist.init_process_group("gloo", rank=rank, world_size=world)    # MASTER_ADDR / MASTER_PORT from the launcher
    lengths = [9000, 4000, 7000, 1200, 6500]
    assignment = sharding.lpt_assign(lengths, world)
    max_slots = max(len(a) for a in assignment)
    local = torch.zeros((max_slots, 2, 8), dtype=torch.int64)
    for slot, job in enumerate(assignment[rank]):
        local[slot] = job + 1
    rows, totals = sharding.exchange_results(local, assignment, len(lengths))
    ok = bool(torch.equal(rows.sum(dim=0), totals)) and [int(rows[j, 0, 0]) for j in range(len(lengths))] == [1, 2, 3, 4, 5]
    dist.barrier()
    if rank == 0:
        print(json.dumps({"n_gpus": world, "ok": ok, "local_rank": os.environ["LOCAL_RANK"],
                          "master": os.environ["MASTER_ADDR"]}))
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
