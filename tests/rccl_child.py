"""Child of tests/test_gpu_rccl.py: a ONE-rank process group on the given backend, exchange_results with the
world == 1 shortcut disabled, so that all_gather_into_tensor / all_reduce on int64 tensors and the second-stream
ordering run through the backend (nccl = RCCL on the GPU box).  Prints one JSON line."""
import json
import os
import sys

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pymasc_amd import sharding  # noqa: E402


def main():
    backend = sys.argv[1]
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", sys.argv[2] if len(sys.argv) > 2 else "29561")
    if backend == "nccl":
        device = torch.device("cuda", 0)
        torch.cuda.set_device(device)
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=device)
    else:
        device = torch.device("cpu")
        dist.init_process_group(backend, rank=0, world_size=1)
    g = torch.Generator().manual_seed(5)
    njobs, nrows, stride = 7, 6, 1001
    assignment = sharding.lpt_assign([float(njobs - i) for i in range(njobs)], 1)
    ok = True
    if backend == "nccl":
        kstream, xstream = torch.cuda.Stream(device), torch.cuda.Stream(device)
        bufs = [torch.zeros((njobs, nrows, stride), dtype=torch.int64, device=device) for _ in range(2)]
        done = [torch.cuda.Event() for _ in range(2)]
        free = [torch.cuda.Event() for _ in range(2)]
        outs = []
        for step in range(6):                      # double-buffered like bench.py: producer stream / exchange stream
            b = step & 1
            ref = torch.randint(0, 2**40, (njobs, nrows, stride), generator=g, dtype=torch.int64)
            with torch.cuda.stream(kstream):
                if step >= 2:
                    kstream.wait_event(free[b])
                bufs[b].copy_(ref.to(device, non_blocking=False))
                # a long-running producer after the copy would hide a missing wait; an in-place op keeps the dependency real
                bufs[b].add_(step)
                done[b].record(kstream)
            with torch.cuda.stream(xstream):
                xstream.wait_event(done[b])
                rows, totals = sharding.exchange_results(bufs[b], assignment, njobs, force_collectives=True)
                free[b].record(xstream)
            outs.append((ref + step, rows, totals))
        torch.cuda.synchronize(device)
        for ref, rows, totals in outs:
            order = torch.tensor(assignment[0])
            want = torch.empty_like(ref)
            want[order] = ref                      # slot s holds job assignment[0][s]
            ok = ok and torch.equal(rows.cpu(), want) and torch.equal(totals.cpu(), ref.sum(0))
    else:
        ref = torch.randint(0, 2**40, (njobs, nrows, stride), generator=g, dtype=torch.int64)
        rows, totals = sharding.exchange_results(ref, assignment, njobs, force_collectives=True)
        order = torch.tensor(assignment[0])
        want = torch.empty_like(ref)
        want[order] = ref
        ok = torch.equal(rows, want) and torch.equal(totals, ref.sum(0))
    print(json.dumps({"ok": bool(ok), "backend": str(dist.get_backend()), "world": dist.get_world_size()}))
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
