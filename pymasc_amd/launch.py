"""One process per GPU, started from a plain `python script.py --gpus N`.

The reference's `-p N` starts its own worker processes (PyMaSC/handler/calc.py:163-192: N x CalcWorker, one
calculator each, `spawn` start method PyMaSC/__init__.py:40-53).  The MI355X equivalent is one rank per GPU under
`torch.distributed`; `spawn_ranks` is the part of `calc.py:176-192` that creates them: N fresh child processes of
the same script with RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT set, rank 0's stdout relayed, any
failing rank ends the run (the reference's '__ERROR__' report + pool teardown, worker.py:91-99, utils/calc.py:132-145).

The parent NEVER touches the GPU (no HIP call, not even torch.cuda.is_available()): children are started as new
processes, nothing is exec'd from a process that has initialised a device.
"""
from __future__ import annotations

import os
import socket
import subprocess
import sys
import time
from typing import List, Optional, Sequence


def needs_spawn(requested: int, environ=None) -> bool:
    """True when N > 1 ranks were asked for and no launcher (torchrun) has set up the ranks already."""
    env = os.environ if environ is None else environ
    return requested > 1 and "WORLD_SIZE" not in env


def free_port() -> int:
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def spawn_ranks(argv: Sequence[str], nranks: int, extra_env: Optional[dict] = None, timeout: Optional[float] = None,
                poll: float = 0.05) -> int:
    """Runs `argv` once per rank (fresh processes), waits for all of them, returns the exit code of the run:
    0 if every rank exited 0, else the first non-zero code (the remaining ranks are terminated by PID).
    Rank 0 inherits stdout (it prints the result line); every rank inherits stderr."""
    port = free_port()
    procs: List[subprocess.Popen] = []
    for r in range(nranks):
        env = dict(os.environ)
        env.update({"RANK": str(r), "LOCAL_RANK": str(r), "WORLD_SIZE": str(nranks), "LOCAL_WORLD_SIZE": str(nranks),
                    "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port)})
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")     # dmabuf IPC only on this pool (RCCL needs it)
        if extra_env:
            env.update(extra_env)
        procs.append(subprocess.Popen(list(argv), env=env, stdout=None if r == 0 else subprocess.DEVNULL))
    t0 = time.monotonic()
    rc = 0
    live = set(range(nranks))

    def stop(ranks):
        """terminate, then kill, exactly the processes started above (PIDs, never a pattern)"""
        for r in ranks:
            if procs[r].poll() is None:
                procs[r].terminate()
        deadline = time.monotonic() + 10
        for r in ranks:
            try:
                procs[r].wait(max(0.1, deadline - time.monotonic()))
            except subprocess.TimeoutExpired:
                procs[r].kill()
                procs[r].wait()

    try:
        while live:
            for r in sorted(live):
                code = procs[r].poll()
                if code is None:
                    continue
                live.discard(r)
                if code != 0 and rc == 0:
                    rc = code
                    print(f"[launch] rank {r} exited with code {code}; stopping the other ranks", file=sys.stderr)
            if rc != 0 or (timeout is not None and time.monotonic() - t0 > timeout):
                if rc == 0:
                    rc = 124
                    print(f"[launch] timeout after {timeout:.0f}s; stopping all ranks", file=sys.stderr)
                stop(sorted(live))
                live.clear()
                break
            if live:
                time.sleep(poll)
    finally:
        # KeyboardInterrupt or any other exception in the loop above: no rank may outlive the launcher holding a GPU and
        # the rendezvous port
        stop([r for r in range(nranks) if procs[r].poll() is None])
    return rc
