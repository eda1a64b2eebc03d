"""The RCCL code of the result exchange, executed on the one GPU a test box has: a one-rank `nccl` process group with
the world == 1 shortcut of sharding.exchange_results disabled (all_gather_into_tensor + all_reduce on int64 device
tensors, exchange on a second stream behind an event).  N > 1 ranks over xGMI remain unmeasured here: the driver's
8-GPU runs are the only place they execute."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run_child(backend, port):
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    res = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "rccl_child.py"), backend, str(port)], cwd=ROOT, env=env,
                         capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stderr[-3000:]
    return json.loads([l for l in res.stdout.splitlines() if l.startswith("{")][-1])


@pytest.mark.gpu
def test_exchange_runs_through_rccl_on_one_rank():
    out = run_child("nccl", 29571)
    assert out == {"ok": True, "backend": "nccl", "world": 1}


def test_forced_collectives_on_gloo():
    out = run_child("gloo", 29572)
    assert out == {"ok": True, "backend": "gloo", "world": 1}
