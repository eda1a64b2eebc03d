"""`python bench.py --gpus N` starts its N ranks itself (pymasc_amd/launch.py; the reference's `-p N` spawns its
workers the same way, handler/calc.py:163-192).  CPU: the launcher end to end with gloo ranks, failure propagation,
and bench.py's own launcher path up to the point where every rank refuses to run without a GPU."""
import json
import os
import subprocess
import sys
import time

import pytest

from pymasc_amd import launch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CHILD = os.path.join(HERE, "launch_child.py")


def _spawn(args, n, timeout=300):
    code = ("import sys; sys.path.insert(0, %r); from pymasc_amd import launch; "
            "sys.exit(launch.spawn_ranks(%r, %d, timeout=240))" % (ROOT, [sys.executable, CHILD] + args, n))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    return subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=timeout, env=env)


def test_needs_spawn():
    assert launch.needs_spawn(2, {}) and launch.needs_spawn(8, {"RANK": "0"})
    assert not launch.needs_spawn(1, {})
    assert not launch.needs_spawn(4, {"WORLD_SIZE": "4"})        # torchrun (the driver's N > 1 command) already did it


@pytest.mark.timeout(600)
@pytest.mark.parametrize("n", [2, 3])
def test_launcher_runs_n_gloo_ranks_end_to_end(n):
    p = _spawn([], n)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout                                # ONE line, from rank 0 only
    d = json.loads(lines[0])
    assert d == {"n_gpus": n, "ok": True, "local_rank": "0", "master": "127.0.0.1"}


@pytest.mark.timeout(600)
def test_a_failing_rank_ends_the_run():
    t0 = time.monotonic()
    p = _spawn(["fail"], 2)
    assert p.returncode == 3, (p.returncode, p.stderr[-2000:])
    assert "rank 1 exited with code 3" in p.stderr
    assert time.monotonic() - t0 < 200                              # the surviving rank was stopped, not waited for


@pytest.mark.timeout(600)
def test_bench_gpus_n_spawns_ranks_before_touching_the_gpu():
    """No GPU here: both ranks of `bench.py --gpus 2` must start (fresh processes with RANK / WORLD_SIZE set) and fail
    loudly; the parent never imports torch.cuda itself and returns their exit code."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is visible: the ranks would run the benchmark")
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       capture_output=True, text=True, timeout=500, env=env)
    assert p.returncode != 0
    assert p.stderr.count("no GPU visible") >= 1 and "[launch] rank" in p.stderr
    assert p.stdout.strip() == ""


def test_an_exception_in_the_launcher_leaves_no_rank_behind(monkeypatch):
    """KeyboardInterrupt (or any exception) while waiting: every rank process is terminated before it propagates."""
    from pymasc_amd import launch
    started = []
    real_popen = subprocess.Popen

    def recording_popen(*a, **k):
        p = real_popen(*a, **k)
        started.append(p)
        return p

    real_sleep = launch.time.sleep
    calls = []

    def interrupt(seconds):          # (launch.time IS the time module: only the launcher's first poll is interrupted)
        calls.append(seconds)
        if len(calls) == 1:
            raise KeyboardInterrupt
        real_sleep(seconds)

    monkeypatch.setattr(launch.subprocess, "Popen", recording_popen)
    monkeypatch.setattr(launch.time, "sleep", interrupt)
    with pytest.raises(KeyboardInterrupt):
        launch.spawn_ranks([sys.executable, "-c", "import time; time.sleep(120)"], 2)
    assert len(started) == 2 and all(p.poll() is not None for p in started)


def test_a_hanging_rank_hits_the_timeout():
    from pymasc_amd import launch
    rc = launch.spawn_ranks([sys.executable, "-c", "import time; time.sleep(120)"], 2, timeout=1.0)
    assert rc == 124
