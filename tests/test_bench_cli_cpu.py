"""bench.py at its edges, without a GPU: what it refuses and how loudly it fails.

The driver reads ONE JSON line from bench.py. A run that cannot measure the hot path — no gfx950 device, a rank count the launcher did
not provide, a mode that is not run over node-range shards — must end non-zero with no result line, never with a line measured on some
other path (missing #1: `--mode churn --gpus 8` used to run replicas under a sharded label)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def bench(*args, env=None):
    e = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE"):
        e.pop(k, None)
    e.update(env or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *args], cwd=ROOT, env=e, capture_output=True, text=True, timeout=600)


def result_lines(stdout):
    out = []
    for line in stdout.splitlines():
        line = line.strip()
        if line.startswith("{"):
            try:
                out.append(json.loads(line))
            except ValueError:
                pass
    return out


def test_more_gpus_than_ranks_is_refused():
    r = bench("--gpus", "2")
    assert r.returncode == 2
    assert "torch.distributed.run" in r.stderr
    assert result_lines(r.stdout) == []


@pytest.mark.parametrize("mode", ["grouped", "enforce"])
def test_modes_that_are_not_sharded_here_are_refused_over_shards(mode):
    r = bench("--mode", mode, "--shards", "2")
    assert r.returncode == 2
    assert "not run over node-range shards" in r.stderr
    assert result_lines(r.stdout) == []


@pytest.mark.parametrize("args", [(), ("--mode", "churn", "--rounds", "1"), ("--mode", "churn", "--shards", "4", "--rounds", "1"), ("--shards", "2",),
                                  ("--mode", "grouped",)])
def test_no_device_no_result_line(args):
    """No gfx950 device here: every mode dies in the engine's constructor (swp_engine_create -> SWP_ENODEVICE). Nothing falls back to the
    oracle or to the host: there is no line to read."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present: this is the CPU container's test")
    r = bench("--steps", "1", "--warmup", "0", *args)
    assert r.returncode != 0
    assert "no HIP device" in r.stderr
    assert result_lines(r.stdout) == []
