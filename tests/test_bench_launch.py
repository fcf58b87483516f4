"""bench.py --gpus N (N > 1) outside torchrun: the process must start its own ranks BEFORE touching the GPU, relay rank 0's
JSON line and propagate failure (the driver's command shape is `python3 bench.py --gpus N ...`)."""
import json
import os
import subprocess
import sys

import pytest

from blsutil import ROOT

sys.path.insert(0, ROOT)
import bench  # noqa: E402


def test_self_launch_builds_a_torchrun_command_and_relays_the_line(monkeypatch, capsys):
    seen = {}

    class Done:
        returncode = 0
        stdout = "noise from a rank\n" + json.dumps({"metric": "m", "value": 1.0, "n_gpus": 2}) + "\n"

    def fake_run(cmd, env=None, stdout=None, text=None):
        seen["cmd"], seen["env"] = cmd, env
        return Done()

    monkeypatch.setattr(bench.subprocess, "run", fake_run)
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    with pytest.raises(SystemExit) as e:
        bench.main(["--gpus", "2", "--steps", "5", "--warmup", "1"])
    assert e.value.code == 0
    cmd = seen["cmd"]
    assert cmd[:3] == [sys.executable, "-m", "torch.distributed.run"]
    assert "--nproc-per-node" in cmd and cmd[cmd.index("--nproc-per-node") + 1] == "2"
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and "--nnodes=1" in cmd
    assert cmd[-6:] == ["--gpus", "2", "--steps", "5", "--warmup", "1"] and cmd[-7].endswith("bench.py")
    assert seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"
    out = capsys.readouterr()
    assert json.loads(out.out.strip())["n_gpus"] == 2 and "noise from a rank" in out.err


def test_self_launch_propagates_a_failing_rank(monkeypatch):
    class Failed:
        returncode = 1
        stdout = ""

    monkeypatch.setattr(bench.subprocess, "run", lambda *a, **k: Failed())
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    with pytest.raises(SystemExit) as e:
        bench.main(["--gpus", "4"])
    assert e.value.code == 1


def test_rank_refuses_a_mismatched_world(monkeypatch):
    monkeypatch.setenv("WORLD_SIZE", "2")
    with pytest.raises(SystemExit) as e:
        bench.main(["--gpus", "4"])
    assert "WORLD_SIZE" in str(e.value.code)


def test_two_ranks_really_start_and_fail_loudly_without_a_gpu():
    """End to end on this GPU-less container: `python bench.py --gpus 2 --backend gloo` starts two ranks through
    torch.distributed.run; each refuses to run without a device and the parent returns non-zero with no result line."""
    import torch
    if torch.cuda.device_count() > 0:
        pytest.skip("a GPU is present: the real path is covered by the gpu-marked rehearsal")
    env = dict(os.environ); env.pop("WORLD_SIZE", None)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo", "--batch-log2", "10"],
                       env=env, capture_output=True, text=True, timeout=300)
    assert p.returncode != 0
    assert '"metric"' not in p.stdout
    assert "needs a GPU" in p.stderr
