"""bench.py's launcher logic on CPU (no kernels run): `--gpus N` without WORLD_SIZE starts its own ranks through torch.distributed.run
with the driver's documented arguments; with WORLD_SIZE set (the driver's own torchrun form) it does not."""
import importlib.util
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench():
    spec = importlib.util.spec_from_file_location("bench_under_test", os.path.join(ROOT, "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_gpus_n_without_a_launcher_starts_its_own_ranks(monkeypatch):
    bench = _bench()
    seen = {}

    def fake_call(cmd, env=None):
        seen["cmd"], seen["env"] = cmd, env
        return 0

    import subprocess
    monkeypatch.setattr(subprocess, "call", fake_call)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        monkeypatch.delenv(k, raising=False)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4", "--steps", "3", "--warmup", "1", "--config", "cr"])
    with pytest.raises(SystemExit) as e:
        bench.main()
    assert e.value.code == 0
    cmd = seen["cmd"]
    assert cmd[:3] == [sys.executable, "-m", "torch.distributed.run"] and "--nnodes=1" in cmd
    assert cmd[cmd.index("--nproc-per-node") + 1] == "4" and cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert 0 < int(cmd[cmd.index("--master-port") + 1]) < 65536
    i = cmd.index(os.path.join(ROOT, "bench.py"))
    assert cmd[i + 1:] == ["--gpus", "4", "--steps", "3", "--warmup", "1", "--config", "cr"]
    assert seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0")


def test_a_failing_rank_fails_the_launcher(monkeypatch):
    bench = _bench()
    import subprocess
    monkeypatch.setattr(subprocess, "call", lambda cmd, env=None: 3)
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "2"])
    with pytest.raises(SystemExit) as e:
        bench.main()
    assert e.value.code == 3
