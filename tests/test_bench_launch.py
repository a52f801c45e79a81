"""bench.py start-up logic that needs no GPU: `python bench.py --gpus N` from a bare shell must start N ranks itself."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_self_launch_builds_the_drivers_torchrun_command(monkeypatch):
    sys.path.insert(0, ROOT)
    import importlib
    bench = importlib.import_module("bench")
    seen = {}

    def fake_call(cmd, env=None):
        seen["cmd"], seen["env"] = cmd, env
        return 0
    import subprocess
    monkeypatch.setattr(subprocess, "call", fake_call)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4", "--steps", "7", "--warmup", "2"])
    assert bench.self_launch(4) == 0
    cmd = seen["cmd"]
    assert cmd[:3] == [sys.executable, "-m", "torch.distributed.run"]
    assert "--nnodes=1" in cmd and cmd[cmd.index("--nproc-per-node") + 1] == "4"
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and int(cmd[cmd.index("--master-port") + 1]) > 0
    i = cmd.index(os.path.join(ROOT, "bench.py"))
    assert cmd[i + 1:] == ["--gpus", "4", "--steps", "7", "--warmup", "2"]          # the ranks get the caller's arguments
    assert seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"


def test_a_bare_gpus_n_invocation_takes_the_self_launch_path(monkeypatch):
    sys.path.insert(0, ROOT)
    import importlib
    bench = importlib.import_module("bench")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "DGCNN_BENCH_SELF_LAUNCHED"):
        monkeypatch.delenv(k, raising=False)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "8"])
    monkeypatch.setattr(bench, "self_launch", lambda n: 40 + n)
    try:
        bench.main()
    except SystemExit as e:
        assert e.code == 48
    else:
        raise AssertionError("main() should exit with the launcher's code")
