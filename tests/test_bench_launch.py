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


def test_gpus_8_end_to_end_with_a_stand_in_communicator():
    """`python bench.py --gpus 8 --dry-run` from a bare shell, no GPU: self-launch through torch.distributed.run, 8 ranks meet
    over the product's id rendezvous (rank 0 AND rank 5 arrive late), every rank issues the same collectives, the timed region
    is fenced by barriers, rank 0 prints ONE JSON line whose rccl_ranks comes from the communicator and which carries every
    rank's ms/step."""
    import json
    import subprocess
    env = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "DGCNN_BENCH_SELF_LAUNCHED"):
        env.pop(k, None)
    env["PYTHONPATH"] = os.pathsep.join([ROOT, os.path.join(ROOT, "dynamic-gcnn_amd"), os.path.join(ROOT, "tests")] +
                                        [p for p in env.get("PYTHONPATH", "").split(os.pathsep) if p])
    env["DGCNN_BENCH_GROUP"] = "stub_rccl:StubGroup"
    env["STUB_RCCL_DELAY_RANK0"] = "1.0"
    env["STUB_RCCL_DELAY_RANK5"] = "2.0"
    env["OMP_NUM_THREADS"] = "1"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--steps", "4", "--warmup", "1", "--dry-run"],
                       env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    out = json.loads(lines[0])
    assert out["dry_run"] is True and out["n_gpus"] == 8 and out["steps"] == 4 and out["value"] is None
    cfg = out["config"]
    assert cfg["rccl_ranks"] == 8 and cfg["rccl_rank"] == 0 and cfg["parallelism"] == "dp8" and cfg["global_batch"] == 192
    pr = cfg["per_rank_ms_per_step"]
    assert len(pr["all"]) == 8 and pr["min"] <= pr["max"] and abs(out["ms_per_step"] - pr["max"]) < 1e-6


def test_a_collective_that_never_completes_ends_the_run_with_one_error_line():
    """the first multi-rank contact must not be able to hang its caller.  Rank 1's third collective (the first
    one of the first step: bring-up holds two) never completes -- nothing raises anywhere, rank 0 blocks waiting for rank 1's
    contribution.  bench.py's deadline (6 s here, 120 s by default) must end the run: ONE JSON line on stdout naming the stage, a
    non-zero exit code, well inside the test's own timeout."""
    import json
    import subprocess
    import time
    env = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "DGCNN_BENCH_SELF_LAUNCHED"):
        env.pop(k, None)
    env["PYTHONPATH"] = os.pathsep.join([ROOT, os.path.join(ROOT, "dynamic-gcnn_amd"), os.path.join(ROOT, "tests")] +
                                        [p for p in env.get("PYTHONPATH", "").split(os.pathsep) if p])
    env["DGCNN_BENCH_GROUP"] = "stub_rccl:StubGroup"
    env["DGCNN_BENCH_DEADLINE"] = "6"
    env["OMP_NUM_THREADS"] = "1"
    for hang, stage in (("1:1", "first step with the gradient all-reduce"), ("1:0", "communicator bring-up")):
        env["STUB_RCCL_HANG"] = hang
        t0 = time.time()
        r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--dry-run"],
                           env=env, capture_output=True, text=True, timeout=300)
        took = time.time() - t0
        assert r.returncode != 0, (r.stdout, r.stderr[-2000:])
        lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
        assert len(lines) == 1, (r.stdout, r.stderr[-2000:])
        out = json.loads(lines[0])
        assert out["value"] is None and out["n_gpus"] == 2 and stage in out["error"] and "did not finish within" in out["error"], out
        assert took < 120, took
