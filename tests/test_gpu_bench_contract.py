"""bench.py's output contract (the driver parses exactly one JSON line from rank 0): keys, types and the two extra objects."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_prints_one_json_line_with_the_contract_keys():
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "3", "--warmup", "2", "--no-cpu-baseline"],
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600, cwd=ROOT)
    assert p.returncode == 0, p.stderr.decode(errors="replace")[-2000:]
    lines = [l for l in p.stdout.decode().splitlines() if l.strip()]
    assert len(lines) == 1, lines
    d = json.loads(lines[0])
    for key, typ in (("metric", str), ("value", float), ("unit", str), ("n_gpus", int), ("steps", int), ("warmup", int),
                     ("ms_per_step", float), ("higher_is_better", bool), ("scaling", str), ("dtype", str), ("data", str),
                     ("config", dict), ("roofline", dict)):
        assert isinstance(d[key], typ), (key, d[key])
    assert "vs_baseline" in d and d["vs_baseline"] is None           # BASELINE.md holds no published number for this metric
    assert d["n_gpus"] == 1 and d["steps"] == 3 and d["warmup"] == 2 and d["scaling"] == "weak" and d["higher_is_better"] is True
    assert d["unit"] == "clouds/s" and d["dtype"] == "f32" and d["data"] == "synthetic" and "workload" in d["config"]
    assert abs(d["value"] - 24 * 1e3 / d["ms_per_step"]) / d["value"] < 1e-3          # whole-job clouds/s == batch / step time
    r = d["roofline"]
    for key in ("bound", "achieved", "peak", "unit", "frac", "traffic", "kernel"):
        assert key in r, key
    assert r["bound"] in ("hbm", "mfma", "valu") and r["unit"] in ("GB/s", "TFLOP/s")
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3 and 0.0 < r["frac"] < 1.0
    assert r["traffic"] is None or r["traffic"] > 0
    c = d["config"]
    assert c["deterministic"] is True                                 # the default (and benched) mode is the bit-reproducible one
    assert c["launch_mode"] in ("launch-plan replay", "eager") and set(c["launch_mode_calibration"]) >= {"eager_ms", "plan_ms"}
    assert c["repeats"] >= 5 and len(c["per_repeat_ms_per_step"]) == c["repeats"]
    assert sorted(c["per_repeat_ms_per_step"])[c["repeats"] // 2] == d["ms_per_step"]       # the MEDIAN region is the reported one
    if c["launch_mode"] == "launch-plan replay":
        lp = c["launches_per_step"]
        assert 60 < lp["kernels"] < 160 and lp["collectives"] == 0 and c["host_enqueue_ms_per_step"] < 0.5 * d["ms_per_step"]
    s = d["edgeconv_stack"]                                           # the EdgeConv stack alone
    assert s["unit"] == "clouds/s" and s["value"] > d["value"]       # the stack alone is faster than the whole model


def test_bench_gpus_2_from_a_bare_shell_launches_two_ranks_itself():
    """`python bench.py --gpus 2 ...` with no torchrun environment re-launches itself as 2 ranks (bench.self_launch); on the
    one-GPU box both ranks share cuda:0 and the collective runs over gloo (RCCL refuses two ranks on one device).  Checks the
    contract of the N > 1 line and that the replica-checksum guard ran (the run would exit non-zero on diverged replicas)."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--same-device", "--backend", "gloo",
                        "--steps", "3", "--warmup", "2", "--no-cpu-baseline"],
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900, cwd=ROOT, env=env)
    assert p.returncode == 0, p.stderr.decode(errors="replace")[-3000:]
    lines = [l for l in p.stdout.decode().splitlines() if l.strip().startswith("{")]
    assert len(lines) == 1, lines
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 3 and d["scaling"] == "weak"
    c = d["config"]
    assert c["parallelism"] == "dp2" and c["global_batch"] == 48
    assert c["collective_backend"] == "gloo" and c["rccl_ranks"] == 0
    assert abs(d["value"] - 48 * 1e3 / d["ms_per_step"]) / d["value"] < 1e-3       # whole-job clouds/s over BOTH ranks
    assert "roofline" in d and "cpu_baseline" not in d and "edgeconv_stack" not in d


@pytest.mark.parametrize("graph", ["0", "plan"])
def test_bench_default_mode_gives_a_reproducible_line(graph):
    """The benched mode is the deterministic one: two runs of the same command end with the same final loss, launched eagerly
    and replayed from a launch plan alike (--repeats fixes the number of steps; its default depends on the box's speed)."""
    outs = []
    for _ in range(2):
        p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "3", "--warmup", "2", "--no-cpu-baseline",
                            "--no-edgeconv-stack", "--graph", graph, "--repeats", "5"],
                           stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600, cwd=ROOT)
        assert p.returncode == 0, p.stderr.decode(errors="replace")[-2000:]
        d = json.loads([l for l in p.stdout.decode().splitlines() if l.strip().startswith("{")][0])
        want = {"0": "eager", "plan": "launch-plan replay"}[graph]
        assert d["config"]["deterministic"] is True and d["config"]["launch_mode"] == want and "edgeconv_stack" not in d
        assert d["config"]["repeats"] == 5
        outs.append(d["config"]["final_loss"])
    assert outs[0] == outs[1], outs


def test_bench_atomics_mode_is_reachable():
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "3", "--warmup", "2", "--no-cpu-baseline",
                        "--no-edgeconv-stack", "--atomics", "--repeats", "5"],
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600, cwd=ROOT)
    assert p.returncode == 0, p.stderr.decode(errors="replace")[-2000:]
    d = json.loads([l for l in p.stdout.decode().splitlines() if l.strip().startswith("{")][0])
    assert d["config"]["deterministic"] is False
