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
    s = d["edgeconv_stack"]                                           # the EdgeConv stack alone
    assert s["unit"] == "clouds/s" and s["value"] > d["value"]       # the stack alone is faster than the whole model
