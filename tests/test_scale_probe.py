"""profiles/scale_probe.sh -- the multi-rank first-contact kit -- driven on a CPU-only box: bench.py --dry-run with the stand-in
communicator (tests/stub_rccl.py) in place of RCCL.  Checks what the script promises for a real N-GPU box: one line per N with the
communicator's own rank count, per-rank ms and the exposed all-reduce time; and, on a failed bring-up, the STAGE that failed."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(tmp_path, ns, **extra):
    env = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "DGCNN_BENCH_SELF_LAUNCHED"):
        env.pop(k, None)
    env["PYTHONPATH"] = os.pathsep.join([ROOT, os.path.join(ROOT, "dynamic-gcnn_amd"), os.path.join(ROOT, "tests")] +
                                        [p for p in env.get("PYTHONPATH", "").split(os.pathsep) if p])
    env.update(DGCNN_BENCH_GROUP="stub_rccl:StubGroup", SCALE_PROBE_BENCH_ARGS="--dry-run", SCALE_PROBE_NGPU="8",
               SCALE_PROBE_OUT=str(tmp_path), SCALE_PROBE_STEPS="3", SCALE_PROBE_TIMEOUT="240", DGCNN_RCCL_TIMEOUT="20",
               OMP_NUM_THREADS="1", PYTHON=sys.executable)
    env.update(extra)
    return subprocess.run(["bash", os.path.join(ROOT, "profiles", "scale_probe.sh")] + [str(n) for n in ns], env=env,
                          capture_output=True, text=True, timeout=900)


def test_probe_reports_every_rank_count_from_the_communicator(tmp_path):
    r = _run(tmp_path, [2, 4])
    assert r.returncode == 0, r.stdout + r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("N=")]
    assert [l.split(":")[0] for l in lines] == ["N=1", "N=2", "N=4"], r.stdout
    for n, l in zip((1, 2, 4), lines):
        assert "ok" in l and "rccl_ranks=%d" % n in l and "per-rank ms" in l, l


def test_probe_names_the_stage_that_failed(tmp_path):
    r = _run(tmp_path, [2], STUB_RCCL_FAIL="1:ncclCommInitRank")
    assert r.returncode != 0
    out = r.stdout
    assert "N=2: FAILED" in out and "stage `ncclCommInitRank` did not complete" in out and "rank 1 in `ncclCommInitRank`" in out, out
    r = _run(tmp_path, [2], STUB_RCCL_FAIL="0:dlopen")
    assert "stage `dlopen` did not complete" in r.stdout and "rank 0 in `dlopen`" in r.stdout, r.stdout
