#!/bin/bash
# Multi-rank first contact: run on a box with N >= 2 MI355X.  Never fakes ranks on one device.
#   bash profiles/scale_probe.sh [N ...]            default: 2 4 8, capped at the number of visible GPUs
# For every N it launches `python bench.py --gpus N` (bench.py starts the N ranks itself through torch.distributed.run, one
# rank per GPU, RCCL through the library's own communicator), then checks the line:
#   config.rccl_ranks == N as RCCL itself reports it (ncclCommCount), per-rank ms/step, the exposed all-reduce time per step,
#   clouds/s and the ratio to N x the one-GPU line measured first on the same box.
# On a failed or hung bring-up it names the stage that failed -- dlopen | unique-id TCP | ncclCommInitRank | first collective --
# from the ranks' DGCNN_RCCL_TRACE lines, with the environment the run used.
# Extra arguments for bench.py: $SCALE_PROBE_BENCH_ARGS (tests: "--dry-run");  python: $PYTHON (default python).
set -u
cd "$(dirname "$0")/.."
PY=${PYTHON:-python}
OUT=${SCALE_PROBE_OUT:-gpurun_out/scale_probe}
mkdir -p "$OUT"
export HSA_ENABLE_IPC_MODE_LEGACY=${HSA_ENABLE_IPC_MODE_LEGACY:-0}      # dmabuf IPC: RCCL across processes needs it on this driver
export DGCNN_RCCL_TRACE=1
export DGCNN_RCCL_TIMEOUT=${DGCNN_RCCL_TIMEOUT:-60}
NGPU=${SCALE_PROBE_NGPU:-$($PY -c 'import torch; print(torch.cuda.device_count())' 2>/dev/null || echo 0)}
LIST="${*:-2 4 8}"
echo "scale_probe: $NGPU GPU(s) visible; HSA_ENABLE_IPC_MODE_LEGACY=$HSA_ENABLE_IPC_MODE_LEGACY MASTER_ADDR=${MASTER_ADDR:-127.0.0.1(self-launch)} NCCL_SOCKET_IFNAME=${NCCL_SOCKET_IFNAME:-unset} DGCNN_RCCL_LIB=${DGCNN_RCCL_LIB:-unset}"
FAIL=0
BASE=""
for N in 1 $LIST; do
  if [ "$N" -gt "$NGPU" ] && [ -z "${SCALE_PROBE_NGPU:-}" ]; then echo "N=$N: skipped (only $NGPU GPU(s) here)"; continue; fi
  J=$OUT/n$N.json; E=$OUT/n$N.err
  timeout ${SCALE_PROBE_TIMEOUT:-900} $PY bench.py --gpus $N --steps ${SCALE_PROBE_STEPS:-20} --warmup 5 --no-cpu-baseline --no-edgeconv-stack ${SCALE_PROBE_BENCH_ARGS:-} > $J 2> $E
  RC=$?
  $PY - "$N" "$J" "$E" "$RC" "$BASE" <<'PY'
import json, re, sys
n, jf, ef, rc, base = int(sys.argv[1]), sys.argv[2], sys.argv[3], int(sys.argv[4]), sys.argv[5]
err = open(ef, errors="replace").read()
line = [l for l in open(jf, errors="replace").read().splitlines() if l.startswith("{")]
stages = {}
for m in re.finditer(r"\[dgcnn\.rccl rank (\d+)/(\d+)\] stage: ([^\n(]+)", err):
    stages[int(m.group(1))] = m.group(3).strip()
def diagnose():
    if not stages:
        return "no rank announced a stage: the ranks did not get as far as the communicator (launcher / import / device selection)"
    order = ["dlopen", "unique-id TCP", "ncclCommInitRank", "first collective", "ready"]
    worst = min(stages.values(), key=lambda s: order.index(s) if s in order else -1)
    who = sorted(r for r, s in stages.items() if s == worst)
    missing = sorted(set(range(n)) - set(stages))
    raised = sorted({(int(m.group(1)), m.group(2)) for m in re.finditer(r"RCCL group, rank (\d+) of \d+, stage `([^`]+)`", err)})
    msg = "stage `%s` did not complete on rank(s) %s" % (worst, who)
    if raised:
        msg += "; raised an error: " + ", ".join("rank %d in `%s`" % rs for rs in raised) + " (ranks still waiting at that stage were stopped by the launcher)"
    else:
        msg += "; no rank raised an error: a hang inside that stage (the probe's timeout ended the run)"
    if missing:
        msg += "; rank(s) %s never reached the communicator" % missing
    return msg
if rc != 0 or not line:
    print("N=%d: FAILED (exit code %d%s): %s" % (n, rc, ", timeout" if rc == 124 else "", diagnose()))
    tail = [l for l in err.splitlines() if l.strip()][-6:]
    print("        last stderr lines:\n          " + "\n          ".join(tail))
    sys.exit(2)
d = json.loads(line[-1]); c = d["config"]
ok = c.get("rccl_ranks") == n
ar = (c.get("allreduce") or {}).get("exposed_ms_per_step") or {}
pr = c.get("per_rank_ms_per_step", {})
eff = ""
if base and d.get("value"):
    eff = "  = %.3f of %d x the one-GPU line" % (d["value"] / (n * float(base)), n)
print("N=%d: %s  rccl_ranks=%s (%s)  backend=%s  %s clouds/s  %.3f ms/step  per-rank ms %s  all-reduce exposed median %s ms%s"
      % (n, "ok" if ok else "RANK COUNT MISMATCH", c.get("rccl_ranks"), c.get("rccl_ranks_source"), c.get("collective_backend"),
         d.get("value"), d["ms_per_step"], pr.get("all"), ar.get("median"), eff))
if "torch.distributed" in str(c.get("collective_backend")) or "own RCCL communicator unavailable" in err:
    print("        NOTE: the library's own communicator did not come up (%s); the run fell back to torch.distributed" % diagnose())
sys.exit(0 if ok else 3)
PY
  R=$?
  [ $R -ne 0 ] && FAIL=1
  if [ "$N" = "1" ] && [ $R -eq 0 ]; then BASE=$($PY -c "import json,sys; print(json.loads([l for l in open('$J') if l.startswith('{')][-1])['value'] or '')"); fi
done
exit $FAIL
