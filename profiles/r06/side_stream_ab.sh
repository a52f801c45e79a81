#!/bin/bash
# Runs ON THE GPU BOX: A/B of how the side stream (weight-gradient GEMMs, CSR build, weight preparation) is created.
#   base      torch.cuda.Stream (normal priority, all CUs)                       -- rounds 1-5
#   lowprio   hipStreamCreateWithPriority(least)
#   cuN       hipExtStreamCreateWithCUMask leaving N compute units to the main stream
#   mainhigh  the step's own stream at the device's greatest priority (side stream normal)
# Alternating runs on one box (boxes differ by +-3 %); prints ms_per_step of every run.
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/side_stream_ab.txt
: > $OUT
run() {  # name, env...
  local name=$1; shift
  local ms=$(env "$@" python bench.py --no-cpu-baseline --no-edgeconv-stack --graph plan --repeats 9 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], min(d['config']['per_repeat_ms_per_step']))")
  echo "$name $ms" | tee -a $OUT
}
for rep in 1 2; do
  run base X=0
  run lowprio DGCNN_SIDE_LOW_PRIORITY=1
  run cu8 DGCNN_SIDE_RESERVE_CUS=8
  run cu16 DGCNN_SIDE_RESERVE_CUS=16
  run cu32 DGCNN_SIDE_RESERVE_CUS=32
  run cu64 DGCNN_SIDE_RESERVE_CUS=64
  run mainhigh DGCNN_BENCH_MAIN_PRIORITY=-1
  run mainhigh_cu16 DGCNN_BENCH_MAIN_PRIORITY=-1 DGCNN_SIDE_RESERVE_CUS=16
done
python - <<'PY'
import ctypes, torch, sys
sys.path.insert(0, "dynamic-gcnn_amd")
from dgcnn import _hip as H
a, b = ctypes.c_int(), ctypes.c_int()
H.load().dgcnn_stream_priority_range(ctypes.byref(a), ctypes.byref(b))
print("hipDeviceGetStreamPriorityRange: least %d greatest %d; CUs %d" % (a.value, b.value, torch.cuda.get_device_properties(0).multi_processor_count))
PY
