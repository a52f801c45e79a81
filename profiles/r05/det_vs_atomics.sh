#!/bin/bash
# kernel-trace of the step in the deterministic (default) and atomics modes, eager launches: where the +3-4 % sit
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r05_det; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for m in det atomics; do
  fl=""; [ $m = atomics ] && fl="--atomics"
  rocprofv3 --kernel-trace --output-format csv -d $O/$m -- python $GRAFT_REPO_ROOT/bench.py --steps 8 --warmup 2 --repeats 1 --graph 0 --no-cpu-baseline --no-edgeconv-stack $fl > $O/$m.log 2>&1
  f=$(ls -t $O/$m/*/*kernel_trace.csv | head -1)
  python $GRAFT_REPO_ROOT/profiles/trace_summary.py $f 10 > $O/$m.txt
  python $GRAFT_REPO_ROOT/profiles/step_timeline.py $f 2 --all > $O/${m}_timeline.txt
done
head -60 $O/det.txt
