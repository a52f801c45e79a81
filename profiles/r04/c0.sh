#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 200 python profiles/config_sweep.py --only 'configs[0]' 2>&1 | grep configs
timeout 200 python profiles/config_sweep.py --only 'configs[0]' --graph 2>&1 | grep configs
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/c0 -- python $GRAFT_REPO_ROOT/profiles/config_sweep.py --only 'configs[0]' > /tmp/c0.log 2>&1 </dev/null
f=$(find /tmp/c0 -name "*kernel_trace.csv" | head -1)
[ -n "$f" ] && python $GRAFT_REPO_ROOT/profiles/step_timeline.py $f 2 --all > $GRAFT_REPO_ROOT/gpurun_out/c0_timeline.txt 2>&1 </dev/null
head -3 $GRAFT_REPO_ROOT/gpurun_out/c0_timeline.txt
