#!/bin/bash
cd /tmp && export TMPDIR=/tmp
timeout 500 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/c2 -- python $GRAFT_REPO_ROOT/profiles/config_sweep.py --only 'bf16 edge-MLP' > /tmp/c2.log 2>&1 </dev/null
f=$(find /tmp/c2 -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && head -32 $f | cut -c1-170
