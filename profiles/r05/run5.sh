#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05_run5; mkdir -p $O
timeout 2400 python -m pytest tests -q -m gpu --durations=5 > $O/tests.log 2>&1
echo "tests rc=$?" >> $O/tests.log
tail -12 $O/tests.log
bash profiles/r05/ab.sh "" "--atomics" 2
