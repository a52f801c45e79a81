#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05_run7; mkdir -p $O
timeout 2400 python -m pytest tests -x -q -m gpu > $O/tests.log 2>&1
echo "tests rc=$?" >> $O/tests.log
tail -6 $O/tests.log
bash profiles/r05/ab.sh "" "--atomics" 3
