#!/bin/bash
# the whole -m gpu suite, log under gpurun_out/r05_gputest_$1.log
cd $GRAFT_REPO_ROOT
timeout ${2:-2400} python -m pytest tests -q -m gpu --durations=8 > gpurun_out/r05_gputest_$1.log 2>&1
echo "rc=$?" >> gpurun_out/r05_gputest_$1.log
tail -40 gpurun_out/r05_gputest_$1.log
