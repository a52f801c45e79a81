#!/bin/bash
cd $GRAFT_REPO_ROOT
bash profiles/collect.sh r05 > gpurun_out/r05_collect.log 2>&1
timeout 1500 python -m pytest tests -q -m gpu > gpurun_out/r05_gputest_run6.log 2>&1
echo "rc=$?" >> gpurun_out/r05_gputest_run6.log
tail -3 gpurun_out/r05_gputest_run6.log
python -m pytest tests -x -q -m gpu 2>&1 | tail -2
