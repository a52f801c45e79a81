#!/bin/bash
cd $GRAFT_REPO_ROOT
bash profiles/collect.sh r05 > gpurun_out/r05_collect.log 2>&1
for i in 4 5; do
  timeout 1500 python -m pytest tests -q -m gpu > gpurun_out/r05_gputest_run$i.log 2>&1
  echo "rc=$?" >> gpurun_out/r05_gputest_run$i.log
  tail -3 gpurun_out/r05_gputest_run$i.log
done
python profiles/config_sweep.py > gpurun_out/r05_config_sweep.txt 2>&1
python profiles/config_sweep.py --only "configs[0]" --graph >> gpurun_out/r05_config_sweep.txt 2>&1
cat gpurun_out/r05_config_sweep.txt | grep -v amdgpu
