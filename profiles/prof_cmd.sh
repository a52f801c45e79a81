#!/bin/bash
# Runs ON THE GPU BOX: rocprofv3 kernel-trace + stats of one command, summary to stdout and to gpurun_out/<tag>_kernel_stats.csv
#   usage: profiles/prof_cmd.sh TAG <command ...>
TAG=$1; shift
OUT=/tmp/prof_$TAG
mkdir -p $GRAFT_REPO_ROOT/gpurun_out
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- "$@" > $OUT.log 2>&1 </dev/null
f=$(find $OUT -name "*kernel_stats.csv" | head -1)
if [ -n "$f" ]; then cp $f $GRAFT_REPO_ROOT/gpurun_out/${TAG}_kernel_stats.csv; head -${LINES_MAX:-25} $f | cut -c1-200; else tail -5 $OUT.log; fi
