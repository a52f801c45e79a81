cd /tmp && export TMPDIR=/tmp
O=/root/repo/gpurun_out/trace_f16b; rm -rf $O; mkdir -p $O
DGCNN_HEAD_PLANES=f16 timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O/trace -- python /root/repo/bench.py --no-cpu-baseline --graph 0 --steps 6 --warmup 2 > $O/log 2>&1
python /root/repo/profiles/step_timeline.py $(ls $O/trace/*/*kernel_trace.csv | head -1) 1 --all > $O/timeline.txt
grep -n "" $O/timeline.txt | sed -n 1p; awk '$1>2300' $O/timeline.txt | cut -c1-118 | head -90
