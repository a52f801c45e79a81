cd /root/repo
for v in "1 1" "0 1" "1 0" "0 0" "1 1"; do set -- $v
echo "WPREP=$1 WGRAD_AFTER_DGRAD=$2"
DGCNN_WPREP=$1 DGCNN_WGRAD_AFTER_DGRAD=$2 python -m pytest tests/test_gpu_baseline_sizes.py -q -s -k test_config1_full_size_training_step 2>&1 | grep "relative Frobenius\|HIP vs fp32\|passed\|failed"
done
