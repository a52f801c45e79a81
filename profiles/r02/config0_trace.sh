#!/bin/bash
# ON THE GPU BOX: kernel trace of configs[0] (B=2, N=512, k=10, 1 EdgeConv) under HIP-graph replay
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/cfg0_trace
rm -rf $O; mkdir -p $O
cat > /tmp/cfg0.py <<'PY'
import os, sys, time
sys.path.insert(0, os.path.join(os.environ["GRAFT_REPO_ROOT"], "dynamic-gcnn_amd"))
import numpy as np, torch, dgcnn
flags = dgcnn.DGCNN_FLAGS(NUM_CLASS=2, FC_LAYERS=2, FC_FILTERS=[512, 256], TRAIN=True, NUM_CHANNEL=3, MODEL_NAME="dgcnn",
                          EDGE_CONV_LAYERS=1, EDGE_CONV_FILTERS=64, KVALUE=10)
tv = dgcnn.trainval(flags).initialize().use_graph(True)
rng = np.random.default_rng(0)
pts = torch.from_numpy(rng.random((2, 512, 3), dtype=np.float32)).cuda()
lab = torch.from_numpy(rng.integers(0, 2, (2, 512)).astype(np.int32)).cuda()
for _ in range(30):
    tv.zero_gradients(None); tv.accum_gradient(None, [pts], [lab]); tv.apply_gradient(None)
torch.cuda.synchronize()
PY
rocprofv3 --kernel-trace --stats --output-format csv -d $O/t -- python /tmp/cfg0.py > $O/log.txt 2>&1
python $R/profiles/trace_summary.py $(find $O/t -name "*kernel_trace.csv") 30 > $O/summary.txt 2>&1
cat $O/summary.txt | cut -c1-130
