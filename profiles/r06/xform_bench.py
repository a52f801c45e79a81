# NOTE: needs profiles/patches/a_operand_transform_experiment.patch applied (dgcnn_gemm_x3_test_xform exists only in the experiment build).
import os, sys, ctypes
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "dynamic-gcnn_amd"))
import torch
from dgcnn import _engine as E, _hip as H
lib = H.load()
lib.dgcnn_gemm_x3_test_xform.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3
R, K, N = 49152, 1728, 512
x = torch.randn(R, K, device="cuda"); W = torch.randn(K, N, device="cuda") * 0.05
dT = torch.randn(R, N, device="cuda"); out = torch.empty(R, N, device="cuda"); dW = torch.zeros(K, N, device="cuda")
cv = torch.zeros(K, device="cuda"); fv = torch.full((K,), -float("inf"), device="cuda")
cv[704:] = 0.1; fv[704:] = 0.0
for _ in range(30):                      # warm the clocks / power state
    E.gemm(x, W, out)
for rep in range(5):
    lib.dgcnn_gemm_x3_test_xform(None, None)
    f0 = timeit(lambda: E.gemm(x, W, out), 40)
    lib.dgcnn_gemm_x3_test_xform(cv.data_ptr(), fv.data_ptr())
    f1 = timeit(lambda: E.gemm(x, W, out), 40)
    lib.dgcnn_gemm_x3_test_xform(None, None)
    w0 = timeit(lambda: E.gemm(x, dT, dW, transA=True, beta=1.0), 40)
    lib.dgcnn_gemm_x3_test_xform(cv.data_ptr(), fv.data_ptr())
    w1 = timeit(lambda: E.gemm(x, dT, dW, transA=True, beta=1.0), 40)
    print("FC0 forward %.1f -> %.1f us with the A transform; weight gradient %.1f -> %.1f us" % (f0, f1, w0, w1))
lib.dgcnn_gemm_x3_test_xform(None, None)
# what it would replace: the BatchNorm pass over MergedEdgeConv's (R, 1024) output
T = torch.randn(R, 1024, device="cuda"); o = torch.empty(R, 1024, device="cuda")
m = torch.zeros(1024, device="cuda"); r = torch.ones(1024, device="cuda"); be = torch.zeros(1024, device="cuda")
print("bn1_act (R,1024): %.1f us" % timeit(lambda: H.call("dgcnn_bn_act_kreduce_f32", T.data_ptr(), R, 1, 1024, m.data_ptr(), r.data_ptr(), be.data_ptr(), 1, o.data_ptr(), 1024, 0, 0, 0, 0, 0)))
