import os, sys
sys.path.insert(0, "/root/repo/dynamic-gcnn_amd")
import torch
from dgcnn import _engine as E
H = E.H
B, N, k = 24, 2048, 20
R = B * N
def timeit(fn, n=40, warm=5):
    for _ in range(warm): fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3
for F in (64, 128):
    UV = torch.randn(R, 2 * F, device="cuda")
    idx = torch.randint(0, N, (B, N, k), device="cuda", dtype=torch.int32)
    # a k-NN-like graph: neighbours near the point (spatial locality of real graphs)
    base = torch.arange(N, device="cuda", dtype=torch.int32)[None, :, None]
    idx_local = ((base + torch.randint(-32, 33, (B, N, k), device="cuda", dtype=torch.int32)) % N).to(torch.int32).contiguous()
    mean, rstd, beta = torch.zeros(F, device="cuda"), torch.ones(F, device="cuda"), torch.zeros(F, device="cuda")
    mm = torch.empty(R, 2 * F, device="cuda"); cnt = torch.empty(R, F, device="cuda")
    mxd = torch.empty(R, F, device="cuda"); mnd = torch.empty(R, F, device="cuda")
    U, V = UV[:, :F], UV[:, F:]
    Ud = torch.randn(R, F, device="cuda"); Vd = torch.randn(R, F, device="cuda")
    def run(ix, Vt, Ut, ld, mo, ldm, meo, ldme, co):
        return timeit(lambda: H.call("dgcnn_edge_bn_act_kreduce_f32", Vt.data_ptr(), ld, Ut.data_ptr(), ld, ix.data_ptr(), B, N, k, F,
                                     mean.data_ptr(), rstd.data_ptr(), beta.data_ptr(), 1, mo.data_ptr(), ldm, 0 if meo is None else meo.data_ptr(), ldme, 0 if co is None else co.data_ptr()))
    print("F=%d random idx  all outputs (as in the model)        %6.1f us" % (F, run(idx, V, U, 2 * F, mm, 2 * F, mm[:, F:], 2 * F, cnt)))
    print("F=%d random idx  max only                             %6.1f us" % (F, run(idx, V, U, 2 * F, mm, 2 * F, None, 0, None)))
    print("F=%d random idx  dense V/U (ld = F), dense outputs     %6.1f us" % (F, run(idx, Vd, Ud, F, mxd, F, mnd, F, cnt)))
    print("F=%d local idx   all outputs                          %6.1f us" % (F, run(idx_local, V, U, 2 * F, mm, 2 * F, mm[:, F:], 2 * F, cnt)))
    st = torch.zeros(H.STAT_SLOTS * 2 * F, dtype=torch.float64, device="cuda")
    t0 = timeit(lambda: H.call("dgcnn_edge_gather_add_f32", V.data_ptr(), 2 * F, U.data_ptr(), 2 * F, idx_local.data_ptr(), B, N, k, F, 0, st.data_ptr()))
    print("F=%d local idx   statistics pass                      %6.1f us" % (F, t0))
