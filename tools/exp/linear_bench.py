"""Tall-skinny fp32 Linear (yolat_linear_fwd) at the training step's shapes: M rows x K -> N, with / without the BatchNorm
prologue on A and the statistics epilogue.  HIP-event times; YOLAT_GEMM_ABL ablations only with the temporary patch."""
import os, sys, torch
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)
import yolat_vectorgraphicsrecognition_amd as yv
from yolat_vectorgraphicsrecognition_amd import ops
g = torch.Generator().manual_seed(0)
def bench(M, K, N, pro, stats):
    A = torch.randn(M, K, generator=g).cuda(); W = torch.randn(N, K, generator=g).cuda(); b = torch.randn(N, generator=g).cuda()
    Y = torch.empty(M, N).cuda()
    sc, sh = torch.rand(K, generator=g).cuda(), torch.randn(K, generator=g).cuda()
    st = torch.empty(int(ops.lib.yolat_bn_stats_elems(M, N)), device="cuda") if stats else None
    def run():
        ops.linear_fwd(A, W, b, Y, a_pro=(sc, sh) if pro else None, a_relu=pro, stats=st)
    for _ in range(3): run()
    torch.cuda.synchronize()
    Ad = A.double()
    if pro:
        Ad = torch.relu(Ad * sc.double() + sh.double())
    want = Ad @ W.double().T + b.double()
    err = float((Y.double() - want).abs().max() / want.abs().max())
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): run()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 20 * 1e3
    mb = (M * K + M * N) * 4 / 1e6
    print("rows_gemm=%s M=%d K=%d N=%d pro=%d stats=%d: %.1f us  (%.0f MB -> %.2f TB/s)  max err %.1e"
          % (os.environ.get("YOLAT_ROWS_GEMM", "1"), M, K, N, pro, stats, us, mb, mb / us, err))
for M, K, N, pro, stats in ((174512, 64, 64, 0, 0), (174512, 64, 128, 0, 0), (174512, 64, 128, 1, 1), (212511, 64, 64, 1, 1)):
    bench(M, K, N, pro, stats)
