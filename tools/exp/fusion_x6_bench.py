"""Time yolat_fusion_pair_eval_x6 built with -DFX_ABLATE=k (tools/exp/_fx/fx_k.so) on cfg-sized synthetic inputs.
usage: python tools/exp/fusion_x6_bench.py N P [k ...]"""
import ctypes
import os
import sys

import torch

here = os.path.dirname(os.path.abspath(__file__))
N, P = int(sys.argv[1]), int(sys.argv[2])
ks = sys.argv[3:] or ["0"]
D, F = 128, 1024
ZW = 2 * F + D
gen = torch.Generator().manual_seed(0)
A = torch.randn(N, D, generator=gen).cuda()
S = torch.randn(P, D, generator=gen).cuda()
seg = (torch.arange(N) * P // N).int().cuda()
W = [torch.randn(F * D, generator=gen).cuda().bfloat16() for _ in range(6)]
t = [torch.randn(F, generator=gen).cuda() for _ in range(2)]
Z = torch.zeros(P, ZW).cuda()
c_p, c_i = ctypes.c_void_p, ctypes.c_int64
for k in ks:
    lib = ctypes.CDLL(os.path.join(here, "_fx", "fx_%s.so" % k))
    fn = lib.yolat_fusion_pair_eval_x6
    fn.argtypes = [c_p, c_i, c_i, c_i, c_p, c_p, c_p, c_p, c_i, c_p, c_p, c_i, c_p, c_i, c_i, c_p, c_p, c_p, c_p, c_p, c_i, c_p]
    st = torch.cuda.current_stream().cuda_stream

    def run():
        rc = fn(A.data_ptr(), D, N, D, W[0].data_ptr(), W[1].data_ptr(), W[2].data_ptr(), t[0].data_ptr(), F, seg.data_ptr(),
                Z.data_ptr(), ZW, S.data_ptr(), D, P, W[3].data_ptr(), W[4].data_ptr(), W[5].data_ptr(), t[1].data_ptr(),
                Z[:, F + D:].data_ptr(), ZW, st)
        assert rc == 0, rc

    for _ in range(3):
        run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        run()
    e1.record()
    torch.cuda.synchronize()
    print("N %d P %d ablate %s WGS %s: %.1f us" % (N, P, k, os.environ.get("YOLAT_FUSION_X6_WGS", "-"), e0.elapsed_time(e1) * 50))
