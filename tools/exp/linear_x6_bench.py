"""Time yolat_linear_x6 against yolat_linear_fwd (fp32 MFMA) on classifier-shaped problems.
usage: python tools/exp/linear_x6_bench.py M K N"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from yolat_vectorgraphicsrecognition_amd._lib import lib, check

M, K, N = (int(a) for a in sys.argv[1:4])
gen = torch.Generator().manual_seed(0)
A = torch.randn(M, K, generator=gen).cuda()
W = (torch.randn(N, K, generator=gen) / K ** 0.5).cuda()
b = torch.randn(N, generator=gen).cuda()
s = (torch.rand(N, generator=gen) + 0.5).cuda()
t = torch.randn(N, generator=gen).cuda()
st = torch.cuda.current_stream().cuda_stream
packed = torch.empty(lib.yolat_split_bf16x3_packed_elems(N, K), dtype=torch.bfloat16, device="cuda")
check(lib.yolat_split_bf16x3_packed(W.data_ptr(), K, N, K, s.data_ptr(), packed.data_ptr(), st))
tf = (s * b + t).contiguous()
o1, o2 = torch.empty(M, N).cuda(), torch.empty(M, N).cuda()


def x6():
    check(lib.yolat_linear_x6(A.data_ptr(), K, M, K, packed.data_ptr(), tf.data_ptr(), 1, N, o1.data_ptr(), N, st))


apk = torch.empty(lib.yolat_split_bf16x3_packed_elems(M, K), dtype=torch.bfloat16, device="cuda")
o3 = torch.empty(M, N).cuda()


def x6pre():
    check(lib.yolat_split_bf16x3_packed(A.data_ptr(), K, M, K, None, apk.data_ptr(), st))
    check(lib.yolat_linear_x6_pre(apk.data_ptr(), M, K, packed.data_ptr(), tf.data_ptr(), 1, N, o3.data_ptr(), N, st))


gpk = torch.empty(lib.yolat_gemm_x6_packed_elems(N, K), dtype=torch.bfloat16, device="cuda")
check(lib.yolat_gemm_x6_pack(W.data_ptr(), K, N, K, s.data_ptr(), gpk.data_ptr(), st))
gwork = torch.empty(max(1, lib.yolat_gemm_x6_work_elems(M, N, K)), device="cuda")
o4 = torch.empty(M, N).cuda()


def gx():
    check(lib.yolat_gemm_x6(A.data_ptr(), K, M, K, gpk.data_ptr(), tf.data_ptr(), 1, N, o4.data_ptr(), N, gwork.data_ptr(), st))


def f32():
    check(lib.yolat_linear_fwd(A.data_ptr(), K, M, K, None, None, 0, W.data_ptr(), K, b.data_ptr(), N, s.data_ptr(),
                               t.data_ptr(), 1, o2.data_ptr(), N, 0, None, st))


for name, fn in (("gemm_x6", gx), ("fp32", f32)):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50):
        fn()
    e1.record()
    torch.cuda.synchronize()
    print("%s M %d K %d N %d: %.2f us" % (name, M, K, N, e0.elapsed_time(e1) * 20))
print("max |gemm_x6 - fp32| / max|fp32| = %.2e" % float((o4 - o2).abs().max() / o2.abs().max()))
