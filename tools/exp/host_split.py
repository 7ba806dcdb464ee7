import os, sys, time, ctypes, torch
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import yolat_vectorgraphicsrecognition_amd as yv, golden_util as gu, bench
from yolat_vectorgraphicsrecognition_amd.plan import EvalPlan
from yolat_vectorgraphicsrecognition_amd._lib import lib
data, slices, optkw, _ = yv.config("2")
model = gu.fill_state_(yv.SparseCADGCN(yv.Opt(**optkw)), 0).cuda().eval()
bench.to_device(data)
S = 8
streams = [torch.cuda.Stream() for _ in range(S)]
plans = [EvalPlan(model) for _ in range(S)]
P = data.bbox.shape[0]
for k in range(S):
    with torch.cuda.stream(streams[k]), torch.no_grad():
        plans[k].run(data.x, data.edge, data.e_attr, data.bbox_idx, P)
torch.cuda.synchronize()
N, E = data.x.shape[0], data.edge.shape[0]
logits = [torch.empty(P, 17, device="cuda") for _ in range(S)]
args = []
for k in range(S):
    p = plans[k]
    args.append((ctypes.byref(p._desc), data.x.data_ptr(), data.x.stride(0), data.edge.data_ptr(), data.edge.stride(0),
                 data.edge.stride(1), data.e_attr.data_ptr(), data.bbox_idx.data_ptr(), N, E, P, logits[k].data_ptr(), 17,
                 p._ws.data_ptr(), p._ws.numel(), p._status.data_ptr(), streams[k].cuda_stream))
n = 1600
t0 = time.perf_counter()
for i in range(n):
    lib.yolat_forward_eval(*args[i % S])
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print("raw C call: host %.1f us/forward, total %.1f us/forward (%.0f graphs/s)" % ((t1 - t0) / n * 1e6, (t2 - t0) / n * 1e6, n / (t2 - t0)))
def step(i):
    k = i % S
    with torch.cuda.stream(streams[k]), torch.no_grad():
        return plans[k].run(data.x, data.edge, data.e_attr, data.bbox_idx, P)
t0 = time.perf_counter()
for i in range(n): step(i)
t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
print("plan.run:   host %.1f us/forward, total %.1f us/forward (%.0f graphs/s)" % ((t1 - t0) / n * 1e6, (t2 - t0) / n * 1e6, n / (t2 - t0)))
