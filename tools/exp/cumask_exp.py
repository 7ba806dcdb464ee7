"""Dev experiment: eval-forward throughput with every stream confined to one XCD (hipExtStreamCreateWithCUMask).
usage: cumask_exp.py <mode> <streams_per_xcd>    mode: none | interleaved (bit i -> XCD i % 8) | blocked (bit i -> XCD i // 32)"""
import ctypes, os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "tests"))
import torch
import golden_util as gu
import yolat_vectorgraphicsrecognition_amd as yv

mode = sys.argv[1] if len(sys.argv) > 1 else "none"
per = int(sys.argv[2]) if len(sys.argv) > 2 else 2
hip = ctypes.CDLL("libamdhip64.so")
data, slices, optkw, _ = yv.config("2")
model = gu.fill_state_(yv.SparseCADGCN(yv.Opt(**optkw)), 0).cuda().eval()
for k in ("x", "edge", "e_attr", "bbox_idx", "bbox", "labels"):
    data[k] = data[k].cuda()
torch.cuda.synchronize()
streams = []
for x in range(8):
    for _ in range(per):
        if mode == "none":
            streams.append(torch.cuda.Stream())
            continue
        bits = [x + 8 * j for j in range(32)] if mode == "interleaved" else [32 * x + j for j in range(32)]
        words = (ctypes.c_uint32 * 8)()
        for b in bits:
            words[b // 32] |= (1 << (b % 32))
        h = ctypes.c_void_p()
        rc = hip.hipExtStreamCreateWithCUMask(ctypes.byref(h), 8, words)
        assert rc == 0, rc
        streams.append(torch.cuda.ExternalStream(h.value))
n = len(streams)

def step(i):
    data._yolat_stage = None
    with torch.cuda.stream(streams[i % n]), torch.no_grad():
        return model(data, slices)[0]

for i in range(3 * n):
    step(i)
torch.cuda.synchronize()
K = 3000
t0 = time.perf_counter()
for i in range(K):
    step(i)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
# single-forward latency on one (masked) stream
t1 = time.perf_counter()
for i in range(200):
    step(0)
torch.cuda.synchronize()
lat = (time.perf_counter() - t1) / 200
print("mode %s, %d streams: %.0f graphs/s; one-stream latency %.3f ms" % (mode, n, K / dt, lat * 1e3))
