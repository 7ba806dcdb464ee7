"""Does the training step's two-stream overlap depend on how many streams the process has used before?  (bench.py's train legs
ran 12 % slower after the hand-over legs had drawn three more streams from torch's pool.)
usage: python tools/exp/side_stream_queue.py <n_streams_used_before> [priority of the side stream, default 0]"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
import yolat_vectorgraphicsrecognition_amd as yv
import golden_util as gu
from yolat_vectorgraphicsrecognition_amd import engine
n_before = int(sys.argv[1])
prio = int(sys.argv[2]) if len(sys.argv) > 2 else 0
keep = []
x = torch.zeros(1024, device="cuda")
for i in range(n_before):
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        x.add_(1)
    keep.append(s)
torch.cuda.synchronize()
if prio:
    engine._SIDE[torch.cuda.current_device()] = {"stream": torch.cuda.Stream(priority=prio), "dirty": False}
data, slices, optkw, _ = yv.config("3")
opt = yv.Opt(**optkw)
for k, v in list(data.__dict__.items()):
    if torch.is_tensor(v):
        data.__dict__[k] = v.cuda()
model = gu.fill_state_(yv.SparseCADGCN(opt), 0).cuda()
tr = yv.Trainer(model, opt, lr=2.5e-4, weight_decay=1e-5)
for _ in range(5):
    data._yolat_stage = None
    tr.step(data, slices)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(50):
    data._yolat_stage = None
    tr.step(data, slices)
torch.cuda.synchronize()
print("streams used before: %2d, side-stream priority %d: %.3f ms per cfg-3 train step; probe %s" % (n_before, prio, (time.perf_counter() - t0) / 50 * 1e3, engine.SIDE_PROBE))
