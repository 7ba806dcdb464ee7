"""The one-launch conv stack (COO instantiation) on a cfg-sized graph with a phase-ablation mask — the unit rocprofv3 --pmc is
run on to attribute LDS bank conflicts / instruction counts to the kernel's phases (tools/exp/r06_lds_pmc.sh).
usage: python tools/exp/conv_local_abl.py [cfg=5] [abl=0] [reps=5]
abl bits (yolat_conv_local_tune; results are WRONG on purpose): 1 no edge steps, 2 no node phase of layers >= 1, 4 no node
phase of layer 0, 8 no outputs"""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import yolat_vectorgraphicsrecognition_amd as yv
from yolat_vectorgraphicsrecognition_amd._lib import lib, check

cfg = sys.argv[1] if len(sys.argv) > 1 else "5"
abl = int(sys.argv[2]) if len(sys.argv) > 2 else 0
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 5
data, slices, optkw, _ = yv.config(cfg)
torch.manual_seed(0)
model = yv.SparseCADGCN(yv.Opt(**optkw)).cuda().eval().set_eval_precision("bf16")
dev = torch.device("cuda")
x, edge, ea, bb = data.x.cuda().contiguous(), data.edge.cuda(), data.e_attr.cuda().contiguous(), data.bbox_idx.cuda()
with torch.no_grad():
    model(data, slices)
plan = model._yolat_plan
h, base = plan._desc_h, plan._desc
N, P, E = int(x.shape[0]), int(data.bbox.shape[0]), int(edge.shape[0])
D, F = base.C * base.n_blocks_out, base.F
feats = torch.zeros((N, D), dtype=torch.bfloat16, device=dev)
Z = torch.zeros((P, 2 * (F + D)), dtype=torch.float32, device=dev)
flag = torch.zeros(1, dtype=torch.int32, device=dev)
status = torch.zeros(1, dtype=torch.int32, device=dev)
ws = torch.empty(int(lib.yolat_batch_locality_workspace_bytes(N, E, P)) + 16, dtype=torch.uint8, device=dev)
st = torch.cuda.current_stream().cuda_stream
lib.yolat_conv_local_tune(int(os.environ.get("CL_NW", "0")), 0, abl, None)


def run():
    check(lib.yolat_conv_stack_local_bf16_coo(ctypes.byref(h), h.conv_local, x.data_ptr(), x.stride(0), edge.data_ptr(),
                                              edge.stride(0), edge.stride(1), ea.data_ptr(), bb.data_ptr(), N, E, P,
                                              feats.data_ptr(), D, Z.data_ptr(), Z.stride(0), flag.data_ptr(), status.data_ptr(),
                                              ws.data_ptr(), ws.numel(), st))


for _ in range(3):
    run()
torch.cuda.synchronize()
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
s.record()
for _ in range(reps):
    run()
e.record()
torch.cuda.synchronize()
print("conv_local COO cfg %s abl %d: %.1f us per launch (local prep included), flag %d status %d"
      % (cfg, abl, s.elapsed_time(e) / reps * 1e3, int(flag.item()), int(status.item())))

# ---- in-kernel phase stamps (s_memtime, shader cycles) of the first two tiles of every workgroup: CL_STAMPS=1
if os.environ.get("CL_STAMPS", "0") == "1":
    import numpy as np
    nwg = 4096
    stamps = torch.zeros(nwg * 64, dtype=torch.int64, device=dev)
    lib.yolat_conv_local_tune(int(os.environ.get("CL_NW", "0")), 0, abl, stamps.data_ptr())
    if os.environ.get("CL_CSR", "0") == "1":
        from yolat_vectorgraphicsrecognition_amd import ops
        from yolat_vectorgraphicsrecognition_amd._lib import GraphCsr
        g = ops.build_graph(edge, ea, bb, N, P)
        gc = GraphCsr(*g.device_pointers())
        check(lib.yolat_conv_stack_local_bf16(ctypes.byref(h), h.conv_local, x.data_ptr(), x.stride(0), ctypes.byref(gc), N, E, P,
                                              feats.data_ptr(), D, Z.data_ptr(), Z.stride(0), flag.data_ptr(), st))
    else:
        run()
    torch.cuda.synchronize()
    lib.yolat_conv_local_tune(0, 0, 0, None)
    st_ = stamps.cpu().numpy().reshape(nwg, 64)
    st_ = st_[st_[:, 0] != 0]
    L = base.n_blocks
    per_tile = 3 + 5 * L
    names = ["tile load (+ sort)", "stream set-up"]
    for l in range(L):
        names += ["L%d node phase" % l, "L%d node barrier" % l, "L%d mean-pool + steps" % l, "L%d finalize+barrier" % l, "L%d outputs" % l]
    for tile in range(2):
        seg = st_[:, tile * per_tile:(tile + 1) * per_tile + 1].astype(np.float64)
        seg = seg[(seg != 0).all(1)]
        d = np.diff(seg, axis=1)
        print("%s tile %d: %d workgroups, total %.0f cycles (to next tile start)"
              % ("CSR" if os.environ.get("CL_CSR", "0") == "1" else "COO", tile, len(seg), (seg[:, -1] - seg[:, 0]).mean()))
        for i, nme in enumerate(names + ["zero Z + barrier"]):
            if i < d.shape[1]:
                print("   %-24s mean %7.0f  p10 %7.0f  p90 %7.0f" % (nme, d[:, i].mean(), np.percentile(d[:, i], 10), np.percentile(d[:, i], 90)))

# ---- per-wave arrival at the end of layer 1's steps / finalize / barrier (first tile): CL_WAVES=1
if os.environ.get("CL_WAVES", "0") == "1":
    import numpy as np
    nwg = 4096
    stamps = torch.zeros(nwg * 64, dtype=torch.int64, device=dev)
    lib.yolat_conv_local_tune(int(os.environ.get("CL_NW", "0")), 0, 16, stamps.data_ptr())
    run()
    torch.cuda.synchronize()
    lib.yolat_conv_local_tune(0, 0, 0, None)
    st_ = stamps.cpu().numpy().reshape(nwg, 64)
    st_ = st_[st_[:, 0] != 0][:, 32:56].reshape(-1, 8, 3).astype(np.float64)
    st_ = st_[(st_ != 0).all((1, 2))]
    t0 = st_[:, :, 0].min(1, keepdims=True)
    print("layer 1, first tile, %d workgroups: per wave (cycles after the first wave finished its steps)" % len(st_))
    for w in range(8):
        print("   wave %d: steps done %6.0f   finalize done %6.0f   barrier passed %6.0f" % (
            w, (st_[:, w, 0] - t0[:, 0]).mean(), (st_[:, w, 1] - t0[:, 0]).mean(), (st_[:, w, 2] - t0[:, 0]).mean()))
