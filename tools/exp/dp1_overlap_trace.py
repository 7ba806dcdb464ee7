"""The data-parallel step with its exchange forced on in a one-rank `nccl` (= librccl) group, for a rocprofv3 kernel trace:
where does the head bucket's all-reduce sit relative to the conv layers' backward?  (tools/exp/r06_dp1_trace.sh)
usage: python tools/exp/dp1_overlap_trace.py [steps=12]"""
import os
import sys

import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
import golden_util as gu
import yolat_vectorgraphicsrecognition_amd as yv

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 12
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29611")
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
data, slices, optkw, _ = yv.config("4")
for k in ("x", "edge", "e_attr", "bbox_idx", "bbox", "labels"):
    data[k] = data[k].cuda()
opt = yv.Opt(**optkw)
model = gu.fill_state_(yv.SparseCADGCN(opt), 0).cuda()
tr = yv.Trainer(model, opt, lr=2.5e-4, weight_decay=1e-5, force_exchange=True)
for _ in range(steps):
    data._yolat_stage = None
    tr.step(data, slices)
torch.cuda.synchronize()
print("steps through yolat_train_step:", tr.plan_steps)
dist.destroy_process_group()
