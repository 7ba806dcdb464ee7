"""Ablation of k_fusion_rows_x6 (YOLAT_FX_ABL bits: 1 no epilogue, 2 no MFMAs, 4 no W staging): per-stage HIP-event
times of the eval plan at cfg 2 and cfg 5.  Results are only timings — the outputs are wrong by construction."""
import os, sys
import torch
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tests"))
import bench
import yolat_vectorgraphicsrecognition_amd as yv
import golden_util as gu

for cfg in sys.argv[1:] or ["2", "5"]:
    data, slices, optkw, n_graphs = yv.config(cfg)
    opt = yv.Opt(**optkw)
    model = gu.fill_state_(yv.SparseCADGCN(opt), 0).cuda().eval()
    bench.to_device(data)

    def step():
        data._yolat_stage = None
        with torch.no_grad():
            return model(data, slices)[0]
    for _ in range(5):
        step()
    torch.cuda.synchronize()
    prof = bench.plan_profile(step, 30)
    for k, v in prof.items():
        if "fusion" in k:
            print("cfg %s ABL=%s %-40s %.2f us" % (cfg, os.environ.get("YOLAT_FX_ABL", "0"), k, 1e3 * v["ms_avg"]))
