"""Dev tool: error of the bf16-storage eval forward against the CPU oracle (fp32) for a few model shapes."""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "tests"))
import numpy as np, torch
import golden_util as gu
from oracle import oracle_torch as orc
import yolat_vectorgraphicsrecognition_amd as yv

for cin, blocks, blocks_out, classes, seed in [(5, 2, 2, 17, 1), (6, 3, 2, 22, 2), (5, 4, 2, 17, 3), (3, 2, 1, 5, 4),
                                               (5, 4, 4, 17, 5), (5, 4, 2, 17, 6), (5, 4, 2, 17, 7)]:
    optkw = dict(n_classes=classes, n_blocks=blocks, n_blocks_out=blocks_out, in_channels=cin)
    d = yv.synth_graph(num_proposals=2000, nodes_lo=2, nodes_hi=40, edge_factor=2.1, n_classes=classes, seed=70 + seed)
    if cin != 5:
        d.x = torch.randn(d.x.shape[0], cin, generator=torch.Generator().manual_seed(cin)) * 0.7
    model = gu.fill_state_(yv.SparseCADGCN(yv.Opt(**optkw)), 40 + seed).cuda().eval().set_eval_precision("bf16")
    ref = gu.fill_state_(orc.SparseCADGCN(orc.Opt(**optkw)), 40 + seed).eval()
    with torch.no_grad():
        got = model(d, None)[0].cpu().double()
        want = ref(d, None)[0].double()
    err = (got - want).abs()
    print("cin %d blocks %d/%d: max err %.2e of scale, rms %.2e, argmax agreement %.4f" % (
        cin, blocks, blocks_out, float(err.max() / want.abs().max()),
        float(err.pow(2).mean().sqrt() / want.pow(2).mean().sqrt()),
        float((got.argmax(1) == want.argmax(1)).double().mean())))
