"""Would ONE cfg-2 forward finish sooner as k independent part-forwards on k streams?  (In eval mode a proposal's logits depend
only on its own nodes and edges, so a batch splits by proposal ranges.)  Emulated with k synthetic batches of P / k proposals
each, one per stream, enqueued round-robin; time per round against the full forward one at a time.
usage: python tools/exp/split_forward_probe.py"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
import golden_util as gu
import yolat_vectorgraphicsrecognition_amd as yv

optkw = dict(n_classes=17, n_blocks=2, n_blocks_out=2)
model = gu.fill_state_(yv.SparseCADGCN(yv.Opt(**optkw)), 0).cuda().eval()


def batch(P, seed):
    d, s = yv.synth_batch(1, seed, num_proposals=P, nodes_lo=25, nodes_hi=25, edges_per_proposal=100)
    for k in ("x", "edge", "e_attr", "bbox_idx", "bbox"):
        d[k] = d[k].cuda()
    return d, s


def timed(parts, streams, rounds=300):
    cur = torch.cuda.current_stream()

    def one_round():
        for (d, s), st in zip(parts, streams):
            st.wait_stream(cur)
            with torch.cuda.stream(st):
                d._yolat_stage = None
                model(d, s)
        for st in streams:
            cur.wait_stream(st)
    with torch.no_grad():
        for _ in range(20):
            one_round()
        torch.cuda.synchronize()
        best = []
        for _ in range(5):
            t0 = time.perf_counter()
            for _ in range(rounds):
                one_round()
                torch.cuda.synchronize()          # one "forward" at a time: latency, not throughput
            best.append((time.perf_counter() - t0) / rounds)
    best.sort()
    return best[2] * 1e6


full = [batch(400, 2)]
print("full forward (P = 400), one stream, synchronised per forward: %.1f us" % timed(full, [torch.cuda.current_stream()]))
for k in (2, 4):
    parts = [batch(400 // k, 10 + i) for i in range(k)]
    streams = [torch.cuda.Stream() for _ in range(k)]
    print("%d parts of P = %d on %d streams, synchronised per round:        %.1f us" % (k, 400 // k, k, timed(parts, streams)))
    print("%d parts of P = %d on ONE stream (serial), synchronised per round: %.1f us" % (k, 400 // k, timed(parts, [torch.cuda.current_stream()] * k)))
