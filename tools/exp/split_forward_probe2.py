"""GPU-side probe: one cfg-2 forward against the same work as TWO half-size forwards (proposal ranges are independent in eval
mode) on two streams, fork / join per round through stream waits, many rounds queued (the host stays ahead: plan.run directly).
usage: python tools/exp/split_forward_probe2.py"""
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
_set = torch._C._cuda_setStream


def batch(P, seed):
    d, s = yv.synth_batch(1, seed, num_proposals=P, nodes_lo=25, nodes_hi=25, edges_per_proposal=100)
    return [d[k].cuda() for k in ("x", "edge", "e_attr", "bbox_idx")] + [P]


def use(st):
    _set(stream_id=st.stream_id, device_index=st.device_index, device_type=st.device_type)


def plans_for(streams, parts):
    out = []
    with torch.no_grad():
        for st, part in zip(streams, parts):
            use(st)
            d = yv.Data(x=part[0])
            d.edge, d.e_attr, d.bbox_idx = part[1], part[2], part[3]
            d.bbox = torch.zeros(part[4], 4, device="cuda")
            model(d, None)
            out.append(model._yolat_plan)
    return out


def timed(parts, streams, rounds=400):
    cur = torch.cuda.current_stream()
    plans = plans_for(streams, parts)
    use(cur)

    def one_round():
        for part, st, pl in zip(parts, streams, plans):
            if st is not cur:
                st.wait_stream(cur)
                use(st)
            pl.run(*part)
        use(cur)
        for st in streams:
            if st is not cur:
                cur.wait_stream(st)
    with torch.no_grad():
        for _ in range(30):
            one_round()
        torch.cuda.synchronize()
        res = []
        for _ in range(5):
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            t0 = time.perf_counter()
            s.record()
            for _ in range(rounds):
                one_round()
            e.record()
            host = (time.perf_counter() - t0) / rounds
            torch.cuda.synchronize()
            res.append((s.elapsed_time(e) / rounds * 1e3, host * 1e6))
    res.sort()
    return res[2]


cur = torch.cuda.current_stream()
print("full forward (P = 400) on one stream:            GPU %.1f us per round (host enqueue %.1f us)" % timed([batch(400, 2)], [cur]))
for k in (2, 3, 4):
    parts = [batch(400 // k + (1 if i < 400 % k else 0), 10 + i) for i in range(k)]
    streams = [cur] + [torch.cuda.Stream() for _ in range(k - 1)]
    print("%d parts on %d streams (fork / join per round):     GPU %.1f us per round (host enqueue %.1f us)" % ((k, k) + timed(parts, streams)))
    print("%d parts on ONE stream:                             GPU %.1f us per round (host enqueue %.1f us)" % ((k,) + timed(parts, [cur] * k)))
