"""Soak: the chained bf16 edge kernel must be bit-identical run to run, on several shapes, many times."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
import torch
import yolat_vectorgraphicsrecognition_amd as yv
import test_gpu_bf16 as T
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 30
shapes = [dict(n_props=3000, nodes_lo=25, nodes_hi=25, edges_per_proposal=150),
          dict(n_props=8000, nodes_lo=25, nodes_hi=25, edges_per_proposal=150),
          dict(n_props=900, nodes_lo=3, nodes_hi=40, edge_factor=3.0),
          dict(n_props=2500, nodes_lo=4, nodes_hi=30, edge_factor=1.2)]
total_bad = 0
for sh in shapes:
    kw = dict(sh)
    args = T._edge_stage_case(yv, kw.pop("n_props"), kw.pop("nodes_lo"), kw.pop("nodes_hi"), 7, **kw)
    want, mscale, flip = T._edge_stage_reference(*args)
    first = T._run_edge_stage(yv, *args, variant=2)
    d = (first.double() - want).abs()
    tol = want.abs() * 2.0 ** -8 + 2e-5 * mscale
    nbad = int((d > tol + flip).sum())
    diffs = 0
    for _ in range(reps):
        o = T._run_edge_stage(yv, *args, variant=2)
        diffs += int((o.view(torch.int16) != first.view(torch.int16)).sum())
    print("shape %s: E=%d  beyond 2 roundings: %d  run-to-run differing elements over %d runs: %d" % (sh, args[0].E, nbad, reps, diffs))
    total_bad += nbad + diffs
print("SOAK", "OK" if total_bad == 0 else "FAILED")
