"""Update profiles/pmc_traffic.json from two rocprofv3 PMC summaries (tools/rocpd_pmc.py output of a --pmc FETCH_SIZE
pass and a --pmc WRITE_SIZE pass of `bench.py --config C --streams 1`): per eval-plan stage the KiB fetched / written
per launch, the summary files they came from and a digest of the kernel sources they were measured on (bench.py
reports `traffic: null` + "STALE" when the sources change afterwards).
usage: python tools/pmc_traffic.py <cfg> <fetch.txt> <write.txt> <committed-name-prefix>"""
import hashlib
import json
import os
import re
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CS = "yolat_vectorgraphicsrecognition_amd/csrc/"
# eval-plan stage (forward_eval.hip YL_STAGE names) -> (kernel name prefix, source files)
STAGES = {
    "fusion_gemm+segmax[N x 128 -> 1024 -> P] | super[P x 128 -> 1024]": ("k_fusion_rows_x6<128>", [CS + "common.hpp", CS + "x6.hpp", CS + "fusion_x6.hip"]),
    "edge_uv_mlp2_mean[E x (U+V+attr) -> 64 -> 64 -> mean]": ("k_edge_uv_mlp2_mean", [CS + "common.hpp", CS + "edge.hip"]),
    "node_uv[UV | lin_r | mlp_node, N x 64 -> 128+64+64]": ("k_gemm_nt_node3", [CS + "common.hpp", CS + "dense.hip"]),
    "graph_prep[csr+attr+segments] + node_uv[layer 0]": ("k_prep_rows_node3", [CS + "common.hpp", CS + "graph.hip"]),
}


def digest(files):
    h = hashlib.sha256()
    for f in files:
        h.update(open(os.path.join(REPO, f), "rb").read())
    return h.hexdigest()[:16]


def parse(path):
    """rocpd_pmc.py table -> {kernel name: (calls, value)} for the launch grid with the most samples (the benched
    workload; other grids of the same kernel come from warm-up / side measurements of other graph sizes)."""
    out = {}
    for line in open(path):
        m = re.match(r"^(\S.*?)\s+(\d+)\s+(\d+)\s+(\d+)\s+([0-9.]+)\s*$", line.rstrip())
        if not m or line.startswith("kernel"):
            continue
        name, calls, val = m.group(1).strip(), int(m.group(4)), float(m.group(5))
        if name not in out or calls > out[name][0]:
            out[name] = (calls, val)
    return out


def main():
    cfg, fetch, write, prefix = sys.argv[1:5]
    f, w = parse(fetch), parse(write)
    path = os.path.join(REPO, "profiles", "pmc_traffic.json")
    table = json.load(open(path)) if os.path.exists(path) else {}
    ent = table.setdefault("cfg%s" % cfg, {})
    for stage, (kern, srcs) in STAGES.items():
        fk = [k for k in f if k.startswith(kern)]
        wk = [k for k in w if k.startswith(kern)]
        if not fk or not wk:
            continue
        # several launches of one kernel per forward (edge: one per layer): average per launch
        fc = sum(f[k][0] for k in fk); fv = sum(f[k][0] * f[k][1] for k in fk) / fc
        wc = sum(w[k][0] for k in wk); wv = sum(w[k][0] * w[k][1] for k in wk) / wc
        ent[stage] = {"kernel": kern, "fetch_kib": round(fv, 1), "write_kib": round(wv, 1), "launches_sampled": fc,
                      "file": "profiles/%s_pmc_fetch.txt + profiles/%s_pmc_write.txt" % (prefix, prefix),
                      "sources": srcs, "source_digest": digest(srcs)}
    json.dump(table, open(path, "w"), indent=1, sort_keys=True)
    print(json.dumps(ent, indent=1))


if __name__ == "__main__":
    main()
