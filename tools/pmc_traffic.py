"""Update profiles/pmc_traffic.json from two rocprofv3 PMC summaries (tools/rocpd_pmc.py output of a --pmc FETCH_SIZE
pass and a --pmc WRITE_SIZE pass of `bench.py --config C --streams 1`): per eval-plan stage the KiB fetched / written
per launch, the summary files they came from and a digest of the kernel sources they were measured on (bench.py
reports `traffic: null` + "STALE" when the sources change afterwards).
usage: python tools/pmc_traffic.py <cfg> <fetch.txt> <write.txt> <committed-name-prefix> [bf16]"""
import hashlib
import json
import os
import re
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CS = "yolat_vectorgraphicsrecognition_amd/csrc/"
# eval-plan stage (forward_eval.hip YL_STAGE names) -> (kernel name prefix, source files)
# pick: which launch grid of the kernel belongs to the stage when one forward launches the kernel with several grids
#   "all"  average over every grid;  "last_layer" / "with_next": the node-tile edge kernel of a small graph is launched once
#   per conv layer — the last layer's launch carries the pooling rider (the largest grid), the others the next layer's node
#   side (EdgeNext: smaller grids)
STAGES = {
    "fusion_gemm+segmax[N x 128 -> 1024 -> P] | super[P x 128 -> 1024]": ("k_fusion_rows_x6<128", [CS + "common.hpp", CS + "x6.hpp", CS + "fusion_x6.hip"], "all"),
    "edge_uv_mlp2_mean[E x (U+V+attr) -> 64 -> 64 -> mean]": ("k_edge_uv_mlp2_mean", [CS + "common.hpp", CS + "edge.hip"], "last_layer"),
    "edge_uv_mlp2_mean+node_uv_next[E x (U+V+attr) -> 64 -> 64 -> mean; N x 64 -> 128+64+64]": ("k_edge_uv_mlp2_mean", [CS + "common.hpp", CS + "edge.hip"], "with_next"),
    "node_uv[UV | lin_r | mlp_node, N x 64 -> 128+64+64]": ("k_gemm_nt_node3", [CS + "common.hpp", CS + "dense.hip"], "all"),
    # small graphs: the one-launch form (k_prep_small); larger ones: the last of the four launches
    "graph_prep[csr+attr+segments] + node_uv[layer 0]": (("k_prep_small", "k_prep_rows_node3"), [CS + "common.hpp", CS + "graph.hip"], "all"),
}


# bf16-storage forward (bf16_eval.hip YL_HSTAGE names; round 3)
STAGES_BF16 = {
    # round 5: the one-launch proposal-local conv stack (the per-layer edge / node launches behind it are dead, gated launches
    # on such a batch: their rows in the PMC files are not the stages' traffic and are dropped below)
    "conv_local_bf16[all conv layers + pooling prologue, one launch]": ("k_conv_local_h", [CS + "common.hpp", CS + "conv_local.hip"], "all"),
    "edge_uv_mlp2_mean_bf16[E x (U+V+attr) -> 64 -> 64 -> mean]": ("k_edge_chain_h", [CS + "common.hpp", CS + "edge_chain.hip"], "all"),
    "fusion_gemm_bf16+segmax[N x 128 -> 1024 -> P] | super[P x 128 -> 1024]": ("k_hfusion_rows8<128", [CS + "common.hpp", CS + "segmax.hpp", CS + "fusion_h8.hip"], "all"),
    "node_uv_bf16[UV | lin_r | mlp_node, N x 64 -> 128+64+64]": ("k_hgemm_node3", [CS + "common.hpp", CS + "bf16_eval.hip"], "all"),
    "graph_prep[csr+attr+segments] + node_uv[layer 0, bf16 out]": ("k_prep_rows_node3", [CS + "common.hpp", CS + "graph.hip"], "all"),
}


def digest(files):
    h = hashlib.sha256()
    for f in files:
        h.update(open(os.path.join(REPO, f), "rb").read())
    return h.hexdigest()[:16]


def parse(path):
    """rocpd_pmc.py table -> {kernel name: [(grid_x, calls, value), ...]}, restricted per kernel to the grids with the
    most samples (the benched workload; other grids of the same kernel come from warm-up / side measurements of other
    graph sizes and have fewer calls)."""
    rows = {}
    for line in open(path):
        m = re.match(r"^(\S.*?)\s+(\d+)\s+(\d+)\s+(\d+)\s+([0-9.]+)\s*$", line.rstrip())
        if not m or line.startswith("kernel"):
            continue
        name, grid, calls, val = m.group(1).strip(), int(m.group(2)), int(m.group(4)), float(m.group(5))
        rows.setdefault(name, []).append((grid, calls, val))
    out = {}
    for name, lst in rows.items():
        top = max(c for _, c, _ in lst)
        out[name] = [r for r in lst if r[1] == top]
    return out


def select(table, kern, pick):
    if isinstance(kern, tuple):          # candidates in order of preference: the first that was launched
        for k in kern:
            r = select(table, k, pick)
            if r is not None:
                return r
        return None
    rows = [r for k in table if k.startswith(kern) for r in table[k]]
    if not rows:
        return None
    if pick != "all" and len(rows) > 1:
        gmax = max(r[0] for r in rows)
        rows = [r for r in rows if (r[0] == gmax) == (pick == "last_layer")]
    elif pick == "with_next" and len(rows) == 1:
        return None                      # a single grid: no launch carried the next layer's node side
    calls = sum(r[1] for r in rows)
    return calls, sum(r[1] * r[2] for r in rows) / calls


def main():
    cfg, fetch, write, prefix = sys.argv[1:5]
    stages = STAGES_BF16 if (len(sys.argv) > 5 and sys.argv[5] == "bf16") else STAGES
    f, w = parse(fetch), parse(write)
    path = os.path.join(REPO, "profiles", "pmc_traffic.json")
    table = json.load(open(path)) if os.path.exists(path) else {}
    ent = table.setdefault("cfg%s" % cfg, {})
    for stage in list(ent):
        if stage not in STAGES and stage not in STAGES_BF16:
            del ent[stage]
    for stage, (kern, srcs, pick) in stages.items():
        fs, ws = select(f, kern, pick), select(w, kern, pick)
        if fs is None or ws is None:
            ent.pop(stage, None)
            continue
        ent[stage] = {"kernel": kern if isinstance(kern, str) else "|".join(kern), "fetch_kib": round(fs[1], 1), "write_kib": round(ws[1], 1), "launches_sampled": fs[0],
                      "launch_pick": pick,
                      "file": "profiles/%s_pmc_fetch.txt + profiles/%s_pmc_write.txt" % (prefix, prefix),
                      "sources": srcs, "source_digest": digest(srcs)}
    if "conv_local_bf16[all conv layers + pooling prologue, one launch]" in ent:
        for stage in list(ent):
            if stage.startswith("edge_uv_mlp2_mean_bf16") or stage.startswith("node_uv_bf16"):
                del ent[stage]
    json.dump(table, open(path, "w"), indent=1, sort_keys=True)
    print(json.dumps(ent, indent=1))


if __name__ == "__main__":
    main()
