#!/bin/bash
mkdir -p gpurun_out/r03b
cd $GRAFT_REPO_ROOT
python tools/exp/hchain_soak.py 10 2>&1 | grep "shape\|SOAK"
for w in 512 768 1024; do YOLAT_HCHAIN_WGS=$w timeout 300 python tools/exp/hedge_bench.py 5 20 2 2>&1 | grep cfg; done | tee gpurun_out/r03b/hedge.txt
timeout 300 python tools/exp/hedge_bench.py 3 20 0 2>&1 | grep cfg | tee -a gpurun_out/r03b/hedge.txt
CMD="python $GRAFT_REPO_ROOT/tools/exp/hedge_bench.py 5 10 2" KRE="k_edge_chain" TAG=hchain bash tools/exp/pmc_kernel.sh > gpurun_out/r03b/pmc_hchain.txt 2>&1
cat gpurun_out/r03b/pmc_hchain.txt
