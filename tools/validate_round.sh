#!/bin/bash
# end-of-round validation on the GPU box: full GPU suite, the default bench line, refreshed profiles + traffic table
# (copy gpurun_out/r03_*.txt, pmc_traffic.json and r03_bench_line.json into profiles/ afterwards)
cd $GRAFT_REPO_ROOT
timeout 3000 python -m pytest tests -m gpu -x -q 2>&1 | tail -5
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r03_bench_line.json 2> gpurun_out/r03_bench_line.log
tail -c 600 gpurun_out/r03_bench_line.json
TAG=r03 timeout 2400 bash tools/profile_r03.sh 2>&1 | tail -5
