#!/bin/bash
# end-of-round validation on the GPU box: full GPU suite, the default bench line, refreshed profiles + traffic table
# (copy gpurun_out/r04_*.txt, pmc_traffic.json and r04_bench_line.json into profiles/ afterwards)
cd $GRAFT_REPO_ROOT
timeout 3200 python -m pytest tests -m gpu -x -q 2>&1 | tail -5
TAG=r04 timeout 2400 bash tools/profile_r04.sh 2>&1 | tail -5
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r04_bench_line.json 2> gpurun_out/r04_bench_line.log
tail -c 900 gpurun_out/r04_bench_line.json
