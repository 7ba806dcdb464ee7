#!/bin/bash
# round-3 first GPU pass: new tests, gradient-gap tables, the extended bench line
mkdir -p gpurun_out/r03a
cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r03a/pytest.log 2>&1; echo "pytest rc $?" >> gpurun_out/r03a/pytest.log
tail -5 gpurun_out/r03a/pytest.log
for c in "3 33" "4 44" "5 55"; do
  timeout 900 python tools/exp/grad_gap.py $c > gpurun_out/r03a/gap_cfg${c%% *}.txt 2>&1
done
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r03a/bench.json 2> gpurun_out/r03a/bench.err; echo "bench rc $?"
tail -c 600 gpurun_out/r03a/bench.err
