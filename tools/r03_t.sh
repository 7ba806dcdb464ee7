#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 2400 python -m pytest tests/test_gpu_model.py tests/test_gpu_configs.py tests/test_gpu_ops.py tests/test_gpu_bf16.py -x -q 2>&1 | tail -2
timeout 300 python bench.py --mode train --config 3 --steps 30 --warmup 5 --no-cpu-baseline --no-roofline --no-extras 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('train3', d['ms_per_step'])"
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
n=r03_train_cfg3_kernel_stats
rm -rf $R/gpurun_out/$n
timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/$n --output-format rocpd -- python $R/bench.py --mode train --config 3 --steps 15 --warmup 3 --no-cpu-baseline --no-roofline --no-extras > $R/gpurun_out/$n.log 2>&1
f=$(find $R/gpurun_out/$n -name "*.db" | head -1)
python $R/tools/rocpd_stats.py $f > $R/gpurun_out/$n.txt
rm -rf $R/gpurun_out/$n
grep "k_bn_merge\|k_bn_finalize\|dispatches" $R/gpurun_out/$n.txt | cut -c1-150
