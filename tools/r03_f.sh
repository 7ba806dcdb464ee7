#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_ops.py -x -q -k "linear_b or wide_weight or gemm_x6" 2>&1 | tail -3
timeout 900 python -m pytest tests/test_gpu_model.py tests/test_gpu_configs.py -x -q -k "train or grad or fused" 2>&1 | tail -3
for np in 3 6; do
YOLAT_GRAD_GEMM_PRODUCTS=$np timeout 300 python bench.py --mode train --config 3 --steps 20 --warmup 5 --no-cpu-baseline --no-roofline --no-extras 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('train3 np=$np', d['ms_per_step'])"
done
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
n=r03_train_cfg3_kernel_stats
rm -rf $R/gpurun_out/$n
timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/$n --output-format rocpd -- python $R/bench.py --mode train --config 3 --steps 10 --warmup 3 --no-cpu-baseline --no-roofline --no-extras > $R/gpurun_out/$n.log 2>&1
f=$(find $R/gpurun_out/$n -name "*.db" | head -1)
python $R/tools/rocpd_stats.py $f > $R/gpurun_out/$n.txt
rm -rf $R/gpurun_out/$n
head -14 $R/gpurun_out/$n.txt | cut -c1-150
