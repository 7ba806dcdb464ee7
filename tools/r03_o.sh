#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
n=r03_train_cfg3_kernel_stats
rm -rf $R/gpurun_out/$n
timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/$n --output-format rocpd -- python $R/bench.py --mode train --config 3 --steps 15 --warmup 3 --no-cpu-baseline --no-roofline --no-extras > $R/gpurun_out/$n.log 2>&1
f=$(find $R/gpurun_out/$n -name "*.db" | head -1)
python $R/tools/rocpd_stats.py $f > $R/gpurun_out/$n.txt
rm -rf $R/gpurun_out/$n
head -3 $R/gpurun_out/$n.txt | cut -c1-150
grep "k_gemm_nt<\|k_gemm_tn" $R/gpurun_out/$n.txt | cut -c1-150
