#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for prec in fp32 bf16; do
n=x_cfg5_$prec
rm -rf $R/gpurun_out/$n
timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/$n --output-format rocpd -- python $R/bench.py --config 5 --precision $prec --steps 10 --warmup 3 --streams 1 --no-cpu-baseline --no-roofline --no-extras > $R/gpurun_out/$n.log 2>&1
f=$(find $R/gpurun_out/$n -name "*.db" | head -1)
python $R/tools/rocpd_stats.py $f > $R/gpurun_out/$n.txt
rm -rf $R/gpurun_out/$n
grep "k_prep\|k_node3\|k_gemm_nt_node3\|k_zero" $R/gpurun_out/$n.txt | cut -c1-150
done
