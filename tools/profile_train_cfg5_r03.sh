cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
n=r03_train_cfg5_kernel_stats
rm -rf $R/gpurun_out/$n
timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/$n --output-format rocpd -- python $R/bench.py --mode train --config 5 --steps 10 --warmup 3 --no-cpu-baseline --no-roofline --no-extras > $R/gpurun_out/$n.log 2>&1
f=$(find $R/gpurun_out/$n -name "*.db" | head -1)
python $R/tools/rocpd_stats.py $f > $R/gpurun_out/$n.txt
rm -rf $R/gpurun_out/$n
