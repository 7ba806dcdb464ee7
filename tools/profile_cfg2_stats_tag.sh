# kernel stats of the cfg-2 eval forward -> gpurun_out/${TAG}_fwd_cfg2_kernel_stats.txt
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
n=${TAG:-r04}_fwd_cfg2_kernel_stats
rm -rf $R/gpurun_out/$n
timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/$n --output-format rocpd -- python $R/bench.py --config 2 --steps 50 --warmup 5 --streams 1 --no-cpu-baseline --no-roofline --no-extras > $R/gpurun_out/$n.log 2>&1
f=$(find $R/gpurun_out/$n -name "*.db" | head -1)
python $R/tools/rocpd_stats.py $f > $R/gpurun_out/$n.txt
rm -rf $R/gpurun_out/$n
head -24 $R/gpurun_out/$n.txt
