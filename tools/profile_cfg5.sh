cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
CMD="python $R/bench.py --config 5 --steps 30 --warmup 5 --no-cpu-baseline --no-roofline --streams 1"
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/kt5 --output-format rocpd -- $CMD > $R/gpurun_out/kt5.log 2>&1
cd $R
f=$(find gpurun_out/kt5 -name "*.db" | head -1); python tools/rocpd_stats.py $f > gpurun_out/kt5.txt
find gpurun_out -name "*.db" -delete
