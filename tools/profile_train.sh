cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
CMD="python $R/bench.py --mode train --steps 20 --warmup 5 --no-cpu-baseline --no-roofline"
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/kt_train --output-format rocpd -- $CMD > $R/gpurun_out/kt_train.log 2>&1
cd $R
f=$(find gpurun_out/kt_train -name "*.db" | head -1); python tools/rocpd_stats.py $f > gpurun_out/kt_train.txt
find gpurun_out -name "*.db" -delete
grep graphs_per_sec gpurun_out/kt_train.log | cut -c1-300
