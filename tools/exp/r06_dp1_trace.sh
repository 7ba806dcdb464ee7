cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf $R/gpurun_out/dp1trace
timeout 600 rocprofv3 --kernel-trace -d $R/gpurun_out/dp1trace --output-format rocpd -- python $R/tools/exp/dp1_overlap_trace.py 24 > $R/gpurun_out/dp1trace.log 2>&1
f=$(find $R/gpurun_out/dp1trace -name "*.db" | head -1)
if [ -n "$f" ]; then python $R/tools/rocpd_overlap.py $f > $R/gpurun_out/r06_dp1_overlap.txt; else tail -5 $R/gpurun_out/dp1trace.log; fi
rm -rf $R/gpurun_out/dp1trace
tail -30 $R/gpurun_out/r06_dp1_overlap.txt; tail -3 $R/gpurun_out/dp1trace.log
