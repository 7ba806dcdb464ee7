set -x
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
CMD="python $R/bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-roofline --streams 1"
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/kt3 --output-format rocpd -- $CMD > $R/gpurun_out/kt3.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $R/gpurun_out/pmc_fetch --output-format rocpd -- $CMD > $R/gpurun_out/pmc_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $R/gpurun_out/pmc_write --output-format rocpd -- $CMD > $R/gpurun_out/pmc_write.log 2>&1
cd $R
for d in kt3; do f=$(find gpurun_out/$d -name "*.db" | head -1); python tools/rocpd_stats.py $f > gpurun_out/$d.txt; done
for d in pmc_fetch pmc_write; do f=$(find gpurun_out/$d -name "*.db" | head -1); python tools/rocpd_pmc.py $f > gpurun_out/$d.txt; done
find gpurun_out -name "*.db" -size +30M -delete
tail -3 gpurun_out/kt3.log
