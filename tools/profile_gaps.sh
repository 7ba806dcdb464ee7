cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace -d $R/gpurun_out/kt4 --output-format rocpd -- python $R/bench.py --steps 200 --warmup 10 --no-cpu-baseline --no-roofline > $R/gpurun_out/kt4.log 2>&1
cd $R
f=$(find gpurun_out/kt4 -name "*.db" | head -1)
python tools/rocpd_gaps.py $f 26
python tools/rocpd_stats.py $f > gpurun_out/kt4.txt
rm -rf gpurun_out/kt4
