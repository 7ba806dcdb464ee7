cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for args in "200000 64 192" "200000 64 64" "200000 64 256" "43520 64 192"; do
n=linx6
rm -rf $R/gpurun_out/$n
timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/$n --output-format rocpd -- python $R/tools/exp/linear_x6_bench.py $args > $R/gpurun_out/$n.log 2>&1
f=$(find $R/gpurun_out/$n -name "*.db" | head -1)
echo "== $args"; python $R/tools/rocpd_stats.py $f | head -8 | cut -c1-70,100-150
rm -rf $R/gpurun_out/$n
done
