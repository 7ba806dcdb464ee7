cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for args in "400 2304 512" "400 512 256" "2000 2304 512"; do
n=linx6
rm -rf $R/gpurun_out/$n
timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/$n --output-format rocpd -- python $R/tools/exp/linear_x6_bench.py $args > $R/gpurun_out/$n.log 2>&1
f=$(find $R/gpurun_out/$n -name "*.db" | head -1)
echo "== $args"; python $R/tools/rocpd_stats.py $f | grep -E "k_linear_x6_sk|k_split_bf16x3_packed|k_gemm_nt" | cut -c1-60,100-160
rm -rf $R/gpurun_out/$n
done
