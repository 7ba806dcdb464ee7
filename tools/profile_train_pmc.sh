# FETCH_SIZE / WRITE_SIZE passes of the cfg-5 train step (one stream), fp32 and bf16 storage: HBM bytes per launch of the
# round-4 training kernels.  Outputs: gpurun_out/${TAG}_train_cfg5_{fp32,bf16}_pmc_{fetch,write}.txt
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
TAG=${TAG:-r04}
run_pmc() {  # name "counters" cmd...
  n=$1; ctr=$2; shift; shift
  rm -rf $R/gpurun_out/$n
  timeout 600 rocprofv3 --kernel-trace --pmc $ctr -d $R/gpurun_out/$n --output-format rocpd -- "$@" > $R/gpurun_out/$n.log 2>&1
  f=$(find $R/gpurun_out/$n -name "*.db" | head -1)
  if [ -n "$f" ]; then python $R/tools/rocpd_pmc.py $f > $R/gpurun_out/$n.txt; else tail -5 $R/gpurun_out/$n.log; fi
  rm -rf $R/gpurun_out/$n
}
T5="python $R/bench.py --mode train --config 5 --steps 6 --warmup 2 --no-cpu-baseline --no-roofline --no-extras --no-side-stream"
run_pmc ${TAG}_train_cfg5_fp32_pmc_fetch "FETCH_SIZE" $T5
run_pmc ${TAG}_train_cfg5_fp32_pmc_write "WRITE_SIZE" $T5
run_pmc ${TAG}_train_cfg5_bf16_pmc_fetch "FETCH_SIZE" $T5 --precision bf16
run_pmc ${TAG}_train_cfg5_bf16_pmc_write "WRITE_SIZE" $T5 --precision bf16
grep -E "^kernel|k_bn_csr_l2_bwd|k_bn_apply_edge|k_lin64_stream<1, false|k_edge_uv_lin1|k_csr_mean_fwd|k_edge_uv_sums|k_bn_csr_partial" $R/gpurun_out/${TAG}_train_cfg5_fp32_pmc_fetch.txt | cut -c1-160
grep -E "^kernel|k_bn_csr_l2_bwd|k_bn_apply_edge|k_lin64_stream<1, false|k_edge_uv_lin1|k_csr_mean_fwd|k_edge_uv_sums|k_bn_csr_partial" $R/gpurun_out/${TAG}_train_cfg5_fp32_pmc_write.txt | cut -c1-160
