# SQ counter passes of the cfg-2 eval forward (one forward at a time), per (kernel, grid)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
CMD="python $R/bench.py --config 2 --steps 50 --warmup 5 --streams 1 --no-cpu-baseline --no-roofline --no-extras"
run_pmc() {
  n=$1; ctr=$2
  rm -rf $R/gpurun_out/$n
  timeout 600 rocprofv3 --kernel-trace --pmc $ctr -d $R/gpurun_out/$n --output-format rocpd -- $CMD > $R/gpurun_out/$n.log 2>&1
  f=$(find $R/gpurun_out/$n -name "*.db" | head -1)
  if [ -n "$f" ]; then python $R/tools/rocpd_pmc.py $f > $R/gpurun_out/$n.txt; else tail -5 $R/gpurun_out/$n.log; fi
  rm -rf $R/gpurun_out/$n
}
run_pmc r02_final_fwd_cfg2_pmc_sq_a "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD"
run_pmc r02_final_fwd_cfg2_pmc_sq_b "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SALU SQ_INSTS_VMEM_WR SQ_VALU_MFMA_BUSY_CYCLES"
grep -E "^kernel|k_fusion_rows|k_edge_uv|k_gemm_nt_sk|k_prep_rows" $R/gpurun_out/r02_final_fwd_cfg2_pmc_sq_a.txt | cut -c1-250
grep -E "^kernel|k_fusion_rows|k_edge_uv|k_gemm_nt_sk|k_prep_rows" $R/gpurun_out/r02_final_fwd_cfg2_pmc_sq_b.txt | cut -c1-250
