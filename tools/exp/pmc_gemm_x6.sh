cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
run_pmc() {
  n=$1; ctr=$2
  rm -rf $R/gpurun_out/$n
  timeout 600 rocprofv3 --kernel-trace --pmc $ctr -d $R/gpurun_out/$n --output-format rocpd -- python $R/tools/exp/linear_x6_bench.py 8000 2304 512 > $R/gpurun_out/$n.log 2>&1
  f=$(find $R/gpurun_out/$n -name "*.db" | head -1)
  python $R/tools/rocpd_pmc.py $f | grep -v "at::\|rocclr\|k_gemm_nt" | head -12
  rm -rf $R/gpurun_out/$n
}
run_pmc gx_a "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_VALU_MFMA_BUSY_CYCLES"
run_pmc gx_b "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD GRBM_GUI_ACTIVE"
