# SQ counter passes of the one-stream cfg-5 train step (fp32): where a wave's cycles go in the training kernels.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
TAG=${TAG:-r04}
run_pmc() {
  n=$1; ctr=$2; shift; shift
  rm -rf $R/gpurun_out/$n
  timeout 600 rocprofv3 --kernel-trace --pmc $ctr -d $R/gpurun_out/$n --output-format rocpd -- "$@" > $R/gpurun_out/$n.log 2>&1
  f=$(find $R/gpurun_out/$n -name "*.db" | head -1)
  if [ -n "$f" ]; then python $R/tools/rocpd_pmc.py $f > $R/gpurun_out/$n.txt; else tail -5 $R/gpurun_out/$n.log; fi
  rm -rf $R/gpurun_out/$n
}
T5="python $R/bench.py --mode train --config 5 --steps 6 --warmup 2 --no-cpu-baseline --no-roofline --no-extras --no-side-stream"
run_pmc ${TAG}_train_cfg5_fp32_pmc_sq_a "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD" $T5
run_pmc ${TAG}_train_cfg5_fp32_pmc_sq_b "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES" $T5
grep -E "^kernel|k_bn_csr_l2_bwd|k_bn_apply_edge|k_lin64_stream<1, false" $R/gpurun_out/${TAG}_train_cfg5_fp32_pmc_sq_a.txt | cut -c1-260
grep -E "^kernel|k_bn_csr_l2_bwd|k_bn_apply_edge|k_lin64_stream<1, false" $R/gpurun_out/${TAG}_train_cfg5_fp32_pmc_sq_b.txt | cut -c1-260
