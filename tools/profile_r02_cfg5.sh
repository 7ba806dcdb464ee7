# Round-2 evidence for the cfg-5 fp32 path: kernel trace + SQ / FETCH / WRITE counter passes of the eval forward
# (k_edge_uv_mlp2_mean is the kernel under study) and a kernel trace of the cfg-5 train step
# (k_csr_mean_fwd_v4 at E = 1.2 M).  TAG names the output files (gpurun_out/${TAG}_*.txt).
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
TAG=${TAG:-r02}
FWD="python $R/bench.py --config 5 --steps 10 --warmup 3 --streams 1 --no-cpu-baseline --no-roofline ${BENCH_EXTRA:-}"
TRN="python $R/bench.py --mode train --config 5 --steps 6 --warmup 2 --no-cpu-baseline --no-roofline"
run_stats() {  # name cmd...
  n=$1; shift
  rm -rf $R/gpurun_out/$n
  timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/$n --output-format rocpd -- "$@" > $R/gpurun_out/$n.log 2>&1
  f=$(find $R/gpurun_out/$n -name "*.db" | head -1)
  if [ -n "$f" ]; then python $R/tools/rocpd_stats.py $f > $R/gpurun_out/$n.txt; else tail -5 $R/gpurun_out/$n.log; fi
  find $R/gpurun_out/$n -name "*.db" -delete
}
run_pmc() {  # name "counters" cmd...
  n=$1; ctr=$2; shift; shift
  rm -rf $R/gpurun_out/$n
  timeout 600 rocprofv3 --kernel-trace --pmc $ctr -d $R/gpurun_out/$n --output-format rocpd -- "$@" > $R/gpurun_out/$n.log 2>&1
  f=$(find $R/gpurun_out/$n -name "*.db" | head -1)
  if [ -n "$f" ]; then python $R/tools/rocpd_pmc.py $f > $R/gpurun_out/$n.txt; else tail -5 $R/gpurun_out/$n.log; fi
  find $R/gpurun_out/$n -name "*.db" -delete
}
run_stats ${TAG}_fwd_cfg5_kernel_stats $FWD
run_pmc ${TAG}_fwd_cfg5_pmc_sq_a "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD" $FWD
run_pmc ${TAG}_fwd_cfg5_pmc_sq_b "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SALU SQ_INSTS_VMEM_WR SQ_VALU_MFMA_BUSY_CYCLES" $FWD
run_pmc ${TAG}_fwd_cfg5_pmc_fetch "FETCH_SIZE" $FWD
run_pmc ${TAG}_fwd_cfg5_pmc_write "WRITE_SIZE" $FWD
if [ -z "$SKIP_TRAIN" ]; then run_stats ${TAG}_train_cfg5_kernel_stats $TRN; fi
ls -la $R/gpurun_out/ | head -40
