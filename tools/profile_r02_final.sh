# Round-2 final evidence: kernel stats of the eval forward at cfg 2 and cfg 5, of the cfg-5 train step (fp32, bf16),
# SQ counter passes of the cfg-5 forward, and the FETCH/WRITE passes behind profiles/pmc_traffic.json.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
TAG=r02_final
run_stats() {  # name cmd...
  n=$1; shift
  rm -rf $R/gpurun_out/$n
  timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/$n --output-format rocpd -- "$@" > $R/gpurun_out/$n.log 2>&1
  f=$(find $R/gpurun_out/$n -name "*.db" | head -1)
  if [ -n "$f" ]; then python $R/tools/rocpd_stats.py $f > $R/gpurun_out/$n.txt; else tail -5 $R/gpurun_out/$n.log; fi
  rm -rf $R/gpurun_out/$n
}
run_pmc() {  # name "counters" cmd...
  n=$1; ctr=$2; shift; shift
  rm -rf $R/gpurun_out/$n
  timeout 600 rocprofv3 --kernel-trace --pmc $ctr -d $R/gpurun_out/$n --output-format rocpd -- "$@" > $R/gpurun_out/$n.log 2>&1
  f=$(find $R/gpurun_out/$n -name "*.db" | head -1)
  if [ -n "$f" ]; then python $R/tools/rocpd_pmc.py $f > $R/gpurun_out/$n.txt; else tail -5 $R/gpurun_out/$n.log; fi
  rm -rf $R/gpurun_out/$n
}
F2="python $R/bench.py --config 2 --steps 50 --warmup 5 --streams 1 --no-cpu-baseline --no-roofline --no-extras"
F5="python $R/bench.py --config 5 --steps 10 --warmup 3 --streams 1 --no-cpu-baseline --no-roofline --no-extras"
run_stats ${TAG}_fwd_cfg2_kernel_stats $F2
run_stats ${TAG}_fwd_cfg5_kernel_stats $F5
run_pmc ${TAG}_fwd_cfg5_pmc_sq_a "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD" $F5
run_pmc ${TAG}_fwd_cfg5_pmc_sq_b "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SALU SQ_INSTS_VMEM_WR SQ_VALU_MFMA_BUSY_CYCLES" $F5
for prec in fp32 bf16; do
  run_stats ${TAG}_train_cfg5_${prec}_kernel_stats python $R/bench.py --mode train --config 5 --precision $prec --steps 6 --warmup 2 --no-cpu-baseline --no-roofline --no-extras
done
TAG=r02 bash $R/tools/pmc_traffic.sh
ls $R/gpurun_out | head -50
