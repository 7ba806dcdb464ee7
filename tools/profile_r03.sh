# Round-3 evidence: kernel stats of the eval forward (cfg 2 fp32, cfg 5 fp32 and bf16 storage), SQ counter passes and
# FETCH/WRITE passes of the bf16 cfg-5 forward (the configuration BASELINE.json configs[4] names), the cfg-3 train step,
# and the traffic table behind bench.py's roofline.traffic.  Outputs: gpurun_out/r03_*.txt (copied to profiles/).
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
TAG=${TAG:-r03}
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
COMMON="--streams 1 --no-cpu-baseline --no-roofline --no-extras"
F2="python $R/bench.py --config 2 --steps 50 --warmup 5 $COMMON"
F5="python $R/bench.py --config 5 --steps 10 --warmup 3 $COMMON"
F5H="python $R/bench.py --config 5 --precision bf16 --steps 10 --warmup 3 $COMMON"
run_stats ${TAG}_fwd_cfg2_kernel_stats $F2
run_stats ${TAG}_fwd_cfg5_kernel_stats $F5
run_stats ${TAG}_fwd_cfg5_bf16_kernel_stats $F5H
run_pmc ${TAG}_fwd_cfg5_bf16_pmc_sq_a "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD" $F5H
run_pmc ${TAG}_fwd_cfg5_bf16_pmc_sq_b "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES" $F5H
run_pmc ${TAG}_fwd_cfg5_bf16_pmc_fetch "FETCH_SIZE" $F5H
run_pmc ${TAG}_fwd_cfg5_bf16_pmc_write "WRITE_SIZE" $F5H
run_pmc ${TAG}_fwd_cfg5_pmc_fetch "FETCH_SIZE" $F5
run_pmc ${TAG}_fwd_cfg5_pmc_write "WRITE_SIZE" $F5
run_pmc ${TAG}_fwd_cfg2_pmc_fetch "FETCH_SIZE" $F2
run_pmc ${TAG}_fwd_cfg2_pmc_write "WRITE_SIZE" $F2
run_stats ${TAG}_train_cfg3_kernel_stats python $R/bench.py --mode train --config 3 --steps 15 --warmup 3 --no-cpu-baseline --no-roofline --no-extras
cd $R
python tools/pmc_traffic.py 2 gpurun_out/${TAG}_fwd_cfg2_pmc_fetch.txt gpurun_out/${TAG}_fwd_cfg2_pmc_write.txt ${TAG}_fwd_cfg2
python tools/pmc_traffic.py 5 gpurun_out/${TAG}_fwd_cfg5_pmc_fetch.txt gpurun_out/${TAG}_fwd_cfg5_pmc_write.txt ${TAG}_fwd_cfg5
python tools/pmc_traffic.py 5 gpurun_out/${TAG}_fwd_cfg5_bf16_pmc_fetch.txt gpurun_out/${TAG}_fwd_cfg5_bf16_pmc_write.txt ${TAG}_fwd_cfg5_bf16 bf16
cp profiles/pmc_traffic.json gpurun_out/pmc_traffic.json
ls gpurun_out | grep ${TAG}_
