# Round-6 evidence: kernel stats of the eval forward (cfg 2 fp32, cfg 5 fp32 and bf16 storage), FETCH/WRITE passes of all
# three, SQ counter passes of the cfg-5 fp32 forward (the wave-specialised fp32 edge kernel) and of the
# bf16 one, the cfg-3 train step TWICE (one stream: per-kernel times that mean something; default: overlap, GPU-busy vs
# wall), the cfg-5 train steps, and the traffic table behind bench.py's roofline.traffic.
# Outputs: gpurun_out/r04_*.txt (copied to profiles/).
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
TAG=${TAG:-r06}
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
T3="python $R/bench.py --mode train --config 3 --steps 15 --warmup 3 --no-cpu-baseline --no-roofline --no-extras"
T5="python $R/bench.py --mode train --config 5 --steps 8 --warmup 2 --no-cpu-baseline --no-roofline --no-extras"
SQA="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD"
SQB="SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES"
run_stats ${TAG}_fwd_cfg2_kernel_stats $F2
run_stats ${TAG}_fwd_cfg5_kernel_stats $F5
run_stats ${TAG}_fwd_cfg5_bf16_kernel_stats $F5H
run_pmc ${TAG}_fwd_cfg5_pmc_sq_a "$SQA" $F5
run_pmc ${TAG}_fwd_cfg5_pmc_sq_b "$SQB" $F5
run_pmc ${TAG}_fwd_cfg5_bf16_pmc_sq_a "$SQA" $F5H
run_pmc ${TAG}_fwd_cfg5_bf16_pmc_sq_b "$SQB" $F5H
run_pmc ${TAG}_fwd_cfg5_bf16_pmc_fetch "FETCH_SIZE" $F5H
run_pmc ${TAG}_fwd_cfg5_bf16_pmc_write "WRITE_SIZE" $F5H
run_pmc ${TAG}_fwd_cfg5_pmc_fetch "FETCH_SIZE" $F5
run_pmc ${TAG}_fwd_cfg5_pmc_write "WRITE_SIZE" $F5
run_pmc ${TAG}_fwd_cfg2_pmc_fetch "FETCH_SIZE" $F2
run_pmc ${TAG}_fwd_cfg2_pmc_write "WRITE_SIZE" $F2
run_stats ${TAG}_train_cfg3_one_stream_kernel_stats $T3 --no-side-stream
run_stats ${TAG}_train_cfg3_kernel_stats $T3
run_stats ${TAG}_train_cfg5_fp32_one_stream_kernel_stats $T5 --no-side-stream
run_stats ${TAG}_train_cfg5_bf16_one_stream_kernel_stats $T5 --precision bf16 --no-side-stream
cd $R
python tools/pmc_traffic.py 2 gpurun_out/${TAG}_fwd_cfg2_pmc_fetch.txt gpurun_out/${TAG}_fwd_cfg2_pmc_write.txt ${TAG}_fwd_cfg2 > /dev/null
python tools/pmc_traffic.py 5 gpurun_out/${TAG}_fwd_cfg5_pmc_fetch.txt gpurun_out/${TAG}_fwd_cfg5_pmc_write.txt ${TAG}_fwd_cfg5 > /dev/null
python tools/pmc_traffic.py 5 gpurun_out/${TAG}_fwd_cfg5_bf16_pmc_fetch.txt gpurun_out/${TAG}_fwd_cfg5_bf16_pmc_write.txt ${TAG}_fwd_cfg5_bf16 bf16 > /dev/null
cp profiles/pmc_traffic.json gpurun_out/pmc_traffic.json
ls gpurun_out | grep ${TAG}_
# the N = 1 bench line of the round's tree
cd $R && timeout 1500 python bench.py > gpurun_out/${TAG}_bench_line.json 2> gpurun_out/${TAG}_bench_stderr.txt; tail -c 1500 gpurun_out/${TAG}_bench_line.json
