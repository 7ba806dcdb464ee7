# PMC passes over the cfg-5 eval forward (BENCH_EXTRA picks the precision); two separate --pmc runs
# (a third pass with TCC_HIT/TCC_MISS/FETCH_SIZE aborted inside rocprofv3 at this size and is not collected)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
CMD="python $R/bench.py --config 5 --steps 10 --warmup 3 --streams 1 --no-cpu-baseline --no-roofline ${BENCH_EXTRA:-}"
i=0
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SALU SQ_INSTS_VMEM_WR SQ_VALU_MFMA_BUSY_CYCLES"; do
  i=$((i+1))
  rm -rf $R/gpurun_out/pmc5_$i
  timeout 600 rocprofv3 --kernel-trace --pmc $set -d $R/gpurun_out/pmc5_$i --output-format rocpd -- $CMD > $R/gpurun_out/pmc5_$i.log 2>&1
  f=$(find $R/gpurun_out/pmc5_$i -name "*.db" | head -1)
  if [ -n "$f" ]; then python $R/tools/rocpd_pmc.py $f > $R/gpurun_out/pmc5_$i.txt; else tail -5 $R/gpurun_out/pmc5_$i.log; fi
  find $R/gpurun_out/pmc5_$i -name "*.db" -delete
done
