cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
CMD="python $R/bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-roofline ${BENCH_EXTRA:-}"
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS -d $R/gpurun_out/pmc_sq --output-format rocpd -- $CMD > $R/gpurun_out/pmc_sq.log 2>&1
cd $R
f=$(find gpurun_out/pmc_sq -name "*.db" | head -1); python tools/rocpd_pmc.py $f > gpurun_out/pmc_sq.txt
find gpurun_out -name "*.db" -delete
