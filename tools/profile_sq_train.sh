cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
CMD="python $R/bench.py --mode train --steps 6 --warmup 2 --no-cpu-baseline --no-roofline"
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS -d $R/gpurun_out/pmc_sqt --output-format rocpd -- $CMD > $R/gpurun_out/pmc_sqt.log 2>&1
cd $R
f=$(find gpurun_out/pmc_sqt -name "*.db" | head -1); python tools/rocpd_pmc.py $f > gpurun_out/pmc_sqt.txt
find gpurun_out -name "*.db" -delete
