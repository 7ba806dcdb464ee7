# LDS bank conflicts / instruction mix of k_conv_local_h by PHASE: the kernel run with phase-ablation masks under separate
# rocprofv3 --pmc passes.  Output: gpurun_out/r06_conv_local_lds_by_phase.txt
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r06_conv_local_lds_by_phase.txt
: > $OUT
CT="SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_INSTS_MFMA SQ_INSTS_VALU SQ_BUSY_CYCLES"
for abl in 0 1 6 7 8 15; do
  n=ldspmc_$abl
  rm -rf $R/gpurun_out/$n
  timeout 300 rocprofv3 --kernel-trace --pmc $CT -d $R/gpurun_out/$n --output-format rocpd -- python $R/tools/exp/conv_local_abl.py 5 $abl 5 > $R/gpurun_out/$n.log 2>&1
  f=$(find $R/gpurun_out/$n -name "*.db" | head -1)
  echo "==== abl $abl ($(grep 'us per launch' $R/gpurun_out/$n.log | tail -1))" >> $OUT
  if [ -n "$f" ]; then python $R/tools/rocpd_pmc.py $f | grep -E "kernel|conv_local_h" >> $OUT; else tail -5 $R/gpurun_out/$n.log >> $OUT; fi
  rm -rf $R/gpurun_out/$n $R/gpurun_out/$n.log
done
cat $OUT
