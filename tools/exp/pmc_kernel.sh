# SQ counter passes over one command; env: CMD (command line), KRE (kernel name regex), TAG
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
TAG=${TAG:-k}
i=0
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAIT_ANY" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_VALU_MFMA_BUSY_CYCLES" \
           "SQ_INSTS_VMEM_WR SQ_INSTS_MFMA SQ_ACTIVE_INST_SCA SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INST_CYCLES_VMEM SQ_THREAD_CYCLES_VALU"; do
  i=$((i+1))
  rm -rf $R/gpurun_out/pmc_${TAG}_$i
  timeout 300 rocprofv3 --kernel-trace --pmc $set -d $R/gpurun_out/pmc_${TAG}_$i --output-format rocpd -- $CMD > $R/gpurun_out/pmc_${TAG}_$i.log 2>&1
  f=$(find $R/gpurun_out/pmc_${TAG}_$i -name "*.db" | head -1)
  if [ -n "$f" ]; then python $R/tools/rocpd_pmc.py $f | grep -E "^kernel|$KRE" > $R/gpurun_out/pmc_${TAG}_$i.txt; else tail -5 $R/gpurun_out/pmc_${TAG}_$i.log; fi
  rm -rf $R/gpurun_out/pmc_${TAG}_$i
done
cat $R/gpurun_out/pmc_${TAG}_*.txt
