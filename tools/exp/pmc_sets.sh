# PMC passes over one command with caller-given counter sets; env: CMD, KRE (kernel regex), TAG, SETS ("a b c;d e f")
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
TAG=${TAG:-k}
i=0
IFS=';' read -ra ARR <<< "$SETS"
for set in "${ARR[@]}"; do
  i=$((i+1))
  rm -rf $R/gpurun_out/pmcs_${TAG}_$i
  timeout 300 rocprofv3 --kernel-trace --pmc $set -d $R/gpurun_out/pmcs_${TAG}_$i --output-format rocpd -- $CMD > $R/gpurun_out/pmcs_${TAG}_$i.log 2>&1
  f=$(find $R/gpurun_out/pmcs_${TAG}_$i -name "*.db" | head -1)
  if [ -n "$f" ]; then python $R/tools/rocpd_pmc.py $f | grep -E "^kernel|$KRE"; else tail -5 $R/gpurun_out/pmcs_${TAG}_$i.log; fi
  rm -rf $R/gpurun_out/pmcs_${TAG}_$i
done
