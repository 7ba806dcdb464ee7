# FETCH_SIZE / WRITE_SIZE passes of the bench command (one forward at a time) for cfg 2 and cfg 5, then
# profiles/pmc_traffic.json (read by bench.py for roofline.traffic).  TAG names the committed summary files.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
TAG=${TAG:-r02}
for cfg in 2 5; do
  steps=50; [ $cfg = 5 ] && steps=10
  CMD="python $R/bench.py --config $cfg --steps $steps --warmup 5 --streams 1 --no-cpu-baseline --no-roofline --no-extras"
  for ctr in FETCH_SIZE WRITE_SIZE; do
    n=pmc_${cfg}_$ctr
    rm -rf $R/gpurun_out/$n
    timeout 600 rocprofv3 --kernel-trace --pmc $ctr -d $R/gpurun_out/$n --output-format rocpd -- $CMD > $R/gpurun_out/$n.log 2>&1
    f=$(find $R/gpurun_out/$n -name "*.db" | head -1)
    if [ -n "$f" ]; then python $R/tools/rocpd_pmc.py $f > $R/gpurun_out/$n.txt; else tail -5 $R/gpurun_out/$n.log; fi
    rm -rf $R/gpurun_out/$n
  done
  cp $R/gpurun_out/pmc_${cfg}_FETCH_SIZE.txt $R/gpurun_out/${TAG}_fwd_cfg${cfg}_pmc_fetch.txt
  cp $R/gpurun_out/pmc_${cfg}_WRITE_SIZE.txt $R/gpurun_out/${TAG}_fwd_cfg${cfg}_pmc_write.txt
  (cd $R && python tools/pmc_traffic.py $cfg gpurun_out/${TAG}_fwd_cfg${cfg}_pmc_fetch.txt gpurun_out/${TAG}_fwd_cfg${cfg}_pmc_write.txt ${TAG}_fwd_cfg${cfg})
done
cp $R/profiles/pmc_traffic.json $R/gpurun_out/pmc_traffic.json
