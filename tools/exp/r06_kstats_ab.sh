# per-kernel rocprof averages of the cfg-2 forward for the tree's build and tools/exp/libyolat_hip_prev.so (round 6)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for tag in new prev; do
  if [ $tag = prev ]; then export YOLAT_LIB_PATH=$R/tools/exp/libyolat_hip_prev.so; else unset YOLAT_LIB_PATH; fi
  rm -rf $R/gpurun_out/kab_$tag
  timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/kab_$tag --output-format rocpd -- python $R/bench.py --config ${1:-2} --steps 200 --warmup 20 --streams 1 --no-cpu-baseline --no-roofline --no-extras > /dev/null 2>&1
  f=$(find $R/gpurun_out/kab_$tag -name "*.db" | head -1)
  echo "== $tag"; python $R/tools/rocpd_stats.py $f | head -10 | cut -c1-150
  rm -rf $R/gpurun_out/kab_$tag
done
