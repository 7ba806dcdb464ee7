# kernel trace of the cfg-5 training step, fp32 and bf16 storage (TAG names the outputs)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
TAG=${TAG:-r02}
for prec in fp32 bf16; do
  n=${TAG}_train_cfg5_${prec}_kernel_stats
  rm -rf $R/gpurun_out/$n
  timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/$n --output-format rocpd -- python $R/bench.py --mode train --config 5 --precision $prec --steps 6 --warmup 2 --no-cpu-baseline --no-roofline --no-extras > $R/gpurun_out/$n.log 2>&1
  f=$(find $R/gpurun_out/$n -name "*.db" | head -1)
  if [ -n "$f" ]; then python $R/tools/rocpd_stats.py $f > $R/gpurun_out/$n.txt; else tail -5 $R/gpurun_out/$n.log; fi
  rm -rf $R/gpurun_out/$n
done
