# kernel stats of the training step (cfg 3 fp32, cfg 5 fp32 / bf16) + the step times of cfg 3 / 4 / 5
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
bash $R/tools/profile_train_cfg3.sh > /dev/null 2>&1
cp $R/gpurun_out/r02_train_cfg3_kernel_stats.txt $R/gpurun_out/r02_final_train_cfg3_kernel_stats.txt
for prec in fp32 bf16; do
  n=r02_final_train_cfg5_${prec}_kernel_stats
  rm -rf $R/gpurun_out/$n
  timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/$n --output-format rocpd -- python $R/bench.py --mode train --config 5 --precision $prec --steps 6 --warmup 2 --no-cpu-baseline --no-roofline --no-extras > $R/gpurun_out/$n.log 2>&1
  f=$(find $R/gpurun_out/$n -name "*.db" | head -1)
  python $R/tools/rocpd_stats.py $f > $R/gpurun_out/$n.txt
  rm -rf $R/gpurun_out/$n
done
cd $R
for c in 3 4 5; do
  python bench.py --mode train --config $c --steps 20 --warmup 5 --no-cpu-baseline --no-roofline --no-extras 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('cfg$c fp32', d['value'], d['ms_per_step'])"
done
python bench.py --mode train --config 5 --precision bf16 --steps 20 --warmup 5 --no-cpu-baseline --no-roofline --no-extras 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('cfg5 bf16', d['value'], d['ms_per_step'])"
