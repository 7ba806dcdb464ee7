cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
n=seq_cfg3
rm -rf $R/gpurun_out/$n
timeout 600 rocprofv3 --kernel-trace -d $R/gpurun_out/$n --output-format rocpd -- python $R/bench.py --mode train --config 3 --steps 6 --warmup 2 --no-cpu-baseline --no-roofline --no-extras > $R/gpurun_out/$n.log 2>&1
f=$(find $R/gpurun_out/$n -name "*.db" | head -1)
python $R/tools/rocpd_sequence.py $f > $R/gpurun_out/$n.txt
rm -rf $R/gpurun_out/$n
head -3 $R/gpurun_out/$n.txt
