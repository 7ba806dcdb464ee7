# Quick kernel-stats passes while iterating: TAG=<tag> bash tools/profile_quick.sh <name> ...   names: f2 f5 f5h t3 t5 t5h
# (forward cfg 2 / cfg 5 fp32 / cfg 5 bf16 storage; one-stream train step cfg 3 / cfg 5 fp32 / cfg 5 bf16 storage).
# Outputs: gpurun_out/${TAG}_<name>_kernel_stats.txt
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
TAG=${TAG:-q}
run_stats() {  # name cmd...
  n=$1; shift
  rm -rf $R/gpurun_out/$n
  timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/$n --output-format rocpd -- "$@" > $R/gpurun_out/$n.log 2>&1
  f=$(find $R/gpurun_out/$n -name "*.db" | head -1)
  if [ -n "$f" ]; then python $R/tools/rocpd_stats.py $f > $R/gpurun_out/$n.txt; else tail -5 $R/gpurun_out/$n.log; fi
  rm -rf $R/gpurun_out/$n
}
COMMON="--streams 1 --no-cpu-baseline --no-roofline --no-extras"
T="--no-cpu-baseline --no-roofline --no-extras --no-side-stream"
for w in "$@"; do
  case $w in
    f2) run_stats ${TAG}_f2_kernel_stats python $R/bench.py --config 2 --steps 50 --warmup 5 $COMMON ;;
    f5) run_stats ${TAG}_f5_kernel_stats python $R/bench.py --config 5 --steps 10 --warmup 3 $COMMON ;;
    f5h) run_stats ${TAG}_f5h_kernel_stats python $R/bench.py --config 5 --precision bf16 --steps 10 --warmup 3 $COMMON ;;
    t3) run_stats ${TAG}_t3_kernel_stats python $R/bench.py --mode train --config 3 --steps 15 --warmup 3 $T ;;
    t5) run_stats ${TAG}_t5_kernel_stats python $R/bench.py --mode train --config 5 --steps 8 --warmup 2 $T ;;
    t5h) run_stats ${TAG}_t5h_kernel_stats python $R/bench.py --mode train --config 5 --steps 8 --warmup 2 --precision bf16 $T ;;
  esac
done
