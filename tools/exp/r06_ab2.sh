# A/B of two builds of the library on one box (round 6): the tree's build against tools/exp/libyolat_hip_prev.so
# ("prev" = another commit of the library: bash tools/exp/r06_build_prev.sh <commit>)
# usage: bash tools/exp/r06_ab2.sh "5:bf16 2:bf16" [train cfgs, e.g. "3:fp32 5:fp32"]
CASES=${1:-"5:bf16 2:bf16 1:bf16 2:fp32 5:fp32"}
TRAIN=${2:-""}
for i in 1 2; do
  echo "== new (round $i)"; timeout 900 python tools/exp/r06_fusion_ab.py YOLAT_AB_NONE - $CASES 2>&1 | grep "^cfg" | awk 'NR%2==0'
  echo "== prev (round $i)"; YOLAT_LIB_PATH=$PWD/tools/exp/libyolat_hip_prev.so timeout 900 python tools/exp/r06_fusion_ab.py YOLAT_AB_NONE - $CASES 2>&1 | grep "^cfg" | awk 'NR%2==0'
  for t in $TRAIN; do
    c=${t%%:*}; p=${t##*:}
    echo "== new train $t"; timeout 900 python tools/exp/train_plan_bench.py $c $p 40 2>&1 | grep "one-call plan"
    echo "== prev train $t"; YOLAT_LIB_PATH=$PWD/tools/exp/libyolat_hip_prev.so timeout 900 python tools/exp/train_plan_bench.py $c $p 40 2>&1 | grep "one-call plan"
  done
done
