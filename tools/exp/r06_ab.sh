for i in 1 2; do
echo "== new"; timeout 600 python tools/exp/cfg5_locality_bench.py 5 50 2>&1 | grep -E "ms per forward|conv_local_bf16" | head -4
echo "== prev"; YOLAT_LIB_PATH=$PWD/tools/exp/libyolat_hip_prev.so timeout 600 python tools/exp/cfg5_locality_bench.py 5 50 2>&1 | grep -E "ms per forward|conv_local_bf16" | head -4
done
