for cfg in 5 3; do
for f in 1 0; do
  FOLD=$f YOLAT_EDGE_WS=0 timeout 120 python tools/exp/edge_bench.py $cfg
  FOLD=$f YOLAT_EDGE_WS=1 YOLAT_EDGE_X6=1 timeout 120 python tools/exp/edge_bench.py $cfg
done; done
for w in 256 384 768 1024; do FOLD=1 YOLAT_EDGE_WS=1 YOLAT_EDGE_WGS=$w timeout 120 python tools/exp/edge_bench.py 5; done
