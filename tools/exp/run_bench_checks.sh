python bench.py --steps 20 --warmup 5 > gpurun_out/bench_n1.json 2> gpurun_out/bench_n1.err; tail -c 300 gpurun_out/bench_n1.err
YOLAT_BENCH_BACKEND=gloo YOLAT_BENCH_DEVICE=0 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 20 --warmup 5 > gpurun_out/bench_n2_gloo.json 2> gpurun_out/bench_n2_gloo.err; tail -c 600 gpurun_out/bench_n2_gloo.err
TAG=r02 bash tools/pmc_traffic.sh > gpurun_out/pmc_traffic.log 2>&1; tail -5 gpurun_out/pmc_traffic.log
FOLD=1 YOLAT_EDGE_VARIANT=3 TAG=x6final bash tools/exp/pmc_edge.sh > gpurun_out/pmc_edge_x6final.txt 2>&1
