set -e
cd yolat_vectorgraphicsrecognition_amd/csrc && make >/dev/null
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-function -DYOLAT_EDGE_STAMPS -c edge.hip -o /tmp/edge_stamps.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../tools/exp/libyolat_hip_stamps.so $(ls *.o | grep -v "^edge.o") /tmp/edge_stamps.o
cd ../..
YOLAT_LIB_PATH=$PWD/tools/exp/libyolat_hip_stamps.so python tools/exp/r06_edge_stamps.py ${1:-2}
