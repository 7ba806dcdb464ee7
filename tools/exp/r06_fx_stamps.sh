# builds the stamped library (fusion_x6.hip with -DYOLAT_FX_STAMPS; every other object from the product build) and runs the probe
set -e
cd yolat_vectorgraphicsrecognition_amd/csrc && make >/dev/null
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-function -DYOLAT_FX_STAMPS -c fusion_x6.hip -o /tmp/fusion_x6_stamps.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-function -DYOLAT_H8_STAMPS -c fusion_h8.hip -o /tmp/fusion_h8_stamps.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../tools/exp/libyolat_hip_stamps.so $(ls *.o | grep -v fusion_x6.o | grep -v fusion_h8.o) /tmp/fusion_x6_stamps.o /tmp/fusion_h8_stamps.o
cd ../..
for c in ${1:-2 5}; do YOLAT_LIB_PATH=$PWD/tools/exp/libyolat_hip_stamps.so python tools/exp/r06_fx_stamps.py $c $2; done
