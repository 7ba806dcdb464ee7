CL_STAMPS=1 timeout 300 python tools/exp/conv_local_abl.py 5 0 20 2>&1 | grep -v amdgpu
CL_STAMPS=1 CL_CSR=1 timeout 300 python tools/exp/conv_local_abl.py 5 0 20 2>&1 | grep -v amdgpu
