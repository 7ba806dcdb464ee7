CL_WAVES=1 timeout 300 python tools/exp/conv_local_abl.py 5 0 5 2>&1 | grep -v amdgpu
