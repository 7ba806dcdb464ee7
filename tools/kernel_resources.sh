#!/bin/bash
# Usage: tools/kernel_resources.sh csrc/file.hip [filter] — VGPRs / spills / scratch / LDS of every kernel in a TU
src=$(realpath "$1"); base=$(basename "$src" .hip); cd /tmp
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -c --save-temps -o /tmp/$base.o "$src" 2>&1 | grep -E "error"
awk '/^ +\.name:/{name=$2} /\.vgpr_count:|\.vgpr_spill_count:|\.private_segment_fixed_size:|\.group_segment_fixed_size:|\.sgpr_count:/{v[$1]=$2} /\.wavefront_size:/{printf "%-90s vgpr=%s spill=%s scratch=%s lds=%s\n", name, v[".vgpr_count:"], v[".vgpr_spill_count:"], v[".private_segment_fixed_size:"], v[".group_segment_fixed_size:"]}' /tmp/$base-hip-amdgcn-amd-amdhsa-gfx950.s | grep -i "${2:-.}" | cut -c1-170
