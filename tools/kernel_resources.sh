#!/bin/bash
# Prints VGPR/SGPR/scratch/occupancy per kernel of vr_kernels.hip (hipcc remark output).
cd "$(dirname "$0")/.."
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off \
  -fhip-fp32-correctly-rounded-divide-sqrt -fno-gpu-flush-denormals-to-zero -x hip \
  -I include -I volrend_amd/csrc -c volrend_amd/csrc/vr_kernels.hip -o /tmp/vr_k.o \
  -Rpass-analysis=kernel-resource-usage "$@" 2>&1 | python3 -c '
import sys,re
cur={}
for line in sys.stdin:
    m2=re.search(r":\s+(Function Name|Name|TotalSGPRs|VGPRs|AGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|LDS Size \[bytes/block\]): (\S+)", line)
    if not m2: continue
    k,v=m2.group(1),m2.group(2)
    if k in ("Function Name","Name"):
        if cur: print(cur)
        cur={"name":v}
    else: cur[k.split(" ")[0]]=v
if cur: print(cur)
' | sed -e 's/_ZN2vr12_GLOBAL__N_1//'
