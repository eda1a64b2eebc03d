#!/usr/bin/env bash
# Container helper: per-kernel VGPRs / scratch / LDS / occupancy of one .hip file (hipcc remark output, no GPU needed).
# usage: tools/kres.sh pymasc_amd/csrc/kernels_sparse.hip [-DFOO=1 ...]
f=$1; shift
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -mllvm -amdgpu-atomic-optimizer-strategy=DPP "$@" -c \
  -Rpass-analysis=kernel-resource-usage "$f" -o /dev/null 2>&1 | python3 -c "
import re,sys
cur=None
for line in sys.stdin:
    m=re.search(r'remark: .*?:\d+:\d+: +(.*?) \[-Rpass', line) or re.search(r': +(Function Name|[A-Za-z ]+): (.*) \[-Rpass', line)
    t=re.search(r'(Function Name|Name): (\S+)', line)
    if t: cur=t.group(2); print(); print(cur[:60], end=' ')
    for k in ('VGPRs','AGPRs','ScratchSize [bytes/lane]','Occupancy [waves/SIMD]','LDS Size [bytes/block]','SGPRs'):
        t=re.search(re.escape(k)+r': (\d+)', line)
        if t: print(k.split()[0]+'='+t.group(1), end=' ')
print()
"
