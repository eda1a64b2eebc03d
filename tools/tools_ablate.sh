#!/usr/bin/env bash
# profiling helper (GPU box): builds ablated variants of the library and times the bench kernels with each
set -e
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
for v in ${VARIANTS:-BASE SP_ABL_NOFLUSH}; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -mllvm -amdgpu-atomic-optimizer-strategy=DPP -fPIC -shared -D$v -o /tmp/libabl.so pymasc_amd/csrc/*.hip 2>/dev/null
  for mode in both ncc; do
    PYMASC_AMD_LIB=/tmp/libabl.so python bench.py --steps 5 --warmup 2 --no-cpu-baseline --mode $mode "$@" 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v', '$mode', round(d['ms_per_step'],3), d['kernel_ms_per_step'])"
  done
done
