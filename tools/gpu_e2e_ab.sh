#!/usr/bin/env bash
# GPU box: same-box A/B of -D variants of the library on the bench's end-to-end legs.  usage: tools/gpu_e2e_ab.sh "<flags>" ...
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
i=0
for flags in "$@"; do
  i=$((i+1))
  /opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -mllvm -amdgpu-atomic-optimizer-strategy=DPP -fPIC -shared $flags -o /tmp/libvar$i.so pymasc_amd/csrc/*.hip 2>/dev/null || { echo "build failed: $flags"; exit 1; }
done
for rep in 1 2; do
  i=0
  for flags in "$@"; do
    i=$((i+1))
    PYMASC_AMD_LIB=/tmp/libvar$i.so python bench.py --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('[$flags]', round(d['ms_per_step'],4), 'e2e', round(d['end_to_end']['ms_per_step'],3), 'calc', round(d['end_to_end_calculator']['ms_per_step'],3))"
  done
done
