#!/usr/bin/env bash
# GPU box: same-box A/B of -D variants of the library on the default bench (kernel time + step time, 3 runs each).
# usage: tools/gpu_r4_ab.sh "<flags of variant 1>" "<flags of variant 2>" ...   ("" = the shipped build)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
i=0
for flags in "$@"; do
  i=$((i+1))
  /opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -mllvm -amdgpu-atomic-optimizer-strategy=DPP -fPIC -shared $flags -o /tmp/libvar$i.so pymasc_amd/csrc/*.hip 2>/dev/null || { echo "build failed: $flags"; exit 1; }
done
for rep in 1 2 3; do
  i=0
  for flags in "$@"; do
    i=$((i+1))
    PYMASC_AMD_LIB=/tmp/libvar$i.so python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-end-to-end --no-ingest ${BENCH_ARGS} 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('[$flags]', round(d['ms_per_step'],4), d['kernel_ms_per_step'])"
  done
done
