#!/usr/bin/env bash
# profiling helper (GPU box): A/B compile-time variants of the library on the default bench, same box, same run.
# usage: VARIANTS="SP_WAVES=3;SP_WAVES=4 -DSP_PIPE=1" MODES="both ncc" tools/tools_variants.sh [bench args]
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
mkdir -p gpurun_out/r2
IFS=';' read -ra VS <<< "${VARIANTS:-SP_WAVES=3}"
for v in "${VS[@]}"; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -mllvm -amdgpu-atomic-optimizer-strategy=DPP -fPIC -shared -D$v -o /tmp/libabl.so pymasc_amd/csrc/*.hip 2>/dev/null || { echo "$v: build failed"; continue; }
  for mode in ${MODES:-both}; do
    PYMASC_AMD_LIB=/tmp/libabl.so python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-end-to-end --mode $mode "$@" 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v', '$mode', round(d['ms_per_step'],4), d['kernel_ms_per_step'])"
  done
done
