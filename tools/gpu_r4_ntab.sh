#!/usr/bin/env bash
# GPU box: A/B of a -D variant of the library on the stress workload: time (3 interleaved runs) + HBM traffic of the variant
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
mkdir -p gpurun_out/r4
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -mllvm -amdgpu-atomic-optimizer-strategy=DPP -fPIC -shared -w $1 -o /tmp/libvar.so pymasc_amd/csrc/*.hip || exit 1
for rep in 1 2 3; do
  for lib in "" /tmp/libvar.so; do
    if [ -n "$lib" ]; then export PYMASC_AMD_LIB=$lib; else unset PYMASC_AMD_LIB; fi
    python bench.py --workload stress --steps 30 --warmup 5 --no-cpu-baseline --no-end-to-end 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('[$lib]', round(d['ms_per_step'],4), d['kernel_ms_per_step'])"
  done
done
export PYMASC_AMD_LIB=/tmp/libvar.so
tools/tools_r4_profile.sh r4/prof_var --workload stress | tail -3
