#!/usr/bin/env bash
# GPU box: same-box A/B of prebuilt libraries on the bench (3 interleaved runs each).
# usage: tools/gpu_r4_libab.sh <lib1.so> <lib2.so> ...   ("" = the in-tree library); BENCH_ARGS env adds bench flags
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
for rep in 1 2 3; do
  for lib in "$@"; do
    if [ -n "$lib" ]; then export PYMASC_AMD_LIB=$R/$lib; else unset PYMASC_AMD_LIB; fi
    python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-end-to-end ${BENCH_ARGS} 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('[$lib]', round(d['ms_per_step'],4), d['kernel_ms_per_step'])"
  done
done
