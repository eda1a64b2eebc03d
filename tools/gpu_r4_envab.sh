#!/usr/bin/env bash
# GPU box: same-box A/B of ENVIRONMENT variants of the shipped library on a bench workload (3 runs each, alternating).
# usage: BENCH_ARGS="--workload stress" tools/gpu_r4_envab.sh "" "PMX_CC_FUSE_MLEN_BIG=0" ...
cd ${GRAFT_REPO_ROOT:-/root/repo}
for rep in 1 2 3; do
  for v in "$@"; do
    env $v python bench.py --steps ${STEPS:-30} --warmup 5 --repeat 1 --no-cpu-baseline --no-end-to-end ${BENCH_ARGS} 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('[$v]', round(d['ms_per_step'],4), d['kernel_ms_per_step'])"
  done
done
